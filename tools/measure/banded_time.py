"""Per-step cost of the band-sharded mesh driver on ONE GPU (world size 1, two
bands on the same device): what the Python step loop costs next to the kernels."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import mesh, dist
shape = (2, 64, 204, 204)
rng = np.random.default_rng(0)
prev = (rng.standard_normal(shape) * 5).astype(np.float32)
x0 = np.zeros(shape, np.float32)
for iters in (100,):
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40), num_iters=iters, max_iters=iters,
                               stop_v_max=1e-9, dt_max=1000, start_cap=0.01, final_cap=10, prefer_orig_order=True)
  for bands in (1, 2, 4):
    dist.relax_mesh_sharded(x0, prev, cfg, bands_per_rank=bands); torch.cuda.synchronize()
    t = time.perf_counter(); r = dist.relax_mesh_sharded(x0, prev, cfg, bands_per_rank=bands); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print('%d band(s) on one rank: %.1f ms for %d steps = %.1f us/step (includes the H2D / D2H of the 21 MB state)' % (
        bands, dt * 1e3, iters, dt / iters * 1e6))
  xt = torch.from_numpy(x0).cuda(); pt = torch.from_numpy(prev).cuda()
  mesh.relax_mesh(xt, pt, cfg); torch.cuda.synchronize()
  t = time.perf_counter(); mesh.relax_mesh(xt, pt, cfg); torch.cuda.synchronize()
  print('single-device relax_mesh: %.1f us/step' % ((time.perf_counter() - t) / iters * 1e6))
