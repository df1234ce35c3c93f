import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import mesh
shapes = [(2, 64, 204, 204), (2, 1, 2048, 2048), (3, 4, 100, 100, 100)]
if len(sys.argv) > 1: shapes = [shapes[int(sys.argv[1])]]
for shape in shapes:
  rng = np.random.default_rng(0)
  prev = (rng.standard_normal(shape) * 5).astype(np.float32)
  nd = shape[0]
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40,) * nd, num_iters=200, max_iters=200,
                               stop_v_max=1e-9, dt_max=1000, start_cap=0.01, final_cap=10, prefer_orig_order=True)
  x = torch.zeros(shape, device='cuda'); pv = torch.from_numpy(prev).cuda()
  kw = {} if nd == 2 else {'mesh_force': mesh.elastic_mesh_3d}
  mesh.relax_mesh(x, pv, cfg, **kw); torch.cuda.synchronize()
  t = time.perf_counter(); mesh.relax_mesh(x, pv, cfg, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t
  nodes = np.prod(shape[1:]); bpn = 56 if nd == 2 else 84
  print(shape, 'us/step %.2f' % (dt / 200 * 1e6), 'Gupd/s %.2f' % (nodes * 200 / dt / 1e9), 'alg GB/s %.0f' % (nodes * 200 * bpn / dt / 1e9))
