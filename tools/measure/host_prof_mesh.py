"""Host-side profile of the headline mesh call (relax_mesh of [2,1,205,205], 1000 FIRE steps)."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from sofima_amd import mesh
rng = np.random.default_rng(0)
prev = torch.from_numpy((rng.standard_normal((2, 1, 205, 205)) * 3).astype(np.float32)).cuda()
cfg = bench.mesh_config() if hasattr(bench, 'mesh_config') else mesh.IntegrationConfig(
    dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40), num_iters=1000, max_iters=1000,
    stop_v_max=1e-9, dt_max=1000, start_cap=0.01, final_cap=10, prefer_orig_order=True)
def step():
  x0 = torch.zeros(prev.shape, dtype=torch.float32, device='cuda')
  return mesh.relax_mesh(x0, prev, cfg)
for _ in range(3): step()
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
pr.disable()
print('call %.3f ms' % (dt * 1e3))
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
