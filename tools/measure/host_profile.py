import sys, os, time, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from sofima_amd import flow_field, mesh
from bench import synth_pair, WARP
pre, post = synth_pair(8192, 0, warp=WARP)
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
for _ in range(3): f = calc.flow_field(a, b, 160, 40, batch_size=1024)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): f = calc.flow_field(a, b, 160, 40, batch_size=1024)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)

# the mesh step of bench.py the same way
import bench
from sofima_amd import mesh as sm
flow = f
pad = bench.PATCH // 2 // bench.STEP
prev_t = torch.from_numpy(bench.mesh_inputs(flow, pad)).cuda()
cfg = sm.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(bench.STEP, bench.STEP),
                           num_iters=1000, max_iters=1000, stop_v_max=0.005, dt_max=1000, start_cap=0.01,
                           final_cap=10, prefer_orig_order=True)
def mesh_step():
  x0 = torch.zeros(prev_t.shape, dtype=torch.float32, device='cuda')
  return sm.relax_mesh(x0, prev_t, cfg)
for _ in range(2): mesh_step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
  mesh_step(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
