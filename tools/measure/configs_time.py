import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import flow_field, mesh
from bench import synth_pair

def timed(fn, n=3):
  fn(); torch.cuda.synchronize()
  t = time.perf_counter()
  for _ in range(n): r = fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t) / n, r

calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
for name, shape, P, S, B in [('cfg1 512^2 P64 S32', (512, 512), 64, 32, 1024),
                             ('cfg3 strip 4096x400 P120 S20 b256', (4096, 400), 120, 20, 256),
                             ('cfg2 8192^2 P160 S40 b1024', (8192, 8192), 160, 40, 1024),
                             ('4096^2 P96 S32', (4096, 4096), 96, 32, 1024),
                             ('4096^2 P128 S32', (4096, 4096), 128, 32, 1024)]:
  pre, post = synth_pair(max(shape), 5)
  pre = pre[:shape[0], :shape[1]]; post = post[:shape[0], :shape[1]]
  a = torch.from_numpy(np.ascontiguousarray(pre)).cuda(); b = torch.from_numpy(np.ascontiguousarray(post)).cuda()
  dt, f = timed(lambda: calc.flow_field(a, b, P, S, batch_size=B))
  npat = f.shape[1] * f.shape[2]
  print('%s: %.2f ms, %.0f Mpix/s, %d patches, %.2f us/patch, %.2f TOP/s' % (
      name, dt * 1e3, shape[0] * shape[1] / dt / 1e6, npat, dt / npat * 1e6, 2.0 * P**4 * npat / dt / 1e12))

pre, post = synth_pair(8192, 5)
rng = np.random.default_rng(0)
pm = rng.random(pre.shape) < 0.05; qm = rng.random(post.shape) < 0.05
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
pmt = torch.from_numpy(pm).cuda(); qmt = torch.from_numpy(qm).cuda()
dt, f = timed(lambda: calc.flow_field(a, b, 160, 40, pre_mask=pmt, post_mask=qmt, batch_size=1024), 2)
print('masked cfg2 8192^2: %.1f ms, %.1f Mpix/s' % (dt * 1e3, 8192 * 8192 / dt / 1e6))

for shape in [(2, 1, 205, 205), (2, 8, 205, 205), (2, 64, 204, 204), (2, 1, 2048, 2048), (3, 1, 48, 48, 48), (3, 4, 100, 100, 100)]:
  rng = np.random.default_rng(0)
  prev = (rng.standard_normal(shape) * 5).astype(np.float32)
  nd = shape[0]
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40,) * nd, num_iters=200, max_iters=200,
                               stop_v_max=1e-9, dt_max=1000, start_cap=0.01, final_cap=10, prefer_orig_order=True)
  x = torch.zeros(shape, device='cuda'); pv = torch.from_numpy(prev).cuda()
  kw = {} if nd == 2 else {'mesh_force': mesh.elastic_mesh_3d}
  dt, _ = timed(lambda: mesh.relax_mesh(x, pv, cfg, **kw), 2)
  nodes = np.prod(shape[1:])
  bpn = 56 if nd == 2 else 84
  print(shape, 'us/step %.2f' % (dt / 200 * 1e6), 'Gupd/s %.2f' % (nodes * 200 / dt / 1e9), 'alg GB/s %.0f' % (nodes * 200 * bpn / dt / 1e9))
