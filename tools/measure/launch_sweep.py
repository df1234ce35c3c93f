import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import flow_field
from bench import synth_pair
def timed(fn, n=3):
  fn(); torch.cuda.synchronize()
  t = time.perf_counter()
  for _ in range(n): r = fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t) / n, r
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
for name, shape, P, S, B in [('cfg3', (4096, 400), 120, 20, 256), ('cfg2', (8192, 8192), 160, 40, 1024), ('P96', (4096, 4096), 96, 32, 1024)]:
  pre, post = synth_pair(max(shape), 5)
  a = torch.from_numpy(np.ascontiguousarray(pre[:shape[0], :shape[1]])).cuda(); b = torch.from_numpy(np.ascontiguousarray(post[:shape[0], :shape[1]])).cuda()
  for lp in (1, 1024, 2048, 3072, 4096, 8192, 16384):
    flow_field.LAUNCH_PATCHES = lp
    dt, f = timed(lambda: calc.flow_field(a, b, P, S, batch_size=B), 4)
    print('%s launch_patches=%d: %.2f ms' % (name, lp, dt * 1e3))
