"""Exact pruning on the other flow geometries of BASELINE.json: cfg 1 (512^2, patch 64,
step 32) and a cfg-3 overlap strip (4096 x 400, patch 120, step 20, batch 256)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np, torch
from bench import synth_pair
from sofima_amd import flow_field as ff

calc = ff.JAXMaskedXCorrWithStatsCalculator()
big_pre, big_post = synth_pair(4096, 77)
for tag, sl, patch, step, batch in (('cfg 1: 512^2, patch 64, step 32', (slice(0, 512), slice(0, 512)), 64, 32, 1024),
                                    ('cfg 3 strip: 4096 x 400, patch 120, step 20', (slice(0, 4096), slice(0, 400)), 120, 20, 256),
                                    ('2048^2, patch 96, step 24', (slice(0, 2048), slice(0, 2048)), 96, 24, 1024)):
  a = torch.from_numpy(np.ascontiguousarray(big_pre[sl])).cuda()
  b = torch.from_numpy(np.ascontiguousarray(big_post[sl])).cuda()
  res = {}
  for mode in ('full', 'pruned', 'full', 'pruned'):  # (alternating: the clocks ramp up during the first runs)
    if mode == 'full':
      os.environ['SFM_MFMA_PRUNE'] = '0'
    else:
      os.environ.pop('SFM_MFMA_PRUNE', None)
    calc.flow_field(a, b, patch, step, batch_size=batch); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
      f = calc.flow_field(a, b, patch, step, batch_size=batch)
    torch.cuda.synchronize()
    res[mode] = ((time.perf_counter() - t0) / 30 * 1e3, np.asarray(f))
  print(f'{tag}: pruned {res["pruned"][0]:.3f} ms, un-pruned {res["full"][0]:.3f} ms, '
        f'identical {np.array_equal(res["pruned"][1], res["full"][1], equal_nan=True)}', flush=True)
