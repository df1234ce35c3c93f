"""Correlation kernel with and without exact tile pruning (8192^2 bench pair)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
import torch
from bench import synth_pair
from sofima_amd import flow_field as ff

pre, post = synth_pair(8192, 1002)
dev = torch.device('cuda:0')
pre_d = torch.from_numpy(pre).to(dev)
post_d = torch.from_numpy(post).to(dev)
calc = ff.JAXMaskedXCorrWithStatsCalculator()
res = {}
for tag, env in (('pruned', None), ('full', '0'), ('pruned', None), ('full', '0')):
  if env is None:
    os.environ.pop('SFM_MFMA_PRUNE', None)
  else:
    os.environ['SFM_MFMA_PRUNE'] = env
  calc.flow_field(pre_d, post_d, 160, 40, batch_size=1024)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(3):
    f = calc.flow_field(pre_d, post_d, 160, 40, batch_size=1024)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 3
  print(f'{tag}: {dt * 1e3:.2f} ms per pair', flush=True)
  res.setdefault(tag, f)
a, b = np.asarray(res['pruned']), np.asarray(res['full'])
print('identical:', np.array_equal(a, b, equal_nan=True))
