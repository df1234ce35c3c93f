"""Masked flow (mask_only_for_patch_selection=False) on the bench workload:
blob masks (5 % of the area in discs: most patches have no masked pixel)
and salt-and-pepper masks (every patch has masked pixels)."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import flow_field
from bench import synth_pair


def timed(fn, n=2):
  fn(); torch.cuda.synchronize()
  t = time.perf_counter()
  for _ in range(n): r = fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t) / n, r


def blobs(shape, frac, radius, seed):
  rng = np.random.default_rng(seed)
  m = np.zeros(shape, bool)
  n = int(frac * shape[0] * shape[1] / (np.pi * radius ** 2))
  yy, xx = np.mgrid[-radius:radius + 1, -radius:radius + 1]
  disc = yy ** 2 + xx ** 2 <= radius ** 2
  for _ in range(n):
    y, x = rng.integers(radius, shape[0] - radius), rng.integers(radius, shape[1] - radius)
    m[y - radius:y + radius + 1, x - radius:x + radius + 1] |= disc
  return m


side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
pre, post = synth_pair(side, 5)
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
dt0, _ = timed(lambda: calc.flow_field(a, b, 160, 40, batch_size=1024))
print('unmasked %d^2: %.1f ms' % (side, dt0 * 1e3))
rng = np.random.default_rng(0)
cases = [('blobs r=130, 5 %% of the area', blobs(pre.shape, 0.05, 130, 1), blobs(pre.shape, 0.05, 130, 2)),
         ('blobs r=40, 5 %% of the area', blobs(pre.shape, 0.05, 40, 3), blobs(pre.shape, 0.05, 40, 4)),
         ('salt and pepper 5 %', rng.random(pre.shape) < 0.05, rng.random(pre.shape) < 0.05)]
only = os.environ.get('MASK_CASE')
for name, pm, qm in cases:
  if only and not name.startswith(only): continue
  # fraction of patches with a masked pixel on either side
  cnt = flow_field._masked_counts(pm, (160, 160), (40, 40)) + flow_field._masked_counts(qm, (160, 160), (40, 40))
  dirty = float((np.asarray(cnt) > 0).mean())
  pmt = torch.from_numpy(pm).cuda(); qmt = torch.from_numpy(qm).cuda()
  for env in ((None,) if only else (None, '0')):
    if env is None: os.environ.pop('SFM_MASKED_FAST', None)
    else: os.environ['SFM_MASKED_FAST'] = env
    dt, f = timed(lambda: calc.flow_field(a, b, 160, 40, pre_mask=pmt, post_mask=qmt, batch_size=1024))
    print('%s (%.0f %% of the patches dirty) %s: %.1f ms = %.1fx unmasked, masked area %.1f %%' % (
        name, 100 * dirty, 'eight-pass form' if env else 'fast form', dt * 1e3, dt / dt0, 100 * pm.mean()))
os.environ.pop('SFM_MASKED_FAST', None)
