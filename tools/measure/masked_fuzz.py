"""Randomised hunt for a difference between the masked path with and without the
overlap-rule skips (SFM_MASKED_DEADROWS); argv[1] = seconds to run."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
from scipy import ndimage
from sofima_amd import flow_field

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(time.time()))
t_end = time.time() + budget
n = bad = 0
while time.time() < t_end:
  py = int(rng.choice([160, 128, 96, 64, 48, 120, 50]))
  px = py if rng.random() < 0.7 else int(rng.choice([160, 96, 64, 70]))
  same = rng.random() < 0.8
  qy, qx = (py, px) if same else (int(rng.integers(py // 2, py + 1)), int(rng.integers(px // 2, px + 1)))
  b = int(rng.integers(1, 24))
  base = ndimage.gaussian_filter(rng.standard_normal((b, py + 8, px + 8)), (0, 1.5, 1.5))
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  prev = base[:, 2:2 + py, 3:3 + px].copy()
  oy, ox = (py - qy) // 2, (px - qx) // 2
  curr = base[:, 4 + oy:4 + oy + qy, 1 + ox:1 + ox + qx].copy()
  dens = float(rng.choice([0.0, 0.001, 0.02, 0.3, 0.7]))
  pm = rng.random(prev.shape) < dens
  cm = rng.random(curr.shape) < float(rng.choice([0.0, 0.01, 0.2]))
  if rng.random() < 0.5:
    pm[::2] = False          # clean patches in between
    cm[::2] = False
  if rng.random() < 0.3:
    pm[:, : py // int(rng.integers(2, 6))] = True
  if not pm.any() and not cm.any():
    pm[0, 0, 0] = True
  os.environ.pop('SFM_MASKED_DEADROWS', None)
  a = flow_field.masked_xcorr(prev, curr, pm, cm, mean=None)
  os.environ['SFM_MASKED_DEADROWS'] = '0'
  c = flow_field.masked_xcorr(prev, curr, pm, cm, mean=None)
  n += 1
  if not np.array_equal(a, c, equal_nan=True):
    bad += 1
    print('MISMATCH', dict(patch=(py, px), post=(qy, qx), b=b, dens=dens), flush=True)
print(f'{n} random masked batches, {bad} mismatches')
