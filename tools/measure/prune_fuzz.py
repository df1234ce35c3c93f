"""Randomised hunt for a pruned / un-pruned mismatch: random textures, shifts, noise,
patch geometries and peak parameters; argv[1] = seconds to run."""
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
  h, w = (int(v) for v in rng.integers(300, 700, 2))
  sig = float(rng.choice([0.8, 1.5, 2.0, 4.0, 10.0]))
  base = ndimage.gaussian_filter(rng.standard_normal((h + 200, w + 200)), sig)
  base = (base - base.min()) / (base.max() - base.min()) * 255
  dy, dx = (int(v) for v in rng.integers(-90, 91, 2)) if rng.random() < 0.3 else (int(v) for v in rng.integers(-8, 9, 2))
  pre = base[100:100 + h, 100:100 + w]
  post = base[100 + dy:100 + dy + h, 100 + dx:100 + dx + w] + rng.standard_normal((h, w)) * float(rng.choice([0, 2, 8, 30]))
  if rng.random() < 0.2:   # flat regions / saturated stripes
    pre = pre.copy(); pre[: h // 3] = 77
  if rng.random() < 0.2:
    post = post.copy(); post[:, ::int(rng.integers(5, 40))] = 255
  pre = np.clip(np.round(pre), 0, 255).astype(np.uint8)
  post = np.clip(np.round(post), 0, 255).astype(np.uint8)
  py = int(rng.choice([160, 160, 128, 96, 64, 48, 120, 100, 50, 33]))
  px = py if rng.random() < 0.7 else int(rng.choice([160, 128, 96, 64, 70, 112]))
  if py > h - 4 or px > w - 4:
    continue
  b = int(rng.integers(1, 40))
  starts = np.stack([rng.integers(-10, h - py + 10, b), rng.integers(-10, w - px + 10, b)], axis=1)
  kw = dict(min_distance=int(rng.choice([1, 2, 2, 3, 7])), threshold_rel=float(rng.choice([0.5, 0.5, 0.3, 0.8, 0.1])),
            peak_radius=int(rng.choice([5, 5, 2, 9, 20])), post_patch_size=(py, px), post_starts=starts)
  mean = None if rng.random() < 0.7 else float(rng.uniform(0, 255))
  args = (pre, post, None, None, (py, px), starts, mean)
  os.environ.pop('SFM_MFMA_PRUNE', None)
  a = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
  os.environ['SFM_MFMA_PRUNE'] = '0'
  c = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
  n += 1
  if not np.array_equal(a, c, equal_nan=True):
    bad += 1
    print('MISMATCH', dict(h=h, w=w, sig=sig, shift=(dy, dx), patch=(py, px), b=b, mean=mean, **{k: v for k, v in kw.items() if k not in ('post_starts',)}), flush=True)
print(f'{n} random cases, {bad} mismatches')
