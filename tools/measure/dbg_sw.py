import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from scipy import ndimage
from sofima_amd import flow_field
from oracle import flow_oracle
py, px, qy, qx = 192, 176, 160, 144
rng = np.random.default_rng(py * 1000 + px)
h, w = 520, 560
base = ndimage.gaussian_filter(rng.standard_normal((h + 8, w + 8)), 1.5)
base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
pre = base[4:4 + h, 4:4 + w].copy()
post = base[6:6 + h, 1:1 + w].copy()
post[::7, ::5] += 3
b = 21
starts = np.stack([rng.integers(-20, h - py + 30, b), rng.integers(-20, w - px + 30, b)], axis=1)
post_starts = starts + np.array([(py - qy) // 2, (px - qx) // 2]) + rng.integers(-9, 10, (b, 2))
kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5, post_patch_size=(qy, qx), post_starts=post_starts)
for mean in (None, 117.5):
  ref = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), starts, mean, method=1, **kw)
  got = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), starts, mean, method=2, **kw)
  fft = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), starts, mean, method=3, **kw)
  want = flow_oracle.batched_xcorr_peaks(pre, post, None, None, (py, px), starts, mean, 2, 0.5, 5, (qy, qx), post_starts, workers=4)
  for i in range(b):
    if not (np.array_equal(np.isnan(got[i]), np.isnan(want[i])) and np.array_equal(got[i, :2], want[i, :2], equal_nan=True)):
      print('mean', mean, 'patch', i, 'starts', starts[i], post_starts[i], '\n  mfma', got[i], '\n  direct', ref[i], '\n  fft', fft[i], '\n  oracle', want[i])
print('done')
from sofima_amd import _abi
def run(sel, mean=None, **opts):
  kw2 = dict(kw); kw2['post_starts'] = post_starts[sel]
  return flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), starts[sel], mean, method=2, **kw2)
print('patch 3 alone', run([3]))
print('patches 3,3,3', run([3, 3, 3]))
print('patches 0..5', run(list(range(6)))[3])
for g_ in (1, 2):
  with _abi.option('SFM_MFMA_GRID', g_):
    print('grid', g_, run(list(range(21)))[3])
# the start that is clamped: un-clamped by hand
st2 = starts.copy(); st2[3, 0] = 328
kw3 = dict(kw)
print('start given clamped', flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), st2, None, method=2, **kw3)[3])
for d_ in (-3, -1, 1, 2):
  st3 = starts.copy(); st3[3, 0] = 328 + d_ if 328 + d_ <= 328 else 328; st3[3, 1] += d_
  ps3 = post_starts.copy(); ps3[3, 1] += d_
  kw4 = dict(kw); kw4['post_starts'] = ps3
  g2 = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), st3, None, method=2, **kw4)[3]
  g1 = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), st3, None, method=1, **kw4)[3]
  print('x shifted by', d_, g2, g1)
