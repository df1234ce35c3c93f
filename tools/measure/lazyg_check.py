"""LAZYG=1 vs 0: peak statistics of random patch batches must agree bit for bit."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
from sofima_amd import _abi, flow_field
from tests.test_gpu_flow import _em_pair
pre, post = _em_pair(3, 500, 540, shift=(3, -5), warp=2.0)
rng = np.random.default_rng(1)
for (py, px) in ((160, 160), (80, 80), (96, 96), (64, 64), (48, 48), (128, 112), (32, 48)):
  b = 40
  starts = np.stack([rng.integers(0, 500 - py, b), rng.integers(0, 540 - px, b)], axis=1)
  kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5, post_patch_size=(py, px), post_starts=starts)
  for mean in (None, 100.0):
    args = (pre, post, None, None, (py, px), starts, mean)
    got = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
    with _abi.option('SFM_MFMA_LAZYG', 0):
      want = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
    same = np.array_equal(got, want, equal_nan=True)
    print((py, px), mean, 'identical' if same else 'DIFFERENT')
    if not same:
      bad = np.nonzero(~np.all((got == want) | (np.isnan(got) & np.isnan(want)), axis=1))[0]
      print('  rows', bad[:8], '\n  got', got[bad[:4]], '\n  want', want[bad[:4]])
