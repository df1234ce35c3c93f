"""Smallest possible run of the pipeline kernel (a hang shows here first)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
from sofima_amd import _abi, flow_field as ff
from tests.test_gpu_flow import _prune_images
pre, post = _prune_images('em', 31, 460, 500)
rng = np.random.default_rng(5)
for b in (1, 2, 3, 8, 40):
  starts = np.stack([rng.integers(-10, 300 + 10, b), rng.integers(-10, 340 + 10, b)], axis=1)
  kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5, post_patch_size=(160, 160),
            post_starts=starts)
  args = (pre, post, None, None, (160, 160), starts, None)
  ref = ff.batched_xcorr_peaks(*args, method=2, **kw)
  for grid in (None, 2):
    with _abi.option('SFM_MFMA_PIPE', 1):
      if grid:
        with _abi.option('SFM_MFMA_GRID', grid):
          got = ff.batched_xcorr_peaks(*args, method=2, **kw)
      else:
        got = ff.batched_xcorr_peaks(*args, method=2, **kw)
    print('batch', b, 'grid', grid, 'identical', np.array_equal(ref, got, equal_nan=True), flush=True)
