"""stitch_rigid._estimate_offset on one 4096 x 300 overlap (device-resident inputs)."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scipy import ndimage
from sofima_amd import stitch_rigid
rng = np.random.default_rng(2)
base = ndimage.gaussian_filter(rng.standard_normal((4200, 420)), 2.0)
base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
pre = np.ascontiguousarray(base[20:4116, 20:320]); post = np.ascontiguousarray(base[31:4127, 14:314])
for rep in range(3):
  torch.cuda.synchronize(); t = time.perf_counter()
  off, pr = stitch_rigid._estimate_offset(pre, post, 10.0, 7)
  torch.cuda.synchronize(); dt = time.perf_counter() - t
print('_estimate_offset 4096x300: %.2f ms, offset %s, peak ratio %.3f' % (dt * 1e3, off, pr))
