"""Wall-clock of the stitching callers on a synthetic montage (host + device),
to be compared with the kernel time of a rocprofv3 trace of the same script."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scipy import ndimage
from sofima_amd import stitch_rigid, stitch_elastic, flow_utils

g, t, ov = (int(v) for v in (sys.argv[1:4] + ['3', '2048', '256'][len(sys.argv) - 1:]))
rng = np.random.default_rng(5)
side = g * (t - ov) + ov + 64
canvas = ndimage.gaussian_filter(rng.standard_normal((side, side), dtype=np.float32), 2.0)
canvas = ((canvas - canvas.min()) / (canvas.max() - canvas.min()) * 255).astype(np.uint8)
tile_map = {}
for ty in range(g):
  for tx in range(g):
    dy, dx = rng.integers(-6, 7, 2)
    y0 = 32 + ty * (t - ov) + dy
    x0 = 32 + tx * (t - ov) + dx
    tile_map[(tx, ty)] = np.ascontiguousarray(canvas[y0:y0 + t, x0:x0 + t])


def timed(name, fn, reps=2):
  fn(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps): r = fn()
  torch.cuda.synchronize()
  print('%-28s %8.1f ms' % (name, (time.perf_counter() - t0) / reps * 1e3), flush=True)
  return r


cx, cy = timed('compute_coarse_offsets', lambda: stitch_rigid.compute_coarse_offsets(
    (g, g), tile_map, overlaps_xy=((ov - 56, ov + 44), (ov - 56, ov + 44)), min_overlap=ov - 100))
coarse = timed('optimize_coarse_mesh', lambda: stitch_rigid.optimize_coarse_mesh(cx, cy))
print('   pairs: %d x, %d y; finite offsets %d / %d' % (
    g * (g - 1), g * (g - 1), np.isfinite(cx[0]).sum() + np.isfinite(cy[0]).sum(), 2 * g * (g - 1)))
fx = timed('compute_flow_map x', lambda: stitch_elastic.compute_flow_map(
    tile_map, cx[:, 0], 0, patch_size=(120, 120), stride=(20, 20), batch_size=256))
fy = timed('compute_flow_map y', lambda: stitch_elastic.compute_flow_map(
    tile_map, cy[:, 0], 1, patch_size=(120, 120), stride=(20, 20), batch_size=256))
f0 = next(iter(fx[0].values()))
timed('clean_flow (one pair)', lambda: flow_utils.clean_flow(f0[:, np.newaxis], 1.4, 1.4, 0, 0))
