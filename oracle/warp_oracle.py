"""CPU oracle for image warping.  TEST INFRASTRUCTURE.

A NumPy / SciPy restatement of `warp.warp_subvolume` (/root/reference/warp.py:
58-186): scipy.interpolate.RegularGridInterpolator for the dense coordinates
(the reference's own call), followed by a restatement of the two OpenCV calls
the reference makes (`cv2.convertMaps` to CV_16SC2, `cv2.remap` with a zero
constant border).  Never imported by anything under `sofima_amd/`.

PARITY UNPINNED beyond the reference's known-answer tests: OpenCV (cv2) cannot
be installed in the build container, so neither this oracle nor the HIP path
could be compared with the reference's output; the fixed-point semantics below
follow OpenCV's published implementation (imgwarp.cpp: initInterTab2D,
remapNearest / remapBilinear / remapLanczos4, 5 fractional map bits, 15-bit
weights for 8-bit images).  Pinned by tests/warp_test.py:27-82 (re-typed in
tests/test_reference_kats.py / tests/test_gpu_warp.py).
"""
from __future__ import annotations

import numpy as np
from scipy import interpolate

TAB = 32
SCALE = 1 << 15


def _lanczos4(x):
  if x < np.finfo(np.float32).eps:
    c = np.zeros(8, np.float32)
    c[3] = 1
    return c
  s45 = 0.70710678118654752440084436210485
  cs = [(1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1),
        (-s45, s45)]
  y0 = -(x + 3) * np.pi * 0.25
  s0, c0 = np.sin(y0), np.cos(y0)
  c = np.zeros(8, np.float32)
  for i in range(8):
    y = -(x + 3 - i) * np.pi * 0.25
    c[i] = np.float32((cs[i][0] * s0 + cs[i][1] * c0) / (y * y))
  return c * (np.float32(1) / np.sum(c, dtype=np.float32))


def _taps(kind, x):
  x = np.float32(x)
  if kind == 'linear':
    return np.array([1 - x, x], np.float32)
  if kind == 'cubic':
    a = np.float32(-0.75)
    c = np.zeros(4, np.float32)
    c[0] = ((a * (x + 1) - 5 * a) * (x + 1) + 8 * a) * (x + 1) - 4 * a
    c[1] = ((a + 2) * x - (a + 3)) * x * x + 1
    c[2] = ((a + 2) * (1 - x) - (a + 3)) * (1 - x) * (1 - x) + 1
    c[3] = 1 - c[0] - c[1] - c[2]
    return c
  return _lanczos4(float(x))


def weight_table(kind, fixed):
  """[32, 32, ks, ks] weights; int32 holding OpenCV's int16 table when `fixed`.

  imgwarp.cpp, initInterTab2D: every product is saturate_cast<short>(v * 2^15);
  when the taps of a phase do not sum to 2^15 the difference goes to the
  largest / smallest of itab[k1 * ks + k2], k1, k2 in {ks/2, ks/2 + 1}, strict
  compares in scan order starting from (ks/2, ks/2), result cast to short.  The
  phases are consecutive in one static array that is filled in order, so for
  ks = 2 the scan reads taps of the next phases (zero at that time) and a
  correction stored there is lost when that phase is written."""
  one = [_taps(kind, i / TAB) for i in range(TAB)]
  ks = len(one[0])
  if not fixed:
    out = np.zeros((TAB, TAB, ks, ks), np.float32)
    for i in range(TAB):
      for j in range(TAB):
        out[i, j] = np.outer(one[i], one[j]).astype(np.float32)
    return out
  mem = [0] * ((TAB * TAB + 4) * ks * ks)
  half = ks // 2
  for i in range(TAB):
    for j in range(TAB):
      at = (i * TAB + j) * ks * ks
      total = 0
      for k1 in range(ks):
        for k2 in range(ks):
          v = np.float32(one[i][k1] * one[j][k2]) * np.float32(SCALE)
          q = int(min(max(np.rint(v), -32768), 32767))
          mem[at + k1 * ks + k2] = q
          total += q
      if total != SCALE:
        diff = total - SCALE
        hi = lo = at + half * ks + half
        for k1 in range(half, half + 2):
          for k2 in range(half, half + 2):
            cur = at + k1 * ks + k2
            if mem[cur] < mem[lo]:
              lo = cur
            elif mem[cur] > mem[hi]:
              hi = cur
        dst = hi if diff < 0 else lo
        mem[dst] = (mem[dst] - diff + 32768) % 65536 - 32768
  return np.array(mem[:TAB * TAB * ks * ks], np.int32).reshape(TAB, TAB, ks, ks)


def _cv_round(v):
  v = np.asarray(v, np.float64)
  bad = ~((v > -2147483648.0) & (v < 2147483647.0))
  r = np.rint(np.where(bad, 0, v)).astype(np.int64)
  return np.where(bad, -2147483648, r)


def remap(img, dx, dy, kind):
  """cv2.remap(img, *cv2.convertMaps(dx, dy, CV_16SC2, nn), interpolation)."""
  h, w = img.shape

  def px(y, x):
    ok = (y >= 0) & (y < h) & (x >= 0) & (x < w)
    return np.where(ok, img[np.clip(y, 0, h - 1), np.clip(x, 0, w - 1)], 0)

  if kind == 'nearest':
    x = np.clip(_cv_round(dx), -32768, 32767)
    y = np.clip(_cv_round(dy), -32768, 32767)
    return px(y, x).astype(img.dtype)
  fx = _cv_round(dx.astype(np.float64) * TAB)
  fy = _cv_round(dy.astype(np.float64) * TAB)
  x0 = np.clip(fx >> 5, -32768, 32767)
  y0 = np.clip(fy >> 5, -32768, 32767)
  fixed = img.dtype == np.uint8
  tab = weight_table(kind, fixed)[fy & 31, fx & 31]   # [oy, ox, ks, ks]
  ks = tab.shape[-1]
  ofs = ks // 2 - 1
  if fixed:
    acc = np.zeros(dx.shape, np.int64)
    for k1 in range(ks):
      for k2 in range(ks):
        acc += tab[..., k1, k2].astype(np.int64) * px(y0 + k1 - ofs, x0 + k2 - ofs)
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
  acc = np.zeros(dx.shape, np.float32)
  for k1 in range(ks):
    for k2 in range(ks):
      acc = acc + tab[..., k1, k2] * px(y0 + k1 - ofs, x0 + k2 - ofs).astype(np.float32)
  if img.dtype == np.uint16:
    return np.clip(_cv_round(acc), 0, 65535).astype(np.uint16)
  return acc.astype(img.dtype)


def warp_subvolume(image, image_box, coord_map, map_box, stride, out_box,
                   interpolation=None, offset=0.0):
  """Boxes are (start xyz, size xyz) pairs."""
  image = np.asarray(image)
  ids = None
  if image.dtype == np.uint64:
    kind = 'nearest'
    ids, inv = np.unique(image, return_inverse=True)
    if ids[0] != 0:
      ids = np.concatenate([[0], ids]).astype(np.uint64)
      inv = inv + 1
    image = inv.reshape(image.shape).astype(np.int32)
  else:
    kind = 'lanczos' if interpolation is None else interpolation
  img_start, map_start = np.asarray(image_box[0]), np.asarray(map_box[0])
  out_start, out_size = np.asarray(out_box[0]), np.asarray(out_box[1])
  skipped = np.all(np.isnan(coord_map), axis=(0, 2, 3))
  my, mx = coord_map.shape[2:]
  hy, hx = np.mgrid[:my, :mx]
  # warp.py:125-128 / map_utils.py:169-186: both additions are made in place,
  # i.e. in the dtype of the map
  abs_map = np.array(coord_map, copy=True)
  if abs_map.dtype not in (np.float32, np.float64):
    abs_map = abs_map.astype(np.float32)
  abs_map[0] += hx[None] * stride
  abs_map[1] += hy[None] * stride
  abs_map += (map_start[:2] * stride - img_start[:2] + offset).reshape(2, 1, 1, 1)
  map_y = (np.arange(my) + map_start[1]) * stride - out_start[1] + offset
  map_x = (np.arange(mx) + map_start[0]) * stride - out_start[0] + offset
  out_y, out_x = np.mgrid[:out_size[1], :out_size[0]]
  warped = np.zeros((image.shape[0], out_size[2], out_size[1], out_size[0]),
                    image.dtype)
  for z in range(image.shape[1]):
    if skipped[z]:
      continue
    nodes = abs_map[:, z]
    dense = [interpolate.RegularGridInterpolator(
        (map_y, map_x), nodes[c], bounds_error=False, fill_value=None)(
            (out_y, out_x)).astype(np.float32) for c in (0, 1)]
    for c in range(image.shape[0]):
      warped[c, z] = remap(image[c, z], dense[0], dense[1], kind)
  return ids[warped] if ids is not None else warped


# ---------------------------------------------------------------------------
# warp.ndimage_warp (/root/reference/warp.py:189-335).  PINNED: the reference
# function itself runs through the stand-ins of tests/golden/_refshim (it needs
# only SciPy besides the stubbed imports) and its outputs are the fixtures of
# tests/golden/ndimage_warp.npz; this restatement is checked against them bit
# for bit (tests/test_oracle_golden.py).  scipy.ndimage.map_coordinates is the
# reference's own third-party call, used here as well.
# ---------------------------------------------------------------------------
def ndimage_abs_map(coord_map, stride, out_scale, map_start=None, image_start=None):
  """The float64 absolute source map of warp.py:250-265: map_utils.to_absolute
  (in-place float32 adds of float64 terms, map_utils.py:169-187), the optional
  box shift (float32 in place again) and the float64 product with out_scale."""
  dim = coord_map.shape[0]
  m = np.array(coord_map, dtype=np.float32)
  idx = np.mgrid[tuple(slice(0, s) for s in m.shape[1:])]
  off_zyx = [h * st for h, st in zip(idx, stride)]
  for i in range(dim):
    m[i, ...] += off_zyx[-(i + 1)]
  if map_start is not None:
    shift = (np.asarray(map_start)[:dim] * np.asarray(stride)[::-1] -
             np.asarray(image_start)[:dim] / np.asarray(out_scale)[:dim])
    m += shift.reshape((dim,) + (1,) * dim)
  reshaper = (slice(None),) + (np.newaxis,) * dim
  return m.copy() * np.array(out_scale[:dim])[reshaper]


def ndimage_warp(image, coord_map, stride, order=1, image_start=None, map_start=None,
                 out_start=None, out_size=None, out_scale=(1.0, 1.0, 1.0)):
  """One work box covering the whole output (the result does not depend on the
  work-box decomposition for order <= 1).  Boxes are xyz `start` / `size`
  vectors.  uint64 label volumes are not covered."""
  from scipy import ndimage
  dim = coord_map.shape[0]
  if dim != image.ndim:
    raise ValueError(f'Dimension mismatch: image: {image.ndim} vs coord map: {dim}')
  if map_start is not None and image_start is None:
    raise ValueError('image_box has to be specified when map_box is used.')
  src_map = ndimage_abs_map(coord_map, stride, out_scale, map_start, image_start)
  if out_size is not None:
    out_shape = tuple(int(v) for v in np.asarray(out_size)[::-1][-dim:])
    out_start = np.asarray(out_start)
  else:
    out_shape = image.shape
    out_start = np.zeros(3, np.int64)
  if map_start is not None:
    offset = (np.asarray(map_start) * np.asarray(stride)[::-1] - out_start)[::-1]
  else:
    offset = (0, 0, 0)
  src_coords = np.mgrid[tuple(slice(0, s) for s in out_shape)]
  src_coords = [(c - o) / s for c, s, o in zip(src_coords, stride, offset)]
  dense = [ndimage.map_coordinates(ev, src_coords, order=1) for ev in src_map[::-1]]
  return ndimage.map_coordinates(image, dense, order=order).astype(image.dtype)
