"""Image warping by coordinate maps on MI355X.

Drop-in for `warp.warp_subvolume` of the reference (warp.py:58-186), the
rendering step that follows mesh relaxation.  Every
section is ONE kernel (`sfm_warp_section`): linear interpolation of the map
nodes to the output pixels (scipy RegularGridInterpolator in the reference),
conversion to OpenCV's 1/32-pixel fixed-point map format and the `cv2.remap`
resampling are fused, so the two dense float maps per section never exist.

Parity note: OpenCV is not installable in the build container, so the
resampling semantics (convertMaps rounding, the 32 x 32-phase weight tables of
initInterTab2D with 15-bit fixed point for 8-bit images, constant zero border)
are restated from OpenCV's published algorithm and pinned by the reference's
own tests (tests/warp_test.py:27-82) only: "parity unpinned" beyond those.
`ndimage_warp` (warp.py:189-335), the SciPy-only rendering path, is built as
well and IS pinned: the reference function runs through the golden shim and
the kernel reproduces its output bit for bit (tests/golden/ndimage_warp.npz).
render_tiles and warp_points are host-side utilities and out of scope.
"""
from __future__ import annotations

import ctypes as C
import functools

import numpy as np
import torch

from . import _abi
from . import _dev

_INTER = {'nearest': 0, 'linear': 1, 'cubic': 2, 'lanczos': 3}
_TAB = 32            # INTER_TAB_SIZE
_COEF_SCALE = 1 << 15  # INTER_REMAP_COEF_SCALE


def _coeffs_1d(kind: int) -> np.ndarray:
  """[32, ksize] float32 tap weights per sub-pixel phase."""
  x = np.arange(_TAB, dtype=np.float32) / np.float32(_TAB)
  if kind == 1:
    return np.stack([1 - x, x], axis=1).astype(np.float32)
  if kind == 2:
    a = np.float32(-0.75)
    c0 = ((a * (x + 1) - 5 * a) * (x + 1) + 8 * a) * (x + 1) - 4 * a
    c1 = ((a + 2) * x - (a + 3)) * x * x + 1
    c2 = ((a + 2) * (1 - x) - (a + 3)) * (1 - x) * (1 - x) + 1
    return np.stack([c0, c1, c2, 1 - c0 - c1 - c2], axis=1).astype(np.float32)
  # Lanczos, a = 4: sin(pi t) sin(pi t / 4) / t^2 through the angle-addition
  # table OpenCV uses for the eight taps
  s45 = 0.70710678118654752440084436210485
  cs = np.array([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45],
                 [0, -1], [-s45, s45]])
  out = np.zeros((_TAB, 8), np.float32)
  for i, xv in enumerate(x):
    if xv < np.finfo(np.float32).eps:
      out[i, 3] = 1
      continue
    y0 = -(float(xv) + 3) * np.pi * 0.25
    s0, c0 = np.sin(y0), np.cos(y0)
    y = -(float(xv) + 3 - np.arange(8)) * np.pi * 0.25
    c = ((cs[:, 0] * s0 + cs[:, 1] * c0) / (y * y)).astype(np.float32)
    out[i] = c * (np.float32(1) / c.sum(dtype=np.float32))
  return out


@functools.lru_cache(maxsize=None)
def _inter_tab(kind: int, fixed: bool) -> np.ndarray:
  """[32 * 32, ksize * ksize] 2-d weights (phase = 32 * fy + fx); int16 with 15
  fractional bits when `fixed` (8-bit images), rounding error folded into one
  tap the way OpenCV does it."""
  c = _coeffs_1d(kind)
  ks = c.shape[1]
  tab = (c[:, None, :, None] * c[None, :, None, :]).reshape(_TAB * _TAB, ks * ks)
  tab = tab.astype(np.float32)
  if not fixed:
    return np.ascontiguousarray(tab)
  it = np.clip(np.rint(tab * np.float32(_COEF_SCALE)), -32768, 32767).astype(np.int32)
  # OpenCV's initInterTab2D, statement for statement: the rounding error of a
  # phase goes to the largest (deficit) / smallest (excess) of the taps
  # (k1, k2) in [ks/2, ks/2 + 2)^2, found with strict compares starting from
  # (ks/2, ks/2), and the corrected tap is cast to short.  The tables of all
  # phases are one array filled phase by phase, and for the 2 x 2 bilinear
  # kernel that index range runs past the phase's own taps into the (still
  # zero) taps of the following phases: the scan sees those zeros, and a
  # correction written there is overwritten when that phase is filled.
  kk = ks * ks
  flat = np.zeros((_TAB * _TAB + 4) * kk, np.int64)
  h = ks // 2
  scan = [k1 * ks + k2 for k1 in (h, h + 1) for k2 in (h, h + 1)]
  for e in range(_TAB * _TAB):
    base = e * kk
    flat[base:base + kk] = it[e]
    diff = int(it[e].sum()) - _COEF_SCALE
    if diff:
      big = small = base + scan[0]
      for off in scan:
        v = flat[base + off]
        if v < flat[small]:
          small = base + off
        elif v > flat[big]:
          big = base + off
      k = big if diff < 0 else small
      flat[k] = ((int(flat[k]) - diff + 32768) & 0xffff) - 32768   # (short)
  it = flat[:_TAB * _TAB * kk].reshape(_TAB * _TAB, kk)
  return np.ascontiguousarray(it.astype(np.int16))


def _box(b):
  """(start xyz, size xyz) of a bounding box object or (start, size) pair."""
  if hasattr(b, 'start') and hasattr(b, 'size'):
    return np.asarray(b.start), np.asarray(b.size)
  return np.asarray(b[0]), np.asarray(b[1])


def _make_contiguous(image: np.ndarray):
  """uint64 segment ids -> dense int32 ids (0 stays 0) and the inverse table."""
  ids, inv = np.unique(image, return_inverse=True)
  if ids[0] != 0:
    ids = np.concatenate([[0], ids]).astype(np.uint64)
    inv = inv + 1
  assert len(ids) < 2**31
  return inv.reshape(image.shape).astype(np.int32), ids


def warp_subvolume(image: np.ndarray, image_box, coord_map: np.ndarray, map_box,
                   stride: float, out_box, interpolation: str | None = None,
                   offset: float = 0.0, parallelism: int = 1) -> np.ndarray:
  """Warps a subvolume of data according to a coordinate map (warp.py:58-186).

  image: [n, z, y, x] uint8 / uint16 / float32 data, or uint64 segmentation
  (nearest neighbour on contiguous ids, mapped back afterwards); coord_map:
  [2, z, y, x] xy 'inverse' map in relative format; boxes: objects with
  `.start` / `.size` in xyz (or (start, size) pairs); `stride`: image pixels
  per map node.  Sections whose map is all NaN are skipped (left zero).
  `parallelism` is accepted for compatibility: sections are enqueued back to
  back on the GPU.
  """
  del parallelism
  dev = _dev.device()
  image = np.asarray(image)
  ids = None
  orig_dtype = image.dtype
  if image.dtype == np.uint64:
    kind = 0
    image, ids = _make_contiguous(image)
    dtype = _abi.DTYPE_I32
  else:
    kind = 3 if interpolation is None else _INTER[interpolation]
    if image.dtype == np.uint32:
      if image.max() >= 2**16:
        raise ValueError(
            'Image warping supported up to uint16 only. For segmentation data, '
            'use uint64.')
      image = image.astype(np.uint16)
    if image.dtype == np.uint8:
      dtype = _abi.DTYPE_U8
    elif image.dtype == np.uint16:
      dtype = _abi.DTYPE_U16
    else:
      image = image.astype(np.float32, copy=False)
      dtype = _abi.DTYPE_F32
  img_start, _ = _box(image_box)
  map_start, _ = _box(map_box)
  out_start, out_size = _box(out_box)
  coord_map = np.asarray(coord_map)
  skipped = np.all(np.isnan(coord_map), axis=(0, 2, 3))

  if coord_map.shape[1] < image.shape[1] or int(out_size[2]) < image.shape[1]:
    # the reference indexes abs_map[:, z] and warped[:, z] for every z of image
    raise ValueError(
        f'z extents disagree: image {image.shape[1]}, coord_map {coord_map.shape[1]}, '
        f'out_box {int(out_size[2])}')

  # absolute source coordinates in the local frame of `image`
  # (map_utils.to_absolute + the box shift, warp.py:125-128): the reference
  # adds both in place, i.e. rounds twice to the MAP's dtype, and interpolates
  # the dense coordinates from those values -- same here, float32 or float64
  my, mx = coord_map.shape[2:]
  hy, hx = np.mgrid[:my, :mx]
  shift = map_start[:2] * stride - img_start[:2] + offset
  map_dtype = np.float64 if coord_map.dtype == np.float64 else np.float32
  abs_map = coord_map.astype(map_dtype, copy=True)
  abs_map[0] += hx * stride
  abs_map[1] += hy * stride
  abs_map += np.asarray(shift, np.float64).reshape(2, 1, 1)[:, None]
  map_t = torch.from_numpy(abs_map).to(dev)

  img_t = torch.from_numpy(np.ascontiguousarray(image).view(
      np.int16 if image.dtype == np.uint16 else image.dtype)).to(dev)
  out_t = torch.zeros((image.shape[0], int(out_size[2]), int(out_size[1]),
                       int(out_size[0])), dtype=img_t.dtype, device=dev)
  d = _abi.SfmWarpDesc()
  d.dtype = dtype
  d.interpolation = _abi.WARP_NEAREST if kind == 0 else _abi.WARP_TABLE
  tab_t = None
  if kind != 0:
    tab = _inter_tab(kind, dtype == _abi.DTYPE_U8)
    tab_t = torch.from_numpy(tab).to(dev)
    d.ksize = int(round(np.sqrt(tab.shape[1])))
    d.weights = tab_t.data_ptr()
  d.image_shape = (C.c_int32 * 2)(*image.shape[2:])
  d.map_shape = (C.c_int32 * 2)(my, mx)
  d.out_shape = (C.c_int32 * 2)(int(out_size[1]), int(out_size[0]))
  d.map_origin = (C.c_double * 2)(
      float(map_start[1] * stride - out_start[1] + offset),
      float(map_start[0] * stride - out_start[0] + offset))
  d.stride = float(stride)
  d.coord_map_f64 = 1 if map_dtype == np.float64 else 0
  d.stream = _dev.stream_ptr()
  lib = _abi.load()
  for z in range(image.shape[1]):
    if skipped[z]:
      continue
    zmap = map_t[:, z].contiguous()   # kept alive until the launches are enqueued
    d.coord_map = zmap.data_ptr()
    for c in range(image.shape[0]):
      d.image = img_t[c, z].data_ptr()
      d.out = out_t[c, z].data_ptr()
      _abi.check(lib.sfm_warp_section(C.byref(d)))
  warped = out_t.cpu().numpy()
  if ids is not None:
    return ids[warped]
  if orig_dtype == np.uint16 or image.dtype == np.uint16:
    warped = warped.view(np.uint16)
  return warped.astype(orig_dtype)


def ndimage_warp(image: np.ndarray, coord_map: np.ndarray, stride, work_size, overlap,
                 order=1, map_coordinates=None, image_box=None, map_box=None, out_box=None,
                 parallelism: int = 1, out_scale=(1.0, 1.0, 1.0)) -> np.ndarray:
  """Warps a subvolume of data using map_coordinates semantics (warp.py:189-335).

  image: [z, ] y, x data (uint8 / uint16 / float32); coord_map: [N, [z,] y, x]
  relative coordinate map; stride: [z,] y, x image voxels per map node;
  image_box / map_box / out_box: objects with `.start` / `.size` in xyz (or
  (start, size) pairs); out_scale: xy[z] out_voxel / source_voxel.

  One kernel computes, per output voxel, what the reference computes with two
  rounds of scipy.ndimage.map_coordinates per work box (dense coordinates by
  order-1 interpolation of the absolute map, then order-`order` sampling of the
  image), in double and in SciPy's operation order -- bit for bit the
  reference's output (tests/golden/ndimage_warp.npz).  `work_size`, `overlap`
  and `parallelism` only shape the reference's host loop and do not change the
  result for orders 0 and 1; they are validated and otherwise unused.  Built:
  order 0 and 1 with the default `map_coordinates`; higher spline orders,
  custom samplers and uint64 label volumes raise NotImplementedError (use
  `warp_subvolume` for label volumes).
  """
  del parallelism
  image = np.asarray(image)
  coord_map = np.asarray(coord_map)
  shape = coord_map.shape[1:]
  dim = len(shape)
  assert dim == len(stride)
  assert dim == len(overlap)
  assert dim == len(work_size)
  if dim != image.ndim:
    raise ValueError(f'Dimension mismatch: image: {image.ndim} vs coord map: {dim}')
  if map_coordinates is not None:
    raise NotImplementedError('ndimage_warp: custom map_coordinates callables run on the host only')
  if image.dtype == np.uint64:
    raise NotImplementedError('ndimage_warp: uint64 label volumes (use warp_subvolume)')
  if order not in (0, 1):
    raise NotImplementedError(f'ndimage_warp: interpolation order {order} (0 and 1 are built)')
  if dim not in (2, 3):
    raise ValueError(f'ndimage_warp: {dim}-d data')
  if map_box is not None and image_box is None:
    raise ValueError('image_box has to be specified when map_box is used.')

  # absolute source map, operation for operation (map_utils.to_absolute adds in
  # place in the map's dtype; the out_scale product is float64): warp.py:250-265
  src = np.array(coord_map, copy=True)
  idx = np.mgrid[tuple(slice(0, s) for s in shape)]
  off_zyx = [h * st for h, st in zip(idx, stride)]
  for i in range(dim):
    src[i, ...] += off_zyx[-(i + 1)]
  if map_box is not None:
    map_start, _ = _box(map_box)
    img_start, _ = _box(image_box)
    src += (map_start[:dim] * np.asarray(stride)[::-1] -
            img_start[:dim] / np.asarray(out_scale)[:dim]).reshape((dim,) + (1,) * dim)
  reshaper = (slice(None),) + (np.newaxis,) * dim
  src = np.ascontiguousarray(src.copy() * np.array(out_scale[:dim])[reshaper], dtype=np.float64)

  if out_box is not None:
    out_start, out_size = _box(out_box)
    out_shape = tuple(int(v) for v in np.asarray(out_size)[::-1][-dim:])
  else:
    out_start = np.zeros(3, np.int64)
    out_shape = image.shape
  if map_box is not None:
    map_start, _ = _box(map_box)
    offset = (map_start * np.asarray(stride)[::-1] - out_start)[::-1]
  else:
    offset = (0,) * dim

  if image.dtype == np.uint8:
    dtype, view = _abi.DTYPE_U8, np.uint8
  elif image.dtype == np.uint16:
    dtype, view = _abi.DTYPE_U16, np.int16
  elif image.dtype == np.float32:
    dtype, view = _abi.DTYPE_F32, np.float32
  else:
    raise NotImplementedError(f'ndimage_warp: {image.dtype} images (uint8, uint16, float32)')
  dev = _dev.device()
  img_t = torch.from_numpy(np.ascontiguousarray(image).view(view)).to(dev)
  map_t = torch.from_numpy(src).to(dev)
  out_t = torch.empty(out_shape, dtype=img_t.dtype, device=dev)
  pad = (1,) * (3 - dim)
  d = _abi.SfmNdWarpDesc()
  d.ndim = dim
  d.dtype = dtype
  d.order = int(order)
  d.image_shape = (C.c_int32 * 3)(*(pad + tuple(image.shape)))
  d.map_shape = (C.c_int32 * 3)(*(pad + tuple(shape)))
  d.out_shape = (C.c_int32 * 3)(*(pad + tuple(out_shape)))
  d.stride = (C.c_double * 3)(*((1.0,) * (3 - dim) + tuple(float(v) for v in stride)))
  d.offset = (C.c_double * 3)(*((0.0,) * (3 - dim) + tuple(float(v) for v in offset[-dim:])))
  d.image = img_t.data_ptr()
  d.src_map = map_t.data_ptr()
  d.out = out_t.data_ptr()
  d.stream = _dev.stream_ptr()
  _abi.check(_abi.load().sfm_ndimage_warp(C.byref(d)))
  warped = out_t.cpu().numpy()
  if image.dtype == np.uint16:
    warped = warped.view(np.uint16)
  return warped.astype(image.dtype)
