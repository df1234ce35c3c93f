"""Rigid tile stitching on MI355X: the callers of the two compute cores.

Drop-in for the reference's `sofima/stitch_rigid.py`.  The device work of that
module is (a) one whole-overlap masked cross-correlation per tile pair and
(b) the relaxation of a spring mesh whose nodes are tiles, with a custom
`mesh_force`; both run on the HIP path of this package:

  _estimate_offset            <-> stitch_rigid.py:39-67     range masks (sfm_range_mask)
                                                            + flow_field (patch = strip)
  compute_coarse_offsets      <-> stitch_rigid.py:104-273   host search over overlaps
  interpolate_missing_offsets <-> stitch_rigid.py:277-327   host
  elastic_tile_mesh[_3d]      <-> stitch_rigid.py:330-473   sfm_mesh_force, tile model
  optimize_coarse_mesh        <-> stitch_rigid.py:476-523   relax_mesh, native force

Offsets are (x, y) vectors in pixels; `cx` / `cy` are [2 | 3, z, y, x] arrays of
desired offsets between tile (x, y) and tile (x+1, y) / (x, y+1).
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Mapping, Sequence

import numpy as np
import torch

from . import _abi
from . import _dev
from . import flow_field
from . import mesh

TileXY = tuple[int, int]
MaskMap = Mapping[TileXY, np.ndarray]
Vector = tuple[int, int] | tuple[int, int, int] | tuple[int] | tuple[Any, ...]


def range_mask(img, range_limit: float, filter_size: int = 10,
               extra_mask=None) -> torch.Tensor:
  """Device mask of the pixels with too little local dynamic range
  (stitch_rigid.py:47-60), optionally OR-ed with `extra_mask`.

  Returns a uint8 CUDA tensor [y, x] (1 = masked) that `flow_field` accepts
  as `pre_mask` / `post_mask` without a host round trip.
  """
  dev = _dev.device()
  t, dtype = _dev.as_device_image(img, dev)
  if t.ndim != 2:
    raise ValueError('range masks are defined for 2-d images')
  d = _abi.SfmRangeMaskDesc()
  d.dtype = dtype
  d.shape = (C.c_int32 * 2)(int(t.shape[0]), int(t.shape[1]))
  d.filter_size = int(filter_size)
  # float images compare in float32 (NumPy's weak Python-scalar promotion)
  d.range_limit = float(np.float32(range_limit)) if dtype == _abi.DTYPE_F32 \
      else float(range_limit)
  d.image = t.data_ptr()
  extra = _dev.as_device_mask(extra_mask, dev)
  if extra is not None:
    if tuple(extra.shape) != tuple(t.shape):
      raise ValueError('mask and image shapes differ')
    d.extra_mask = extra.data_ptr()
  d.stream = _dev.stream_ptr()
  out = torch.empty(t.shape, dtype=torch.uint8, device=dev)
  _abi.check(_abi.load().sfm_range_mask(C.byref(d), out.data_ptr()))
  return out


def _estimate_offset(a: np.ndarray, b: np.ndarray, range_limit: float,
                     filter_size: int = 10,
                     masks: tuple[np.ndarray, np.ndarray] | None = None
                     ) -> tuple[list[float], float]:
  """Estimates the global offset vector between images 'a' and 'b'
  (stitch_rigid.py:39-67): areas with insufficient dynamic range (and the
  custom overlap masks) are excluded from ONE masked correlation of the whole
  strips (patch = image, step 1, batch 1)."""
  a_mask = range_mask(a, range_limit, filter_size,
                      None if masks is None else masks[0])
  b_mask = range_mask(b, range_limit, filter_size,
                      None if masks is None else masks[1])
  mfc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  xo, yo, _, pr = mfc.flow_field(
      a, b, pre_mask=a_mask, post_mask=b_mask, patch_size=tuple(a.shape),
      step=(1, 1), batch_size=1).squeeze()
  return [xo, yo], abs(pr)


def _estimate_offset_horiz(overlap: int, left: np.ndarray, right: np.ndarray,
                           range_limit: float, filter_size: int,
                           masks=None) -> tuple[list[float], float]:
  return _estimate_offset(left[:, -overlap:], right[:, :overlap], range_limit,
                          filter_size, masks)


def _estimate_offset_vert(overlap: int, top: np.ndarray, bot: np.ndarray,
                          range_limit: float, filter_size: int,
                          masks=None) -> tuple[list[float], float]:
  return _estimate_offset(top[-overlap:, :], bot[:overlap, :], range_limit,
                          filter_size, masks)


def _search_pair_offset(pre, post, axis: int, overlaps: Sequence[int],
                        max_ortho_shift: int, min_range: Sequence[float],
                        min_overlap: int, filter_size: int, masks):
  """Offset of one tile pair: tries dynamic-range limits and overlap widths in
  the order of the reference's `_find_offset` (stitch_rigid.py:150-219)."""
  estimate = _estimate_offset_horiz if axis == 0 else _estimate_offset_vert

  def valid(off):
    return abs(off[1 - axis]) < max_ortho_shift and abs(off[axis]) >= min_overlap

  def crop(m, width, tail):
    if axis == 0:
      m = m[:, -width:] if tail else m[:, :width]
    else:
      m = m[-width:, :] if tail else m[:width, :]
    # a fully masked overlap disables masking of that side
    return np.zeros_like(m) if np.all(m) else m

  chosen = None
  for range_limit in min_range:
    candidates = []
    best_pr, best_idx = 0.0, -1
    single_peak = False
    for width in overlaps:
      ov_masks = None
      if masks is not None:
        ov_masks = (crop(masks[0], width, True), crop(masks[1], width, False))
      off, pr = estimate(width, pre, post, range_limit, filter_size, ov_masks)
      off[axis] -= width
      if pr == 0.0:            # a single correlation peak: take it
        chosen, single_peak = off, True
        break
      candidates.append(off)
      if pr > best_pr and valid(off):
        best_pr, best_idx = pr, len(candidates) - 1
    if single_peak:
      break
    # consecutive widths that agree beat the best peak ratio
    closest, closest_idx = np.inf, 0
    for i in range(len(candidates) - 1):
      gap = np.abs(candidates[i + 1][axis] - candidates[i][axis])
      if gap < closest and valid(candidates[i + 1]):
        closest, closest_idx = gap, i
    if closest < 20:
      chosen = candidates[closest_idx + 1]
      break
    if best_idx >= 0:
      chosen = candidates[best_idx]
      break
  if chosen is None or abs(chosen[axis]) < min_overlap:
    return np.inf, np.inf
  return chosen


def compute_coarse_offsets(yx_shape: tuple[int, int],
                           tile_map: Mapping[TileXY, np.ndarray],
                           overlaps_xy=((200, 300), (200, 300)),
                           min_range=(10, 100, 0), min_overlap=160,
                           filter_size=10, mask_map: MaskMap | None = None
                           ) -> tuple[np.ndarray, np.ndarray]:
  """Coarse offset between every neighbouring tile pair
  (stitch_rigid.py:104-273).

  Returns (conn_x, conn_y), each [2, 1, *yx_shape]: the XY offset of tile
  (x+1, y) resp. (x, y+1) relative to tile (x, y); inf where no estimate met
  the criteria, nan where a tile is missing.
  """
  conns = []
  for axis in (0, 1):
    conn = np.full((2, 1, yx_shape[0], yx_shape[1]), np.nan)
    step = (1, 0) if axis == 0 else (0, 1)
    for y in range(yx_shape[0] - step[1]):
      for x in range(yx_shape[1] - step[0]):
        nbor = (x + step[0], y + step[1])
        if (x, y) not in tile_map or nbor not in tile_map:
          continue
        masks = None
        if mask_map is not None:
          width = max(overlaps_xy[axis])
          if axis == 0:
            masks = (mask_map[(x, y)][:, -width:], mask_map[nbor][:, :width])
          else:
            masks = (mask_map[(x, y)][-width:], mask_map[nbor][:width])
        conn[:, 0, y, x] = _search_pair_offset(
            tile_map[(x, y)], tile_map[nbor], axis, overlaps_xy[axis],
            max(overlaps_xy[1 - axis]), min_range, min_overlap, filter_size,
            masks)
    conns.append(conn)
  return conns[0], conns[1]


def interpolate_missing_offsets(conn: np.ndarray, axis: int,
                                max_r: int = 4) -> np.ndarray:
  """Replaces inf entries of a coarse offset array (in place) by the mean of
  the nearest finite neighbours along `axis` (-1: x, -2: y) within `max_r`
  (stitch_rigid.py:277-327)."""
  if conn.ndim != 4:
    raise ValueError('conn array must have rank 4')
  missing = np.isinf(conn[0, 0, ...])
  if not np.any(missing):
    return conn
  n = conn.shape[axis]
  for y, x in zip(*np.where(missing)):
    pos = [0, 0, int(y), int(x)]
    for r in range(1, max_r):
      found = []
      for sign in (-1, 1):
        q = list(pos)
        q[axis] += sign * r
        if 0 <= q[axis] < n and np.isfinite(conn[tuple(q)]):
          found.append(conn[:, q[1], q[2], q[3]])
      if found:
        conn[:, 0, y, x] = np.mean(found, axis=0)
        break
  return conn


def elastic_tile_mesh(x, cx, cy, k=None, stride=None, prefer_orig_order=False,
                      links=None) -> _dev.DeviceArray:
  """Force on the nodes of a 2-d tile mesh (stitch_rigid.py:330-388).

  x: [2, z, y, x] mesh where every node is a tile; cx / cy: desired XY offsets
  between (x, y) and (x+1, y) / (x, y+1) tiles.  The other arguments are unused
  (compatibility with the mesh solver).
  """
  if np.shape(x)[0] != 2:
    raise ValueError('x must be [2, z, y, x]')
  return mesh.TileMeshForce(cx, cy)(x, k, stride, prefer_orig_order, links)


def elastic_tile_mesh_3d(x, cx, cy, k=None, stride=None,
                         prefer_orig_order=False, links=None) -> _dev.DeviceArray:
  """Force on the nodes of a 3-d tile mesh (stitch_rigid.py:391-473);
  x, cx, cy: [3, z, y, x]."""
  if np.shape(x)[0] != 3:
    raise ValueError('x must be [3, z, y, x]')
  return mesh.TileMeshForce(cx, cy)(x, k, stride, prefer_orig_order, links)


def optimize_coarse_mesh(cx, cy, cfg: mesh.IntegrationConfig | None = None,
                         mesh_fn=elastic_tile_mesh) -> np.ndarray:
  """Rough initial positions of the tiles (stitch_rigid.py:476-523).

  With this module's `elastic_tile_mesh` / `elastic_tile_mesh_3d` the force is
  evaluated inside the HIP integrator; any other `mesh_fn(x, cx, cy, k, stride,
  prefer_orig_order)` is called once per step on the device-resident state.
  """
  if cfg is None:
    cfg = mesh.IntegrationConfig(
        dt=0.001, gamma=0.0, k0=0.0, k=0.1, stride=(1, 1), num_iters=1000,
        max_iters=100000, stop_v_max=0.001, dt_max=100)
  if mesh_fn is elastic_tile_mesh or mesh_fn is elastic_tile_mesh_3d:
    force = mesh.TileMeshForce(cx, cy)
    want = 2 if mesh_fn is elastic_tile_mesh else 3
    if force.ncomp != want:
      raise ValueError(f'{mesh_fn.__name__} needs [{want}, z, y, x] offsets')
  else:
    def force(x, *args, **kwargs):
      return mesh_fn(x, cx, cy, *args, **kwargs)
  # all-zero initial state = the regular grid layout with no overlap
  res = mesh.relax_mesh(np.zeros_like(cx), None, cfg, mesh_force=force)
  return np.array(res[0])
