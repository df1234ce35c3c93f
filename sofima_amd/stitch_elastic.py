"""Target mesh of an elastic tile montage on MI355X.

Drop-in for the device part of the reference's `stitch_elastic.py`:
`compute_target_mesh` (stitch_elastic.py:624-676, with `_update_mesh` :573-620
and `_apply_flow` :456-570) -- the `prev_fn` that `mesh.relax_mesh` evaluates
inside every force evaluation when a montage is relaxed.  The host-side
drivers of that file (`compute_flow_map`, `aggregate_arrays`) are out of scope;
their outputs (`fx`, `fy`, `x`, `nbors`) are the inputs here.

The notebooks build the prev_fn as
    jit(lambda x: transpose(vmap(partial(compute_target_mesh, x=x, fx=fx, fy=fy,
                                         stride=stride))(nbors), [1, 0, 2, 3]))
`TargetMeshFn(nbors, fx, fy, stride)` is that function as one HIP kernel;
`mesh.relax_mesh(x, None, cfg, prev_fn=TargetMeshFn(...))` runs it natively.
"""
from __future__ import annotations

import ctypes as C
import enum

import numpy as np
import torch

from . import _abi
from . import _dev
from ._dev import DeviceArray


class NeighborInfo(enum.IntEnum):
  """Indices in a neighbour-info row (stitch_elastic.py:43-72)."""
  nbor_idx = 0
  flow_idx = 1
  coarse_offset_ortho = 2
  flow_size_ortho = 3
  flow_size_overlap = 4
  fine_off_x = 5
  fine_off_y = 6
  dim = 7
  coarse_offset_z = 8
  flow_size_z = 9
  fine_off_z = 10


class TargetMeshFn:
  """prev_fn(x) for all tiles of a montage.

  In-plane: x [2, N, y, x] -> [2, N, y, x] (flows fx / fy [2, n, y, x], stride
  yx, 8 neighbour fields).  Volumetric: x [3, N, z, y, x] -> the same (flows
  [3, n, z, y, x], stride zyx, 11 neighbour fields).
  """

  def __init__(self, nbors, fx, fy, stride=(20, 20)):
    dev = _dev.device()
    nb = np.ascontiguousarray(np.asarray(nbors), dtype=np.int32)
    if nb.ndim != 3 or nb.shape[1] != 4 or nb.shape[2] < 8:
      raise ValueError('nbors must be [n_tiles, 4, 8 or 11]')
    self.fx = _dev.as_device_f32(fx, dev, copy=False)
    self.fy = _dev.as_device_f32(fy, dev, copy=False)
    ncomp = int(self.fx.shape[0])
    if ncomp not in (2, 3) or self.fx.ndim != ncomp + 2 or self.fy.ndim != ncomp + 2:
      raise ValueError('flows must be [2, n, y, x] or [3, n, z, y, x]')
    if ncomp == 3 and nb.shape[2] < 11:
      raise ValueError('volumetric montages need the 11-field NeighborInfo rows')
    if len(stride) != ncomp:
      raise ValueError('stride must be [z]yx with one entry per spatial dimension')
    self.ncomp = ncomp
    self.nbors = torch.from_numpy(nb).to(dev)
    self.stride = tuple(float(s) for s in stride)
    d = _abi.SfmTargetMeshDesc()
    d.ncomp = ncomp
    d.n_tiles = nb.shape[0]
    lead = [1] * (3 - ncomp)
    d.fx_shape = (C.c_int32 * 3)(*lead, *self.fx.shape[2:])
    d.fy_shape = (C.c_int32 * 3)(*lead, *self.fy.shape[2:])
    d.n_fx = self.fx.shape[1]
    d.n_fy = self.fy.shape[1]
    d.nbor_fields = nb.shape[2]
    d.stride = (C.c_float * 3)(*([1.0] * (3 - ncomp)), *self.stride)
    d.nbors = self.nbors.data_ptr()
    d.fx = self.fx.data_ptr()
    d.fy = self.fy.data_ptr()
    self.desc = d

  def bind(self, x_t: torch.Tensor) -> _abi.SfmTargetMeshDesc:
    if (x_t.ndim != self.ncomp + 2 or x_t.shape[0] != self.ncomp or
        x_t.shape[1] != self.desc.n_tiles):
      raise ValueError('x must be [2, n_tiles, y, x] or [3, n_tiles, z, y, x]')
    self.desc.mesh_shape = (C.c_int32 * 3)(*([1] * (3 - self.ncomp)), *x_t.shape[2:])
    return self.desc

  def __call__(self, x) -> DeviceArray:
    dev = _dev.device()
    x_t = _dev.as_device_f32(x, dev, copy=False)
    d = self.bind(x_t)
    out = torch.empty_like(x_t)
    _abi.check(_abi.load().sfm_target_mesh(C.byref(d), x_t.data_ptr(),
                                           out.data_ptr(), _dev.stream_ptr()))
    return DeviceArray(out)


def compute_target_mesh(nbor_data, x, fx, fy, stride=(20, 20)) -> np.ndarray:
  """Target positions for ONE tile mesh (stitch_elastic.py:624-676).

  nbor_data: [4, 8 or 11] neighbour info of the tile; x, fx, fy as in the
  reference (in-plane [2, n, y, x] or volumetric [3, n, z, y, x]).
  Only this tile is evaluated (`SfmTargetMeshDesc.n_eval = 1`).
  """
  dev = _dev.device()
  x_t = _dev.as_device_f32(x, dev, copy=False)
  fn = TargetMeshFn(np.asarray(nbor_data)[None], fx, fy, stride)
  d = fn.desc
  # ONE row of neighbour info evaluated against the meshes of all n tiles
  d.n_tiles = int(x_t.shape[1])
  d.n_eval = 1
  d.mesh_shape = (C.c_int32 * 3)(*([1] * (3 - fn.ncomp)), *x_t.shape[2:])
  out = torch.empty((fn.ncomp, 1) + tuple(x_t.shape[2:]), dtype=torch.float32,
                    device=dev)
  _abi.check(_abi.load().sfm_target_mesh(C.byref(d), x_t.data_ptr(), out.data_ptr(),
                                         _dev.stream_ptr()))
  return out[:, 0].cpu().numpy()


def _overlap_strips(pre, post, off_xy, axis: int, stride):
  """Overlap strips of an adjacent tile pair, aligned to the flow stride
  (stitch_elastic.py:232-262), and the offset recorded for them (:277-280).

  off_xy: coarse (x, y) offset of `post`; stride is yx.
  """
  along = 1 - axis                    # image axis of the tile-tile connection
  # strip width: the start inside `pre` is moved down to a stride multiple
  extent = pre.shape[along]
  start = (extent + int(off_xy[axis])) // stride[along] * stride[along]
  overlap = extent - start
  # orthogonal shift, rounded to the stride of that image axis
  s_ortho = stride[::-1][1 - axis]
  ortho = int(s_ortho * np.round(off_xy[1 - axis] / s_ortho))
  pre_sl = [slice(None), slice(None)]
  post_sl = [slice(None), slice(None)]
  pre_sl[along] = slice(start, None)
  post_sl[along] = slice(0, overlap)
  if ortho > 0:                       # post sits further along the ortho axis
    pre_sl[axis] = slice(ortho, None)
    post_sl[axis] = slice(None, -ortho)
  elif ortho < 0:
    pre_sl[axis] = slice(None, ortho)
    post_sl[axis] = slice(-ortho, None)
  off = (-overlap, ortho) if axis == 0 else (ortho, -overlap)
  return pre[tuple(pre_sl)], post[tuple(post_sl)], off


# Tile pairs whose flow_field() call may be enqueued before the oldest result is
# fetched (compute_flow_map, compute_flow_map3d): enough for the host to stay
# ahead of the GPU, small enough to bound the device memory held by queued pairs.
MAX_PAIRS_IN_FLIGHT = 8


def compute_flow_map(tile_map, offset_map: np.ndarray, axis: int,
                     patch_size=(120, 120), stride=(20, 20),
                     batch_size: int = 256):
  """Fine flow between horizontally (axis 0) or vertically (axis 1) adjacent
  2-d tiles (stitch_elastic.py:198-282).

  tile_map: (x, y) -> tile image; offset_map: [2, y, x] coarse XY offsets of
  the (x+1, y) / (x, y+1) tile.  Returns ({(x, y): flow [4, gy, gx]},
  {(x, y): (off_x, off_y) at which the flow was computed}); the flow arrays are
  NaN-padded by patch // 2 // stride nodes so that they are aligned with the
  tile's mesh.  Every pair is one `flow_field` call on the overlap strips,
  i.e. the int8 matrix-core correlation for uint8 tiles.
  """
  from . import flow_field
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  patch_size = tuple(int(p) for p in patch_size)
  stride = tuple(int(s) for s in stride)
  pads = [(0, 0)] + [(p // 2 // s, p // 2 // s - 1)
                     for p, s in zip(patch_size, stride)]
  grid_y, grid_x = offset_map.shape[-2:]
  flows, offsets = {}, {}
  # every pair is enqueued before the first result is fetched: the fields stay
  # in HBM until the loop is done, so the host prepares pair n + 1 while the GPU
  # works on pair n (a strip pair of the 8 x 8 montage: 0.97 -> 0.7x ms)
  pending = []
  for y in range(grid_y - axis):
    for x in range(grid_x - (1 - axis)):
      off_xy = offset_map[:, y, x]
      if np.isnan(off_xy[0]):
        continue
      pre, post, off = _overlap_strips(
          tile_map[x, y], tile_map[x + (1 - axis), y + axis], off_xy, axis,
          stride)
      pending.append(((x, y), calc.flow_field(
          pre, post, patch_size=patch_size, step=stride, batch_size=batch_size,
          device_output=True)))
      offsets[x, y] = off
      # bounded run-ahead: strips and workspaces of the queued pairs stay
      # allocated until the compute stream has passed them
      if len(pending) >= MAX_PAIRS_IN_FLIGHT:
        key, f = pending.pop(0)
        flows[key] = np.pad(np.asarray(f), pads, constant_values=np.nan)
  for key, f in pending:
    flows[key] = np.pad(np.asarray(f), pads, constant_values=np.nan)
  return flows, offsets


def _aligned_overlap3d(tile_shape, offset, axis: int, stride):
  """Overlap region of a 3-d tile pair, aligned to the flow stride
  (stitch_elastic.py:125-180).

  tile_shape, offset: xyz; stride: zyx.  The neighbour sits at the grid step
  along `axis` plus `offset`.  Its position is nudged so that, inside the
  current tile, the overlap starts on a multiple of s = stride[2 - axis] along
  every axis (the reference aligns ALL axes to the stride of the connection
  axis).  Returns (start in the current tile, start in the neighbour, size)
  -- xyz integers -- and the xyz offset recorded for the pair.
  """
  shape = np.asarray(tile_shape, dtype=np.float64)
  # (the reference keeps the neighbour position in a connectomics BoundingBox,
  # whose corners are integers: a fractional coarse offset is truncated there --
  # third-party semantics, absent from /root/reference, pinned for integer
  # offsets only -- and the recorded offsets are ints)
  pos = np.array([shape[0] * (1 - axis) + offset[0], shape[1] * axis + offset[1],
                  offset[2]], dtype=np.float64)
  pos = np.trunc(pos)

  def overlap(nb):
    lo = np.maximum(0.0, nb)
    hi = np.minimum(shape, nb + shape)
    return lo, lo - nb, hi - lo        # start in current, start in neighbour, size

  s = stride[2 - axis]
  cur0, nb0, size0 = overlap(pos)
  nudge = np.zeros(3)
  # along the connection: the strip starts on a multiple of s inside the current
  # tile (moved towards its origin, i.e. the strip only grows)
  first = shape[axis] - size0[axis]
  nudge[axis] = -((shape[axis] - first // s * s) - size0[axis])
  for ax in range(3):
    if ax == axis:
      continue
    if cur0[ax] > 0:
      nudge[ax] = s * np.round(cur0[ax] / s) - cur0[ax]
    elif nb0[ax] > 0:
      nudge[ax] = -(s * np.round(nb0[ax] / s) - nb0[ax])
  pos = pos + nudge
  cur, nb, size = overlap(pos)
  assert np.all(cur % s == 0) and np.all(nb % s == 0)
  rec = pos.copy()
  rec[axis] = -size[axis]
  return (cur.astype(int), nb.astype(int), size.astype(int),
          tuple(int(v) for v in rec))


def compute_flow_map3d(tile_map, tile_shape, offset_map: np.ndarray, axis: int,
                       patch_size=(120, 120, 120), stride=(40, 40, 40),
                       batch_size: int = 16):
  """Fine flow between horizontally (axis 0) or vertically (axis 1) adjacent
  3-d tiles (stitch_elastic.py:85-194).

  tile_map: (x, y) -> [1, z, y, x] volume (any indexable); tile_shape: xyz;
  offset_map: [3, 1, y, x] coarse xyz offsets of the (x+1, y) / (x, y+1) tile;
  patch_size / stride: zyx.  Returns ({(x, y): flow [5, gz, gy, gx]}, {(x, y):
  xyz offset at which the neighbour was positioned for the flow}); flows are
  NaN-padded by patch // 2 // stride nodes like the reference's.  Every pair is
  one volumetric `flow_field` call (FFT form) on the aligned overlap.
  """
  from . import flow_field
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  patch_size = tuple(int(p) for p in patch_size)
  stride = tuple(int(v) for v in stride)
  pads = [(0, 0)] + [(p // 2 // v, p // 2 // v - 1) for p, v in zip(patch_size, stride)]
  grid_y, grid_x = offset_map.shape[-2:]
  flows, offsets = {}, {}
  pending = []
  for y in range(grid_y - axis):
    for x in range(grid_x - (1 - axis)):
      cur, nb, size, rec = _aligned_overlap3d(tile_shape, offset_map[:, 0, y, x], axis,
                                              stride)

      def crop(tile, start):          # xyz start / size -> [z, y, x] block
        sl = tuple(slice(int(a), int(a + n)) for a, n in zip(start[::-1], size[::-1]))
        return np.asarray(tile[(slice(None),) + sl]).squeeze(axis=0)

      pre = crop(tile_map[x, y], cur)
      post = crop(tile_map[x + (1 - axis), y + axis], nb)
      assert pre.shape == post.shape
      pending.append(((x, y), calc.flow_field(
          pre, post, patch_size=patch_size, step=stride, batch_size=batch_size,
          device_output=True)))
      offsets[x, y] = rec
      if len(pending) >= MAX_PAIRS_IN_FLIGHT:   # two overlap volumes + FFT workspace each
        key, f = pending.pop(0)
        flows[key] = np.pad(np.asarray(f), pads, constant_values=np.nan)
  for key, f in pending:   # fetched after the last pair is enqueued (compute_flow_map)
    flows[key] = np.pad(np.asarray(f), pads, constant_values=np.nan)
  return flows, offsets
