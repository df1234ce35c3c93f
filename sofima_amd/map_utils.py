"""Coordinate-map composition on MI355X.

Drop-in for `map_utils.compose_maps_fast` of the reference
(map_utils.py:616-734) and `map_utils.mask_irregular` (:737-786); the other
functions of the reference's map_utils.py (Delaunay inversion, resampling, ...)
are host geometry and out of scope.
"""
from __future__ import annotations

import collections.abc
import ctypes as C
from typing import Sequence

import numpy as np
import torch

from . import _abi
from . import _dev
from ._dev import DeviceArray


def _as_vec(value, dim):
  if not isinstance(value, collections.abc.Sequence):
    return (value,) * dim
  assert len(value) == dim, f'Dimension mismatch: {value=} vs {dim=}'
  return tuple(value)


def compose_maps_fast(map1, start1: Sequence[float], stride1, map2,
                      start2: Sequence[float], stride2,
                      mode: str = 'nearest') -> DeviceArray:
  """Composes two coordinate maps: map2(map1(z, y, x)) over map1's area.

  Same contract as the reference: maps are [2 or 3, z, y, x] in relative
  format, `start*` are [z]yx origins, `stride*` scalars or [z]yx tuples; `mode`
  is 'nearest' or 'constant' (out-of-range samples are NaN, like the
  reference's cval).  Invalid (NaN) entries are not interpolated.
  """
  assert np.shape(map1)[0] == np.shape(map2)[0]
  dim = np.shape(map1)[0]
  if mode not in ('nearest', 'constant'):
    raise NotImplementedError(mode)
  stride1 = _as_vec(stride1, dim)
  stride2 = _as_vec(stride2, dim)
  dev = _dev.device()
  m1 = _dev.as_device_f32(map1, dev, copy=False)
  m2 = _dev.as_device_f32(map2, dev, copy=False)
  if m1.ndim != 4 or m2.ndim != 4:
    raise ValueError('maps must be [2 or 3, z, y, x]')
  d = _abi.SfmComposeDesc()
  d.ncomp = dim
  d.mode = 0 if mode == 'nearest' else 1
  d.shape1 = (C.c_int32 * 3)(*m1.shape[1:])
  d.shape2 = (C.c_int32 * 3)(*m2.shape[1:])

  def zyx(v, fill):
    v = [float(a) for a in np.asarray(v).ravel()][-dim:]
    return (C.c_float * 3)(*([fill] * (3 - dim) + v))

  d.start1 = zyx(start1, 0.0)
  d.start2 = zyx(start2, 0.0)
  d.stride1 = zyx(stride1, 1.0)
  d.stride2 = zyx(stride2, 1.0)
  d.map1 = m1.data_ptr()
  d.map2 = m2.data_ptr()
  d.stream = _dev.stream_ptr()
  out = torch.empty_like(m1)
  _abi.check(_abi.load().sfm_compose_maps(C.byref(d), out.data_ptr()))
  return DeviceArray(out)


def mask_irregular(coord_map, stride: Sequence[float], frac: float,
                   max_frac: float | None = None,
                   dilation_iters: int = 1) -> np.ndarray:
  """Masks stretched / folded parts of a [2, y, x] relative coordinate map.

  Same contract as the reference (map_utils.py:737-786): masked entries are
  replaced with NaN IN PLACE (NumPy arrays are written back; torch tensors /
  DeviceArrays are modified on the device) and the bool mask [y, x] is
  returned.  `stride` is (x, y).  Computed in float32.
  """
  shape = np.shape(coord_map)
  assert len(shape) == 3
  assert shape[0] == 2
  if max_frac is None:
    max_frac = 2 - frac
  stride_x, stride_y = (float(v) for v in np.asarray(stride).ravel())
  dev = _dev.device()
  host = coord_map if isinstance(coord_map, np.ndarray) else None
  m = _dev.as_device_f32(coord_map, dev, copy=host is not None)
  d = _abi.SfmMaskIrregularDesc()
  d.shape = (C.c_int32 * 2)(int(shape[1]), int(shape[2]))
  d.stride = (C.c_float * 2)(stride_x, stride_y)
  d.frac = float(frac)
  d.max_frac = float(max_frac)
  d.dilation_iters = int(dilation_iters)
  d.stream = _dev.stream_ptr()
  bad = torch.empty(tuple(shape[1:]), dtype=torch.uint8, device=dev)
  _abi.check(_abi.load().sfm_mask_irregular(C.byref(d), m.data_ptr(), bad.data_ptr()))
  bad_h = bad.cpu().numpy().astype(bool)
  if host is not None:
    # in-place contract: the masked map computed on the device is copied back
    # (unmasked entries round-trip through float32 unchanged for float32 input)
    if host.dtype == np.float32:
      host[...] = m.cpu().numpy()
    else:
      host[:, bad_h] = np.nan
  return bad_h
