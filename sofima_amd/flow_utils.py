"""Flow-field clean-up on MI355X.

Drop-in for `flow_utils.clean_flow` of the reference (flow_utils.py:37-78), the
quality filter between flow estimation and mesh relaxation (SURVEY.md 8f,
rank 2).  `reconcile_flows` and the other helpers of the reference's
flow_utils.py are host-side post-processing and out of scope.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _abi
from . import _dev
from ._dev import DeviceArray


def clean_flow(flow, min_peak_ratio: float, min_peak_sharpness: float,
               max_magnitude: float, max_deviation: float,
               dim: int = 2) -> DeviceArray:
  """Removes flow vectors that do not fulfill quality requirements.

  Same contract as the reference: `flow` is [c, z, y, x] with c = dim (vectors
  only) .. dim + 2 (vectors, peak sharpness, peak ratio); the result is the
  [dim, z, y, x] vector field with NaN where the sharpness / ratio / magnitude
  / deviation-from-the-3x3(x3)-median criteria fail.  Accepts NumPy arrays,
  torch tensors or DeviceArrays; the result stays on the device (np.asarray()
  copies it back) and is computed in float32.
  """
  assert dim in (2, 3)
  dev = _dev.device()
  f = _dev.as_device_f32(flow, dev, copy=False)
  if f.ndim != 4:
    raise ValueError('flow must be [c, z, y, x]')
  assert dim <= f.shape[0] <= dim + 2
  d = _abi.SfmCleanFlowDesc()
  d.dim = dim
  d.channels = f.shape[0]
  d.shape = (C.c_int32 * 3)(*f.shape[1:])
  d.min_peak_ratio = float(min_peak_ratio)
  d.min_peak_sharpness = float(min_peak_sharpness)
  d.max_magnitude = float(max_magnitude)
  d.max_deviation = float(max_deviation)
  d.flow = f.data_ptr()
  d.stream = _dev.stream_ptr()
  out = torch.empty((dim,) + tuple(f.shape[1:]), dtype=torch.float32, device=dev)
  _abi.check(_abi.load().sfm_clean_flow(C.byref(d), out.data_ptr()))
  return DeviceArray(out)


def apply_mask(flow: np.ndarray, mask: np.ndarray) -> None:
  """In-place NaN masking of a host flow array (flow_utils.py:32-34)."""
  for i in range(flow.shape[0]):
    flow[i, ...][mask] = np.nan
