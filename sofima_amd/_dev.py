"""Device-memory plumbing: torch tensors as HBM owners, nothing else."""
from __future__ import annotations

import numpy as np
import torch

from . import _abi


def device(dev=None) -> torch.device:
  _abi.require_gpu()
  if dev is None:
    return torch.device('cuda', torch.cuda.current_device())
  return torch.device(dev)


def stream_ptr() -> int:
  return torch.cuda.current_stream().cuda_stream


class DeviceArray:
  """Array living in HBM, returned where the reference returns a jax.Array.

  Supports what callers of the reference do with results: `np.array(x)` /
  `np.asarray(x)`, `.shape`, `.dtype`, indexing (returns NumPy) and being
  passed back into `relax_mesh` / `velocity_verlet` without a host round trip.
  """

  __array_priority__ = 100

  def __init__(self, tensor: torch.Tensor):
    self.tensor = tensor

  @property
  def shape(self):
    return tuple(self.tensor.shape)

  @property
  def ndim(self):
    return self.tensor.ndim

  @property
  def dtype(self):
    return np.dtype(np.float32) if self.tensor.dtype == torch.float32 else \
        np.dtype(str(self.tensor.dtype).replace('torch.', ''))

  def __array__(self, dtype=None, copy=None):
    arr = self.tensor.detach().cpu().numpy()
    return arr.astype(dtype) if dtype is not None else arr

  def __getitem__(self, idx):
    return np.asarray(self)[idx]

  def __len__(self):
    return self.tensor.shape[0]

  def block_until_ready(self):
    torch.cuda.synchronize(self.tensor.device)
    return self

  def __repr__(self):
    return f'DeviceArray(shape={self.shape}, device={self.tensor.device})'


def as_device_f32(x, dev, copy=True) -> torch.Tensor:
  """float32 contiguous device tensor from NumPy / DeviceArray / torch input.

  float64 inputs are down-cast like JAX does with x64 disabled.
  """
  if isinstance(x, DeviceArray):
    t = x.tensor
  elif isinstance(x, torch.Tensor):
    t = x
  else:
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
    return t.to(dev)
  t = t.to(device=dev, dtype=torch.float32).contiguous()
  return t.clone() if copy else t


def as_device_image(img, dev):
  """(tensor, dtype_tag): uint8 stays uint8, everything else becomes float32."""
  if isinstance(img, DeviceArray):
    img = img.tensor
  if isinstance(img, torch.Tensor):
    t = img
    if t.dtype not in (torch.uint8, torch.float32):
      t = t.to(torch.float32)
    return t.to(dev).contiguous(), (
        _abi.DTYPE_U8 if t.dtype == torch.uint8 else _abi.DTYPE_F32)
  arr = np.asarray(img)
  if arr.dtype == np.uint8:
    return upload(np.ascontiguousarray(arr), dev), _abi.DTYPE_U8
  arr = np.ascontiguousarray(arr, dtype=np.float32)
  return upload(arr, dev), _abi.DTYPE_F32


_upload_streams = {}


def upload(arr: np.ndarray, dev) -> torch.Tensor:
  """Host array -> device tensor on a side stream.  A copy from pageable memory
  makes the host wait for the stream it is enqueued on; on the compute stream
  that is everything launched so far, and a loop of flow_field() calls on NumPy
  strips (stitch_elastic.compute_flow_map) could never run ahead of the GPU.
  On its own stream the host waits for the copy alone."""
  dev = torch.device(dev)
  if dev.type != 'cuda':
    return torch.from_numpy(arr).to(dev)
  key = (dev.index if dev.index is not None else torch.cuda.current_device())
  side = _upload_streams.get(key)
  if side is None:
    side = _upload_streams[key] = torch.cuda.Stream(device=dev)
  cur = torch.cuda.current_stream(dev)
  with torch.cuda.stream(side):
    t = torch.from_numpy(arr).to(dev)      # returns when the copy has finished
  t.record_stream(cur)                     # allocated on `side`, used on `cur`
  return t


def as_device_mask(mask, dev):
  if mask is None:
    return None
  if isinstance(mask, DeviceArray):
    mask = mask.tensor
  if isinstance(mask, torch.Tensor):
    return (mask != 0).to(device=dev, dtype=torch.uint8).contiguous()
  arr = np.ascontiguousarray(np.asarray(mask) != 0).view(np.uint8)
  return torch.from_numpy(arr).to(dev)


def workspace(nbytes: int, dev) -> torch.Tensor:
  return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
