"""ctypes binding of libsofima_amd.so (include/sofima_amd.h).

PyTorch-ROCm tensors are used only as the owner of device memory and streams;
every kernel is reached through the C ABI.  There is no CPU fallback: if the
library is missing or no GPU is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import _build

_lib = None
_lock = threading.Lock()

SFM_OK = 0
DTYPE_U8 = 0
DTYPE_F32 = 1
DTYPE_U16 = 2
DTYPE_I32 = 3
WARP_NEAREST = 0
WARP_TABLE = 1
XCORR_AUTO = 0
XCORR_DIRECT = 1
XCORR_MFMA_I8 = 2
XCORR_FFT = 3
MAX_LINKS = 13

i32 = C.c_int32


class SfmXcorrDesc(C.Structure):
  _fields_ = [
      ('ndim', i32),
      ('dtype', i32),
      ('pre_image', C.c_void_p),
      ('post_image', C.c_void_p),
      ('pre_mask', C.c_void_p),
      ('post_mask', C.c_void_p),
      ('pre_shape', i32 * 3),
      ('post_shape', i32 * 3),
      ('pre_mask_shape', i32 * 3),
      ('post_mask_shape', i32 * 3),
      ('patch', i32 * 3),
      ('post_patch', i32 * 3),
      ('pre_starts', C.c_void_p),
      ('post_starts', C.c_void_p),
      ('batch', i32),
      ('group', i32),
      ('use_mean', i32),
      ('mean', C.c_float),
      ('min_distance', i32),
      ('threshold_rel', C.c_float),
      ('peak_radius', i32 * 3),
      ('method', i32),
      ('workspace', C.c_void_p),
      ('workspace_bytes', C.c_size_t),
      ('stream', C.c_void_p),
  ]


class SfmMaskCountDesc(C.Structure):
  _fields_ = [
      ('ndim', i32),
      ('shape', i32 * 3),
      ('patch', i32 * 3),
      ('step', i32 * 3),
      ('mask', C.c_void_p),
      ('stream', C.c_void_p),
  ]


class SfmPeaksDesc(C.Structure):
  _fields_ = [
      ('ndim', i32),
      ('batch', i32),
      ('shape', i32 * 3),
      ('center_offset', C.c_float * 3),
      ('min_distance', i32),
      ('threshold_rel', C.c_float),
      ('peak_radius', i32 * 3),
      ('surface', C.c_void_p),
      ('workspace', C.c_void_p),
      ('workspace_bytes', C.c_size_t),
      ('stream', C.c_void_p),
  ]


class SfmComposeDesc(C.Structure):
  _fields_ = [
      ('ncomp', i32),
      ('mode', i32),
      ('shape1', i32 * 3),
      ('shape2', i32 * 3),
      ('start1', C.c_float * 3),
      ('start2', C.c_float * 3),
      ('stride1', C.c_float * 3),
      ('stride2', C.c_float * 3),
      ('map1', C.c_void_p),
      ('map2', C.c_void_p),
      ('stream', C.c_void_p),
  ]


class SfmCleanFlowDesc(C.Structure):
  _fields_ = [
      ('dim', i32),
      ('channels', i32),
      ('shape', i32 * 3),
      ('min_peak_ratio', C.c_float),
      ('min_peak_sharpness', C.c_float),
      ('max_magnitude', C.c_float),
      ('max_deviation', C.c_float),
      ('flow', C.c_void_p),
      ('stream', C.c_void_p),
  ]


class SfmMaskIrregularDesc(C.Structure):
  _fields_ = [
      ('shape', i32 * 2),
      ('stride', C.c_float * 2),
      ('frac', C.c_float),
      ('max_frac', C.c_float),
      ('dilation_iters', i32),
      ('stream', C.c_void_p),
  ]


class SfmFlowStartsDesc(C.Structure):
  _fields_ = [
      ('ndim', i32),
      ('n', i32),
      ('step', i32 * 3),
      ('patch', i32 * 3),
      ('post_patch', i32 * 3),
      ('pre_shape', i32 * 3),
      ('post_shape', i32 * 3),
      ('positions', C.c_void_p),
      ('pre_targeting_field', C.c_void_p),
      ('post_targeting_field', C.c_void_p),
      ('pre_targeting_shape', i32 * 3),
      ('post_targeting_shape', i32 * 3),
      ('pre_targeting_step', i32 * 3),
      ('post_targeting_step', i32 * 3),
      ('pre_starts', C.c_void_p),
      ('post_starts', C.c_void_p),
      ('pre_offsets', C.c_void_p),
      ('post_offsets', C.c_void_p),
      ('stream', C.c_void_p),
  ]


class SfmFlowScatterDesc(C.Structure):
  _fields_ = [
      ('ndim', i32),
      ('n', i32),
      ('grid', i32 * 3),
      ('peaks', C.c_void_p),
      ('positions', C.c_void_p),
      ('pre_offsets', C.c_void_p),
      ('post_offsets', C.c_void_p),
      ('out', C.c_void_p),
      ('stream', C.c_void_p),
  ]


class SfmWarpDesc(C.Structure):
  _fields_ = [
      ('dtype', i32),
      ('interpolation', i32),
      ('ksize', i32),
      ('image_shape', i32 * 2),
      ('map_shape', i32 * 2),
      ('out_shape', i32 * 2),
      ('map_origin', C.c_double * 2),
      ('stride', C.c_double),
      ('image', C.c_void_p),
      ('coord_map', C.c_void_p),
      ('weights', C.c_void_p),
      ('out', C.c_void_p),
      ('stream', C.c_void_p),
      ('coord_map_f64', i32),
  ]


class SfmNdWarpDesc(C.Structure):
  _fields_ = [
      ('ndim', i32),
      ('dtype', i32),
      ('order', i32),
      ('image_shape', i32 * 3),
      ('map_shape', i32 * 3),
      ('out_shape', i32 * 3),
      ('stride', C.c_double * 3),
      ('offset', C.c_double * 3),
      ('image', C.c_void_p),
      ('src_map', C.c_void_p),
      ('out', C.c_void_p),
      ('stream', C.c_void_p),
  ]


class SfmRangeMaskDesc(C.Structure):
  _fields_ = [
      ('dtype', i32),
      ('shape', i32 * 2),
      ('filter_size', i32),
      ('range_limit', C.c_double),
      ('image', C.c_void_p),
      ('extra_mask', C.c_void_p),
      ('stream', C.c_void_p),
  ]


class SfmTargetMeshDesc(C.Structure):
  _fields_ = [
      ('ncomp', i32),
      ('n_tiles', i32),
      ('mesh_shape', i32 * 3),
      ('fx_shape', i32 * 3),
      ('fy_shape', i32 * 3),
      ('n_fx', i32),
      ('n_fy', i32),
      ('nbor_fields', i32),
      ('stride', C.c_float * 3),
      ('nbors', C.c_void_p),
      ('fx', C.c_void_p),
      ('fy', C.c_void_p),
      ('n_eval', i32),
  ]


FORCE_SPRINGS = 0
FORCE_TILE_MESH = 1
FORCE_EXTERNAL = 2
SfmForceCallback = C.CFUNCTYPE(C.c_int, C.c_void_p)


class SfmMeshDesc(C.Structure):
  _fields_ = [
      ('ncomp', i32),
      ('shape', i32 * 4),
      ('stride', C.c_double * 3),
      ('k', C.c_double),
      ('k0', C.c_double),
      ('prefer_orig_order', i32),
      ('n_links', i32),
      ('links', (i32 * 3) * MAX_LINKS),
      ('dt', C.c_double),
      ('gamma', C.c_double),
      ('num_iters', i32),
      ('fire', i32),
      ('f_alpha', C.c_double),
      ('f_inc', C.c_double),
      ('f_dec', C.c_double),
      ('alpha0', C.c_double),
      ('n_min', i32),
      ('dt_max', C.c_double),
      ('final_cap', C.c_double),
      ('cap_scale', C.c_double),
      ('cap_upscale_every', i32),
      ('remove_drift', i32),
      ('x', C.c_void_p),
      ('v', C.c_void_p),
      ('a', C.c_void_p),
      ('prev', C.c_void_p),
      ('workspace', C.c_void_p),
      ('workspace_bytes', C.c_size_t),
      ('stream', C.c_void_p),
      ('target', C.POINTER(SfmTargetMeshDesc)),
      ('force_kind', i32),
      ('cx', C.c_void_p),
      ('cy', C.c_void_p),
      ('ext_force', C.c_void_p),
      ('force_cb', SfmForceCallback),
      ('force_user', C.c_void_p),
      ('ext_prev', C.c_void_p),
      ('prev_cb', SfmForceCallback),
      ('prev_user', C.c_void_p),
  ]


class SfmMeshShard(C.Structure):
  _fields_ = [
      ('own_y0', i32),
      ('own_y1', i32),
      ('global_nodes', C.c_int64),
      ('n_ranks', i32),
      ('sums', C.c_void_p),
      ('my_sums', C.c_void_p),
      ('phase', i32),
      ('cap0', C.c_float),
  ]


_fp = C.POINTER(C.c_float)
# host-staged transport of the banded loop (SfmHostHaloFn / SfmHostAllgatherFn)
HOST_HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _fp, _fp, C.c_int, _fp, _fp,
                           C.c_size_t)
HOST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _fp, _fp, C.c_size_t)


class SfmBandedDesc(C.Structure):
  _fields_ = [
      ('n_local', i32),
      ('bands', C.POINTER(SfmMeshDesc)),
      ('shards', C.POINTER(SfmMeshShard)),
      ('comm', C.c_void_p),
      ('rank', i32),
      ('n_ranks', i32),
      ('flags', i32),
      ('comm_stream', C.c_void_p),
      ('scratch', C.c_void_p),
      ('scratch_bytes', C.c_size_t),
      ('host_halo', HOST_HALO_FN),
      ('host_allgather', HOST_ALLGATHER_FN),
      ('host_user', C.c_void_p),
  ]


BANDED_LOOPBACK = 1
BANDED_NO_OVERLAP = 2
COMM_ID_BYTES = 128
REDUCE_SUM = 0
REDUCE_MAX = 1


class SfmFireState(C.Structure):
  _fields_ = [('dt', C.c_float), ('alpha', C.c_float), ('n_pos', i32),
              ('cap', C.c_float)]


class SfmProfile(C.Structure):
  _fields_ = [('kernel_ms', C.c_double * 2), ('launches', C.c_int64 * 2),
              ('clock_mhz', C.c_double * 2),
              ('tiles_skipped', C.c_int64 * 2), ('tiles_drawn', C.c_int64 * 2),
              ('col_tiles_skipped', C.c_int64 * 2),
              ('mfma_issued', C.c_int64 * 2),
              ('tiles_abandoned', C.c_int64 * 2)]


class SfmChunkStats(C.Structure):
  _fields_ = [('e_kin', C.c_float), ('v_max', C.c_float)]


# name -> (restype, argtypes); mirrors include/sofima_amd.h one to one.
SIGNATURES = {
    'sfm_version': (C.c_int, []),
    'sfm_last_error': (C.c_char_p, []),
    'sfm_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'sfm_set_option': (C.c_int, [C.c_char_p, C.c_char_p]),
    'sfm_get_option': (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t]),
    'sfm_profile_enable': (C.c_int, [C.c_int]),
    'sfm_profile_read': (C.c_int, [C.POINTER(SfmProfile)]),
    'sfm_xcorr_workspace_bytes': (C.c_size_t, [C.POINTER(SfmXcorrDesc)]),
    'sfm_xcorr_peaks': (C.c_int, [C.POINTER(SfmXcorrDesc), C.c_void_p]),
    'sfm_xcorr_surface': (C.c_int, [C.POINTER(SfmXcorrDesc), C.c_void_p]),
    'sfm_mask_patch_counts': (C.c_int, [C.POINTER(SfmMaskCountDesc), C.c_void_p]),
    'sfm_peaks_workspace_bytes': (C.c_size_t, [C.POINTER(SfmPeaksDesc)]),
    'sfm_peaks': (C.c_int, [C.POINTER(SfmPeaksDesc), C.c_void_p]),
    'sfm_compose_maps': (C.c_int, [C.POINTER(SfmComposeDesc), C.c_void_p]),
    'sfm_clean_flow': (C.c_int, [C.POINTER(SfmCleanFlowDesc), C.c_void_p]),
    'sfm_mask_irregular': (C.c_int, [C.POINTER(SfmMaskIrregularDesc), C.c_void_p,
                                     C.c_void_p]),
    'sfm_flow_starts': (C.c_int, [C.POINTER(SfmFlowStartsDesc)]),
    'sfm_flow_scatter': (C.c_int, [C.POINTER(SfmFlowScatterDesc)]),
    'sfm_warp_section': (C.c_int, [C.POINTER(SfmWarpDesc)]),
    'sfm_ndimage_warp': (C.c_int, [C.POINTER(SfmNdWarpDesc)]),
    'sfm_range_mask': (C.c_int, [C.POINTER(SfmRangeMaskDesc), C.c_void_p]),
    'sfm_target_mesh': (C.c_int, [C.POINTER(SfmTargetMeshDesc), C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    'sfm_mesh_workspace_bytes': (C.c_size_t, [C.POINTER(SfmMeshDesc)]),
    'sfm_mesh_force': (C.c_int, [C.POINTER(SfmMeshDesc), C.c_void_p]),
    'sfm_mesh_relax_chunk': (C.c_int, [C.POINTER(SfmMeshDesc),
                                       C.POINTER(SfmFireState),
                                       C.POINTER(SfmChunkStats)]),
}
SIGNATURES.update({
    'sfm_mesh_shard_begin': (C.c_int, [C.POINTER(SfmMeshDesc), C.POINTER(SfmMeshShard),
                                       C.POINTER(SfmFireState)]),
    'sfm_mesh_shard_advance': (C.c_int, [C.POINTER(SfmMeshDesc),
                                         C.POINTER(SfmMeshShard)]),
    'sfm_mesh_shard_integrate': (C.c_int, [C.POINTER(SfmMeshDesc),
                                           C.POINTER(SfmMeshShard)]),
    'sfm_mesh_shard_finish': (C.c_int, [C.POINTER(SfmMeshDesc), C.POINTER(SfmMeshShard),
                                        C.POINTER(SfmFireState),
                                        C.POINTER(SfmChunkStats)]),
    'sfm_mesh_banded_scratch_bytes': (C.c_size_t, [C.POINTER(SfmBandedDesc)]),
    'sfm_mesh_relax_banded': (C.c_int, [C.POINTER(SfmBandedDesc), C.POINTER(SfmFireState),
                                        C.POINTER(SfmChunkStats)]),
    'sfm_comm_unique_id': (C.c_int, [C.c_void_p]),
    'sfm_comm_init': (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]),
    'sfm_comm_destroy': (C.c_int, [C.c_void_p]),
    'sfm_debug_fft1d': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p]),
    'sfm_comm_halo_exchange': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    'sfm_comm_allgather': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    'sfm_comm_allreduce_scalars': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t,
                                             C.c_int, C.c_void_p]),
})


class SofimaAmdError(RuntimeError):
  """A C-ABI call returned an error code."""


def lib_path() -> str:
  return os.environ.get('SOFIMA_AMD_LIB', _build.LIB_PATH)


def load():
  """Loads libsofima_amd.so; raises if it has not been built."""
  global _lib
  with _lock:
    if _lib is not None:
      return _lib
    path = lib_path()
    if not os.path.exists(path):
      raise SofimaAmdError(
          f'{path} not found: build it with `python -m sofima_amd._build` '
          '(or __graft_entry__.build()). There is no CPU fallback.')
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
      fn = getattr(lib, name)
      fn.restype = res
      fn.argtypes = args
    _lib = lib
    return _lib


def check(rc: int):
  if rc != SFM_OK:
    msg = load().sfm_last_error().decode('utf-8', 'replace')
    raise SofimaAmdError(f'libsofima_amd error {rc}: {msg}')


_explicit = {}   # switches set through set_option (name -> str), for option()


def set_option(name: str, value=None):
  """Sets a behaviour switch of the library (include/sofima_amd.h lists them);
  None removes the explicit setting.  Explicit settings win over the
  environment variable of the same name, which is only the default."""
  check(load().sfm_set_option(name.encode(), None if value is None else
                              str(value).encode()))
  if value is None:
    _explicit.pop(name, None)
  else:
    _explicit[name] = str(value)


def get_option(name: str):
  """The value in effect (explicit setting, else environment), or None."""
  buf = C.create_string_buffer(64)
  rc = load().sfm_get_option(name.encode(), buf, 64)
  if rc == 1:
    return None
  check(rc)
  return buf.value.decode()


class option:
  """`with _abi.option('SFM_MFMA_PRUNE', 0): ...` -- a switch for one block
  (tests, A/B measurements); restores the previous explicit setting, or the
  environment default when there was none."""

  def __init__(self, name, value):
    self.name, self.value = name, value

  def __enter__(self):
    self.prev = _explicit.get(self.name)
    set_option(self.name, self.value)
    return self

  def __exit__(self, *exc):
    set_option(self.name, self.prev)
    return False


def device_count() -> int:
  n = C.c_int(0)
  check(load().sfm_device_count(C.byref(n)))
  return n.value


def require_gpu():
  """Fails loudly when the HIP path cannot run (no silent fallback)."""
  import torch
  if not torch.cuda.is_available() or device_count() < 1:
    raise SofimaAmdError(
        'sofima_amd needs an MI355X (gfx950) GPU: no HIP device is visible and '
        'there is no CPU fallback.')
