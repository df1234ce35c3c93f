"""Hand-written FFT correlation (csrc/sfm_fft_own.hip, the only FFT of the
library) vs numpy.fft and the float64 oracle (-m gpu)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import flow_oracle

pytestmark = pytest.mark.gpu


def _fft1d(x, n, inverse):
  """x: [n_in, pencils] complex64 -> [n, pencils] through sfm_debug_fft1d."""
  import torch
  from sofima_amd import _abi
  lib = _abi.load()
  n_in, pencils = x.shape
  xin = torch.from_numpy(np.ascontiguousarray(x.astype(np.complex64))).cuda()
  out = torch.zeros((n, pencils), dtype=torch.complex64, device='cuda')
  rc = lib.sfm_debug_fft1d(xin.data_ptr(), out.data_ptr(), n, n_in, pencils, int(inverse),
                           torch.cuda.current_stream().cuda_stream)
  assert rc == 0
  torch.cuda.synchronize()
  return out.cpu().numpy()


@pytest.mark.parametrize('n', [160, 80, 2, 4, 6, 10, 30, 96, 120, 128, 150, 240, 256, 320, 400,
                               512, 600, 750, 960, 1000])
@pytest.mark.parametrize('inverse', [False, True])
def test_pencil_fft_matches_numpy(gpu, n, inverse):
  rng = np.random.default_rng(n)
  for n_in, pencils in ((n, 37), (max(1, n // 2), 16), (n, 1)):
    x = (rng.standard_normal((n_in, pencils)) + 1j * rng.standard_normal((n_in, pencils)))
    got = _fft1d(x, n, inverse)
    xp = np.zeros((n, pencils), np.complex128)
    xp[:n_in] = x
    want = np.fft.ifft(xp, axis=0) * n if inverse else np.fft.fft(xp, axis=0)
    np.testing.assert_allclose(got, want, atol=2e-5 * np.abs(want).max())


def _with_env(env, fn):
  old = {k: os.environ.get(k) for k in env}
  os.environ.update(env)
  try:
    return fn()
  finally:
    for k, v in old.items():
      if v is None:
        os.environ.pop(k, None)
      else:
        os.environ[k] = v


@pytest.mark.parametrize('p,q', [((20, 24, 30), (20, 24, 30)), ((16, 40, 25), (9, 33, 25)),
                                 ((80, 80, 80), (80, 80, 80)), ((5, 7, 12), (5, 6, 11))])
def test_own_fft_correlation_matches_plans_and_oracle(gpu, p, q):
  """Full volumetric correlation surfaces: hand-written transforms == float64
  oracle (the 80^3 case, too slow for the oracle, is covered by the flow test)."""
  from sofima_amd import flow_field
  rng = np.random.default_rng(sum(p))
  b = 3 if p[0] < 80 else 2
  a = rng.integers(0, 255, (b,) + p).astype(np.uint8)
  c = rng.integers(0, 255, (b,) + q).astype(np.uint8)
  own = flow_field.masked_xcorr(a, c, dim=3, method=3, mean=None)
  scale = np.abs(own).max()
  assert np.isfinite(own).all() and scale > 0
  if p[0] < 80:
    a0 = a.astype(np.float64) - a.reshape(b, -1).mean(1)[:, None, None, None]
    c0 = c.astype(np.float64) - c.reshape(b, -1).mean(1)[:, None, None, None]
    want = flow_oracle.xcorr_surface(a0, c0, dim=3, dtype=np.float64)
    np.testing.assert_allclose(own, want, atol=1e-5 * scale)


def test_own_fft_flow_3d(gpu):
  """flow_field on a shifted volume through the hand-written transforms: the
  exact shift everywhere, the same field twice."""
  from scipy import ndimage
  from sofima_amd import flow_field
  rng = np.random.default_rng(3)
  vol = ndimage.gaussian_filter(rng.standard_normal((110, 130, 150)), 1.5)
  vol = ((vol - vol.min()) / (vol.max() - vol.min()) * 255).astype(np.uint8)
  pre, post = vol[:100, :120, :140], vol[3:103, 2:122, 5:145]
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  run = lambda: calc.flow_field(pre, post, (40, 48, 64), 20, batch_size=8)
  own = run()
  assert (own[0] == 5).all() and (own[1] == 2).all() and (own[2] == 3).all()
  np.testing.assert_array_equal(own, run())


@pytest.mark.parametrize('p,q', [((12, 20, 30), (12, 20, 30)), ((16, 24, 25), (9, 20, 25))])
def test_own_fft_masked_volumes_match_plans_and_oracle(gpu, p, q):
  """Masked (Padfield) volumetric correlation: six spectra and six inverse
  transforms of products through the hand-written passes == oracle."""
  from sofima_amd import flow_field
  rng = np.random.default_rng(sum(q))
  b = 3
  a = rng.integers(0, 255, (b,) + p).astype(np.uint8)
  c = rng.integers(0, 255, (b,) + q).astype(np.uint8)
  am = rng.random(a.shape) < 0.2
  cm = rng.random(c.shape) < 0.1
  am[0, :3] = True
  own = flow_field.masked_xcorr(a, c, am, cm, dim=3, method=3)
  want = flow_oracle.xcorr_surface(a, c, am, cm, dim=3)
  flipped = (own == 0) != (want == 0)
  assert flipped.mean() < 2e-3
  bad = (np.abs(own - want) > 5e-5) & ~flipped
  assert bad.mean() < 1e-4


@pytest.mark.parametrize('p,q,dtype', [((160, 160), (160, 160), np.float32),
                                        ((200, 180), (200, 180), np.uint8),
                                        ((256, 256), (256, 256), np.uint8),
                                        ((300, 90), (300, 90), np.uint8),     # F = 600: 8 pencils
                                        ((96, 130), (64, 100), np.float32),
                                        ((700, 40), (700, 40), np.uint8),     # F = 1440: 4 pencils
                                        ((1200, 64), (1200, 64), np.uint8),   # y 2400 = 48 x 50
                                        ((4096, 24), (4000, 20), np.uint8),   # y 8100 = 90 x 90
                                        ((64, 1200), (64, 1200), np.uint8),   # long x: axis swap
                                        ((30, 2500), (24, 2300), np.float32),
                                        ((900, 1000), (900, 1000), np.uint8),  # both long
                                        ((1100, 870), (1000, 866), np.float32)])
def test_own_fft_in_plane_patches_match_plans_and_oracle(gpu, p, q, dtype):
  """2-D patches that take the FFT form (float images, patches wider than 160):
  the hand-written passes with the z passes skipped and the product riding on
  the y pass == float64 oracle (both the 16- and the 8-pencil tiles; a long y
  axis through the four-step split; a long x axis through the axis swap)."""
  from sofima_amd import flow_field
  rng = np.random.default_rng(sum(p))
  b = 3 if p[0] * p[1] < 500000 else 2
  a = rng.integers(0, 255, (b,) + p).astype(dtype)
  c = rng.integers(0, 255, (b,) + q).astype(dtype)
  own = flow_field.masked_xcorr(a, c, dim=2, method=3, mean=None)
  scale = np.abs(own).max()
  a0 = a.astype(np.float64) - a.reshape(b, -1).mean(1)[:, None, None]
  c0 = c.astype(np.float64) - c.reshape(b, -1).mean(1)[:, None, None]
  want = flow_oracle.xcorr_surface(a0, c0, dim=2, dtype=np.float64)
  np.testing.assert_allclose(own, want, atol=1e-5 * scale)


def test_own_fft_masked_in_plane_patches(gpu):
  """Masked (Padfield) 2-D float patches through the hand-written passes ==
  oracle; also a strip with a long axis either way (the `_estimate_offset`
  regime: four-step split, axis swap)."""
  from sofima_amd import flow_field
  rng = np.random.default_rng(9)
  for b, p, q in ((3, (120, 140), (120, 140)), (1, (1500, 60), (1500, 60)),
                  (2, (48, 1100), (48, 1100)), (1, (880, 900), (880, 900))):
    _masked_case(flow_field, rng, b, p, q)


def _masked_case(flow_field, rng, b, p, q):
  a = rng.integers(0, 255, (b,) + p).astype(np.float32)
  c = rng.integers(0, 255, (b,) + q).astype(np.float32)
  am = rng.random(a.shape) < 0.2
  cm = rng.random(c.shape) < 0.1
  am[0, :30, :10] = True
  own = flow_field.masked_xcorr(a, c, am, cm, dim=2, method=3)
  want = flow_oracle.xcorr_surface(a, c, am, cm, dim=2)
  flipped = (own == 0) != (want == 0)
  assert flipped.mean() < 2e-3
  bad = (np.abs(own - want) > 5e-5) & ~flipped
  assert bad.mean() < 1e-4


def test_oversize_volume_is_refused_before_any_allocation(gpu):
  """A volumetric patch whose padded extent is beyond the hand-written transforms
  (an axis > 1728) is an error with a message when the workspace is sized:
  nothing is allocated or launched."""
  import ctypes as C
  import torch
  from sofima_amd import _abi, flow_field as ff
  lib = _abi.load()
  d = _abi.SfmXcorrDesc()
  d.ndim = 3
  d.dtype = _abi.DTYPE_F32
  shape = (1000, 8, 8)
  d.pre_shape = (C.c_int32 * 3)(*shape)
  d.post_shape = (C.c_int32 * 3)(*shape)
  d.patch = (C.c_int32 * 3)(*shape)          # padded z extent: 2000 > 1728
  d.post_patch = (C.c_int32 * 3)(*shape)
  d.batch = 1
  d.group = 1
  d.use_mean = 0
  d.min_distance = 2
  d.threshold_rel = 0.5
  d.peak_radius = (C.c_int32 * 3)(5, 5, 5)
  d.method = _abi.XCORR_FFT
  img = torch.zeros(shape, dtype=torch.float32, device=gpu)
  st = torch.zeros((2, 1, 3), dtype=torch.int32, device=gpu)
  d.pre_image = d.post_image = img.data_ptr()
  d.pre_starts = d.post_starts = st.data_ptr()
  before = torch.cuda.memory_allocated(gpu)
  assert lib.sfm_xcorr_workspace_bytes(C.byref(d)) == 0
  assert b'beyond the hand-written' in lib.sfm_last_error()
  with pytest.raises(_abi.SofimaAmdError, match='hand-written'):
    ff.JAXMaskedXCorrWithStatsCalculator(method=_abi.XCORR_FFT).flow_field(
        img, img, shape, 1, batch_size=1)
  assert torch.cuda.memory_allocated(gpu) <= before + (1 << 20)
