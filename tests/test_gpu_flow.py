"""HIP flow-field path vs golden vectors, reference KATs and the oracle (-m gpu)."""
import numpy as np
import pytest

from oracle import flow_oracle
from tests.util import check_flow, check_sharpness, em_texture

pytestmark = pytest.mark.gpu

METHODS = [1, 0]  # SFM_XCORR_DIRECT, SFM_XCORR_AUTO (MFMA where eligible)
# Sharpness of the exact-integer matrix-core kernel against the float32
# direct-sum kernel (NOT an oracle comparison): the direct kernel's 25 600-term
# float32 sums carry ~1e-5 x |peak| of rounding in the window minimum, so the
# reciprocal criterion (window minimum to 2e-5 x |peak|) applies everywhere.  A
# sign flip cannot pass below |sharpness| = 5e4 (|1/got - 1/want| would be > 2e-5).
VS_F32_KERNEL = dict(inv_atol=2e-5, inv_from=0.0)


# -- the reference's own known-answer tests (tests/flow_field_test.py) -------
@pytest.mark.parametrize('method', METHODS)
def test_kat_delta_images(gpu, method):
  from sofima_amd import flow_field
  pre_image = np.zeros((120, 120), dtype=np.uint8)
  post_image = np.zeros((120, 120), dtype=np.uint8)
  pre_image[60, 60] = 255
  post_image[70, 53] = 255
  calculator = flow_field.JAXMaskedXCorrWithStatsCalculator(method=method)
  field = calculator.flow_field(pre_image, post_image, patch_size=80, step=40,
                                batch_size=4)
  np.testing.assert_array_equal([4, 2, 2], field.shape)
  np.testing.assert_array_equal(7 * np.ones((2, 2)), field[0, ...])
  np.testing.assert_array_equal(-10 * np.ones((2, 2)), field[1, ...])
  np.testing.assert_array_equal(np.zeros((2, 2)), field[3, ...])

  post_image[54, 68] = 255
  post_image_mask = np.zeros((128, 128), dtype=bool)
  post_image_mask[:55, :70] = 1
  field = calculator.flow_field(pre_image, post_image, patch_size=80, step=40,
                                post_mask=post_image_mask, batch_size=4)
  np.testing.assert_array_equal([4, 2, 2], field.shape)
  np.testing.assert_array_equal(7 * np.ones((2, 2)), field[0, ...])
  np.testing.assert_array_equal(-10 * np.ones((2, 2)), field[1, ...])
  np.testing.assert_array_equal(np.zeros((2, 2)), field[3, ...])


def test_kat_3d(gpu):
  from sofima_amd import flow_field
  pre_image = np.zeros((50, 100, 100), dtype=np.uint8)
  post_image = np.zeros((50, 100, 100), dtype=np.uint8)
  pre_image[25, 50, 50] = 255
  post_image[22, 45, 54] = 255
  calculator = flow_field.JAXMaskedXCorrWithStatsCalculator()
  flow = calculator.flow_field(pre_image, post_image, patch_size=(40, 80, 80),
                               step=10, batch_size=1)
  np.testing.assert_array_equal([5, 2, 3, 3], flow.shape)
  np.testing.assert_array_equal(np.full([2, 3, 3], -4), flow[0, ...])
  np.testing.assert_array_equal(np.full([2, 3, 3], 5), flow[1, ...])
  np.testing.assert_array_equal(np.full([2, 3, 3], 3), flow[2, ...])


def test_kat_peak(gpu):
  from sofima_amd import flow_field
  hy, hx = np.mgrid[:50, :50]
  cy, cx = 20, 28
  hy = cy - hy
  hx = cx - hx
  r = np.sqrt(2 * hx**2 + hy**2)
  peak_max = 10
  xcorr = peak_max * np.exp(-r / 4)
  peaks = flow_field._batched_peaks(xcorr[np.newaxis, ...], (25, 25),
                                    min_distance=2, threshold_rel=0.5,
                                    peak_radius=(2, 3))
  np.testing.assert_array_equal([1, 4], peaks.shape)
  x32 = xcorr.astype(np.float32)
  peak_support = np.min(x32[cy - 2:cy + 3, cx - 3:cx + 4])
  assert peaks[0, 0] == 3
  assert peaks[0, 1] == -5
  assert peaks[0, 2] == np.float32(peak_max) / peak_support
  assert peaks[0, 3] == 0


@pytest.mark.parametrize('method', METHODS)
def test_kat_post_targeting(gpu, method):
  from sofima_amd import flow_field
  pre_image = np.zeros((120, 120), dtype=np.uint8)
  post_image = np.zeros((120, 120), dtype=np.uint8)
  pre_image[50, 55] = 255
  post_image[100, 100] = 255
  calculator = flow_field.JAXMaskedXCorrWithStatsCalculator(method=method)
  field = calculator.flow_field(pre_image, post_image, patch_size=80, step=40,
                                batch_size=4)
  np.testing.assert_array_equal(np.isnan(field[:, 0, 0]), True)
  post_targeting_field = np.full((2, 2, 2), 40.0, dtype=np.float32)
  field = calculator.flow_field(pre_image, post_image, patch_size=80, step=40,
                                batch_size=4,
                                post_targeting_field=post_targeting_field,
                                post_targeting_step=40)
  np.testing.assert_array_equal([4, 2, 2], field.shape)
  np.testing.assert_array_equal(-45 * np.ones((2, 2)), field[0, ...])
  np.testing.assert_array_equal(-50 * np.ones((2, 2)), field[1, ...])


# -- golden vectors ------------------------------------------------------------
def test_masked_xcorr_golden(gpu, golden):
  from sofima_amd import flow_field
  g = golden('xcorr_np')
  got = flow_field.masked_xcorr(g['a'], g['b'])
  scale = np.abs(g['unmasked']).max()
  np.testing.assert_allclose(got, g['unmasked'], atol=1e-5 * scale)
  got = flow_field.masked_xcorr(g['a'], g['b'], g['am'], g['bm'])
  np.testing.assert_allclose(got, g['masked'], atol=2e-5)
  got = flow_field.masked_xcorr(g['a3'], g['b3'], dim=3)
  scale = np.abs(g['unmasked3']).max()
  np.testing.assert_allclose(got, g['unmasked3'], atol=1e-5 * scale)
  got = flow_field.masked_xcorr(g['a3'], g['b3'], g['am3'], g['bm3'], dim=3)
  np.testing.assert_allclose(got, g['masked3'], atol=2e-5)
  # one-sided mask: float32 semantics (see oracle docstring on the 0.3 tie)
  got = flow_field.masked_xcorr(g['a'], g['b'], g['am'], None)
  want = flow_oracle.xcorr_surface(g['a'], g['b'], g['am'], None)
  np.testing.assert_allclose(got, want, atol=2e-5)


def test_batched_peaks_golden(gpu, golden):
  from sofima_amd import flow_field
  g = golden('peaks')
  got = flow_field._batched_peaks(g['imgs'], (15, 15), 2, 0.5, 5)
  np.testing.assert_array_equal(got, g['out_all'])
  got = flow_field._batched_peaks(g['imgs'][:1], (15, 15), 2, 0.5, 5)
  np.testing.assert_array_equal(got, g['out_0'])
  got = flow_field._batched_peaks(g['imgs'][5:], (15, 15), 2, 0.5, (2, 3))
  np.testing.assert_array_equal(got, g['out_r'])
  got = flow_field._batched_peaks(g['vol'], (4, 5, 6), 1, 0.5, (1, 2, 2))
  np.testing.assert_array_equal(got, g['out_3d'])


def test_peaks_plateau_overflow(gpu):
  """More equal-valued peaks than the candidate list holds -> rescan path."""
  from sofima_amd import flow_field
  img = np.full((2, 80, 80), 2.0, np.float32)
  img[1, 40, 41] = 3.0
  got = flow_field._batched_peaks(img, (40, 40), 2, 0.5, 5)
  want = flow_oracle.batched_peaks(img, (40, 40), 2, 0.5, 5)
  np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize('method', METHODS)
def test_flow_golden_2d(gpu, golden, method):
  from sofima_amd import flow_field
  g = golden('flow2d')
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator(method=method)
  pre, post = g['pre'], g['post']
  check_flow(calc.flow_field(pre, post, 48, 24, batch_size=8), g['plain'])
  check_flow(
      calc.flow_field(pre, post, 48, 24, pre_mask=g['pre_mask'],
                      post_mask=g['post_mask'], batch_size=8), g['masked'])
  check_flow(
      calc.flow_field(pre, post, 48, 24, pre_mask=g['pre_mask'],
                      post_mask=g['post_mask'],
                      mask_only_for_patch_selection=True, max_masked=0.5,
                      batch_size=16), g['masksel'])
  check_flow(
      calc.flow_field(pre, post, 48, 24, batch_size=8, post_patch_size=32),
      g['postpatch'])
  check_flow(
      calc.flow_field(pre, post, (48, 32), (24, 16),
                      selection_mask=np.pad(g['sel'], ((0, 0), (0, 4))),
                      batch_size=5), g['selected'])
  check_flow(
      calc.flow_field(pre, post, 48, 24, batch_size=8,
                      pre_targeting_field=g['tg_pre'], pre_targeting_step=48,
                      post_targeting_field=g['tg_post'],
                      post_targeting_step=64), g['targeted'])
  calc_m = flow_field.JAXMaskedXCorrWithStatsCalculator(
      mean=120.0, peak_min_distance=3, peak_radius=(3, 4), method=method)
  check_flow(calc_m.flow_field(g['pre_f'], g['post_f'], 40, 20, batch_size=64),
             g['float_mean'])


def test_flow_golden_3d(gpu, golden):
  from sofima_amd import flow_field
  g = golden('flow3d')
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  check_flow(calc.flow_field(g['pre'], g['post'], (16, 24, 24), 8,
                             batch_size=4), g['plain'])


# -- against the oracle at production patch geometry -----------------------------
def _em_pair(seed, h, w, shift=(3, -5), warp=0.0):
  from scipy import ndimage
  rng = np.random.default_rng(seed)
  m = 16
  base = ndimage.gaussian_filter(rng.standard_normal((h + 2 * m, w + 2 * m)), 2.0)
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  dy, dx = shift
  pre = base[m:m + h, m:m + w].copy()
  post = base[m + dy:m + dy + h, m + dx:m + dx + w].copy()
  if warp:
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    d = warp * np.sin(2 * np.pi * xx / 512) * np.cos(2 * np.pi * yy / 512)
    post = ndimage.map_coordinates(post.astype(np.float32), [yy + d, xx - d],
                                   order=1, mode='nearest')
    post = np.clip(np.round(post), 0, 255).astype(np.uint8)
  return pre, post


@pytest.mark.parametrize('method', METHODS)
def test_flow_production_geometry_vs_oracle(gpu, method):
  """patch 160 / step 40 (em_2d defaults), unmasked raw path, warped pair."""
  from sofima_amd import flow_field
  pre, post = _em_pair(5, 640, 720, warp=3.0)
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator(method=method)
  got = calc.flow_field(pre, post, 160, 40, batch_size=64)
  want = flow_oracle.flow_field(pre, post, 160, 40, batch_size=64, workers=4)
  check_flow(got, want)


def test_surface_u8_exact_vs_direct_oracle(gpu):
  """uint8 patches: the surface equals the exact integer correlation of the
  mean-subtracted patches to float32 rounding."""
  from sofima_amd import flow_field
  rng = np.random.default_rng(9)
  a = rng.integers(0, 256, (3, 96, 96)).astype(np.float32)
  b = rng.integers(0, 256, (3, 96, 96)).astype(np.float32)
  a0 = a - a.mean(axis=(1, 2), keepdims=True, dtype=np.float32)
  b0 = b - b.mean(axis=(1, 2), keepdims=True, dtype=np.float32)
  want = flow_oracle.xcorr_surface_direct(a0, b0)
  got = flow_field.masked_xcorr(a0, b0)
  np.testing.assert_allclose(got, want, atol=1e-5 * np.abs(want).max())


# -- int8 MFMA kernel against the general direct kernel over patch geometries ----
@pytest.mark.parametrize('py,px,qy,qx', [
    (17, 17, 17, 17), (33, 40, 33, 40), (48, 48, 32, 32), (50, 70, 50, 70),
    (64, 64, 64, 64), (100, 96, 60, 80), (127, 129, 127, 129),
    (160, 160, 160, 160), (160, 160, 96, 112), (200, 150, 200, 150),
])
def test_mfma_matches_direct_kernel(gpu, py, px, qy, qx):
  from sofima_amd import flow_field
  rng = np.random.default_rng(py * 1000 + px)
  from scipy import ndimage
  h, w = 420, 460
  base = ndimage.gaussian_filter(rng.standard_normal((h + 8, w + 8)), 1.5)
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  pre = base[4:4 + h, 4:4 + w].copy()
  post = base[6:6 + h, 1:1 + w].copy()
  post[::7, ::5] += 3                      # not an exact copy
  b = 13
  # starts include negative / overshooting values: clamped like dynamic_slice
  starts = np.stack([rng.integers(-20, h - py + 30, b),
                     rng.integers(-20, w - px + 30, b)], axis=1)
  post_starts = starts + np.array([(py - qy) // 2, (px - qx) // 2])
  kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5,
            post_patch_size=(qy, qx), post_starts=post_starts)
  for mean in (None, 117.5):
    ref = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), starts,
                                         mean, method=1, **kw)
    got = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), starts,
                                         mean, method=2, **kw)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_array_equal(got[:, :2], ref[:, :2])
    ok = np.isfinite(ref[:, 2])
    check_sharpness(got[ok, 2], ref[ok, 2], **VS_F32_KERNEL)
    np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=1e-3, atol=1e-6)


def test_mfma_extreme_contrast_and_flat_patches(gpu):
  """Full 0..255 range (centre clamped to 128) and constant patches."""
  from sofima_amd import flow_field
  rng = np.random.default_rng(5)
  pre = rng.integers(0, 256, (256, 256)).astype(np.uint8)
  post = np.roll(pre, (2, -3), axis=(0, 1)).copy()
  pre[:64, :64] = 0            # flat patches -> all-zero surface -> NaN
  post[:64, :64] = 255
  calc2 = flow_field.JAXMaskedXCorrWithStatsCalculator(method=2)
  calc1 = flow_field.JAXMaskedXCorrWithStatsCalculator(method=1)
  a = calc2.flow_field(pre, post, 64, 32, batch_size=16)
  b = calc1.flow_field(pre, post, 64, 32, batch_size=16)
  assert np.isnan(a[:, 0, 0]).all()
  np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
  np.testing.assert_array_equal(a[:2], b[:2])
  m = np.isfinite(b[0])
  assert (a[0][m] == 3).mean() > 0.9 and (a[1][m] == -2).mean() > 0.9


def test_flow_field_empty_selection_and_single_patch(gpu):
  from sofima_amd import flow_field
  rng = np.random.default_rng(6)
  img = rng.integers(0, 256, (96, 96)).astype(np.uint8)
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  sel = np.zeros((2, 2), bool)
  f = calc.flow_field(img, img, 64, 32, selection_mask=sel, batch_size=4)
  assert f.shape == (4, 2, 2) and np.isnan(f).all()
  f = calc.flow_field(img, img, 96, 96, batch_size=1)       # patch == image
  assert f.shape == (4, 1, 1)
  np.testing.assert_array_equal(f[:2, 0, 0], [0, 0])


# -- search-window geometry on the matrix cores (pre patch 161 .. 320 wide against a post
# patch of up to 160: processor/flow.py:577,792-803 calls flow_field(patch_size = 160 +
# 2 search_radius, post_patch_size = 160)) -------------------------------------------------
@pytest.mark.parametrize('py,px,qy,qx', [
    (240, 240, 160, 160), (320, 320, 160, 160), (192, 176, 160, 144), (320, 200, 120, 160),
    (200, 320, 160, 96), (161, 161, 160, 160),
])
def test_search_window_mfma_matches_direct_kernel(gpu, py, px, qy, qx):
  """The wide variants of the int8 kernel (one wave per SIMD, accumulators in the upper
  register file) against the float direct kernel: identical peaks, statistics within the
  float kernel's tolerance; starts that overshoot the image on every side (clamped like
  lax.dynamic_slice, flow_field.py:320-325), per-patch and fixed means."""
  from sofima_amd import flow_field
  rng = np.random.default_rng(py * 1000 + px)
  from scipy import ndimage
  h, w = 520, 560
  base = ndimage.gaussian_filter(rng.standard_normal((h + 8, w + 8)), 1.5)
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  pre = base[4:4 + h, 4:4 + w].copy()
  post = base[6:6 + h, 1:1 + w].copy()
  post[::7, ::5] += 3
  b = 21
  starts = np.stack([rng.integers(-20, h - py + 30, b),
                     rng.integers(-20, w - px + 30, b)], axis=1)
  post_starts = starts + np.array([(py - qy) // 2, (px - qx) // 2]) + rng.integers(-9, 10, (b, 2))
  kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5,
            post_patch_size=(qy, qx), post_starts=post_starts)
  for mean in (None, 117.5):
    ref = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), starts,
                                         mean, method=1, **kw)
    got = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), starts,
                                         mean, method=2, **kw)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_array_equal(got[:, :2], ref[:, :2])
    ok = np.isfinite(ref[:, 2])
    # (the float kernel sums up to 102 400 products per output here, 4 x the 160^2 case the
    # VS_F32_KERNEL criterion was sized on: window minimum to 8e-5 x |peak|.  The oracle
    # comparison of the same kernel is test_search_window_flow_vs_oracle.)
    check_sharpness(got[ok, 2], ref[ok, 2], inv_atol=8e-5, inv_from=0.0)
    np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize('patch', [240, 320])
def test_search_window_flow_vs_oracle(gpu, patch):
  """flow_field(patch_size = 160 + 2 search_radius, post_patch_size = 160) -- the call of
  EstimateMissingFlow -- on whole reference batches (5 of 64 + a ragged one) against the
  oracle, matrix-core form (the automatic choice) and FFT form; and two batches with
  clamped starts through batched_xcorr_peaks."""
  from sofima_amd import flow_field
  pre, post = _em_pair(29, 900, 980, shift=(7, -11), warp=2.0)
  kw = dict(patch_size=patch, step=40, post_patch_size=160, batch_size=64)
  want = flow_oracle.flow_field(pre, post, workers=8, **kw)
  assert np.isfinite(want[:2]).mean() > 0.9 and want.shape[1] * want.shape[2] > 2 * 64
  for method in (0, 2, 3):
    got = flow_field.JAXMaskedXCorrWithStatsCalculator(method=method).flow_field(pre, post, **kw)
    check_flow(got, want)
  # clamped starts, two whole batches (batch coupling as in the reference)
  rng = np.random.default_rng(patch)
  b = 128
  starts = np.stack([rng.integers(-40, 900 - patch + 40, b),
                     rng.integers(-40, 980 - patch + 40, b)], axis=1)
  post_starts = starts + (patch - 160) // 2
  for lo in (0, 64):
    sl = slice(lo, lo + 64)
    w_pk = flow_oracle.batched_xcorr_peaks(pre, post, None, None, (patch, patch), starts[sl], None,
                                           2, 0.5, 5, (160, 160), post_starts[sl], workers=8)
    g_pk = flow_field.batched_xcorr_peaks(pre, post, None, None, (patch, patch), starts[sl], None,
                                          min_distance=2, threshold_rel=0.5, peak_radius=5,
                                          post_patch_size=(160, 160), post_starts=post_starts[sl],
                                          method=2)
    np.testing.assert_array_equal(np.isnan(g_pk), np.isnan(w_pk))
    np.testing.assert_array_equal(g_pk[:, :2], w_pk[:, :2])
    np.testing.assert_allclose(g_pk[:, 3], w_pk[:, 3], rtol=1e-4, atol=1e-6)
    ok = np.isfinite(w_pk[:, 2])
    check_sharpness(g_pk[ok, 2], w_pk[ok, 2])


@pytest.mark.parametrize('patch,post_patch', [((80, 80), (160, 160)), ((81, 64), (160, 150)),
                                              ((160, 160), (80, 96))])
def test_single_cell_grid_with_unequal_patch_sizes_vs_oracle(gpu, patch, post_patch):
  """ONE grid cell (the call pattern of whole-overlap correlations) with a post
  patch larger / smaller than the pre patch: the pre start is clip(0 - (patch -
  post_patch) // 2, 0) (flow_field.py:601-602, :621-622) -- +40 for 80 vs 160 --
  on the single-patch fast path of device_plan as on the general one."""
  from sofima_amd import flow_field
  pre, post = _em_pair(17, 170, 180, shift=(2, -3))
  post = post[:post_patch[0] + 5, :post_patch[1] + 7].copy()
  kw = dict(patch_size=patch, step=(400, 400), post_patch_size=post_patch, batch_size=4)
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  got = calc.flow_field(pre, post, **kw)
  assert got.shape == (4, 1, 1)
  want = flow_oracle.flow_field(pre, post, **kw)
  check_flow(got, want)
  assert np.isfinite(want[:2]).all()
  # the general plan (a selection mask disables the fast path) gives the same
  sel = np.ones((1, 1), bool)
  np.testing.assert_array_equal(calc.flow_field(pre, post, selection_mask=sel, **kw), got)


@pytest.mark.parametrize('py,px,qy,qx', [(48, 48, 48, 48), (64, 80, 40, 64),
                                         (160, 160, 160, 160)])
def test_masked_mfma_matches_direct_kernel(gpu, py, px, qy, qx):
  """Padfield NCC from eight int8 MFMA correlations vs the f32 direct kernel,
  including full-range pixels (a - centre = -128 exercises the square split)."""
  from sofima_amd import flow_field
  rng = np.random.default_rng(py + 7 * qx)
  from scipy import ndimage
  h, w = 400, 420
  base = ndimage.gaussian_filter(rng.standard_normal((h + 8, w + 8)), 1.5)
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  pre = base[4:4 + h, 4:4 + w].copy()
  post = base[6:6 + h, 1:1 + w].copy()
  pre[10:14, 10:300] = 0
  pre[14:18, 10:300] = 255
  pre_mask = rng.random((h, w)) < 0.1
  pre_mask[200:260, 100:220] = True
  post_mask = np.zeros((h + 6, w + 3), bool)      # larger than the image
  post_mask[50:120, 300:] = True
  post_mask[rng.random(post_mask.shape) < 0.05] = True
  b = 9
  starts = np.stack([rng.integers(0, h - py, b), rng.integers(0, w - px, b)], 1)
  post_starts = starts + np.array([(py - qy) // 2, (px - qx) // 2])
  kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5,
            post_patch_size=(qy, qx), post_starts=post_starts)
  for masks in ((pre_mask, post_mask), (None, post_mask), (pre_mask, None)):
    ref = flow_field.batched_xcorr_peaks(pre, post, masks[0], masks[1], (py, px),
                                         starts, None, method=1, **kw)
    got = flow_field.batched_xcorr_peaks(pre, post, masks[0], masks[1], (py, px),
                                         starts, None, method=2, **kw)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_array_equal(got[:, :2], ref[:, :2])
    ok = np.isfinite(ref[:, 2])
    check_sharpness(got[ok, 2], ref[ok, 2], **VS_F32_KERNEL)
    np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=1e-4, atol=1e-6)


def test_masked_flow_production_geometry_vs_oracle(gpu):
  from sofima_amd import flow_field
  pre, post = _em_pair(11, 520, 560, warp=2.0)
  rng = np.random.default_rng(2)
  pre_mask = np.zeros(pre.shape, bool)
  pre_mask[100:180, 200:330] = True
  post_mask = rng.random(post.shape) < 0.03
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  got = calc.flow_field(pre, post, 160, 40, pre_mask=pre_mask,
                        post_mask=post_mask, batch_size=32)
  want = flow_oracle.flow_field(pre, post, 160, 40, pre_mask=pre_mask,
                                post_mask=post_mask, batch_size=32, workers=4)
  check_flow(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize('shape,patch,step', [
    ((300, 517), (160, 160), (40, 40)),
    ((97, 64), (17, 32), (5, 8)),
    ((9, 40, 50), (4, 16, 16), (2, 8, 8)),
    ((33, 33), (33, 33), (7, 7)),
    ((40, 9000), (16, 160), (8, 40)),      # several column tiles per row of patches
    ((20, 5000), (8, 3000), (4, 500)),     # patch wider than a tile
    ((300, 70), (280, 33), (5, 3)),        # more rows than a packed byte counter holds
    ((4096, 300), (4096, 300), (1, 1)),    # whole-overlap strip: one output, rows split
    ((300, 5000), (290, 4990), (4, 6)),    # wide whole-overlap patches: column segments
])
def test_mask_patch_counts_match_summed_area_table(gpu, shape, patch, step):
  """sfm_mask_patch_counts == flow_field.py:575-589 box query, exactly."""
  from sofima_amd import flow_field as ff
  rng = np.random.default_rng(7)
  mask = rng.random(shape) < (0.97 if shape[0] == 300 else 0.3)
  want = ff._host_masked_counts(mask, patch, step)
  got = ff._masked_counts(mask, patch, step)
  assert got.shape == want.shape
  np.testing.assert_array_equal(got.astype(np.int64), want.astype(np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize('method,masked,dtype', [
    (0, False, np.uint8), (1, False, np.uint8), (0, True, np.uint8),
    (0, False, np.float32),
])
def test_many_batches_per_call_equal_batch_by_batch(gpu, monkeypatch, method,
                                                    masked, dtype):
  """`group` keeps the reference's per-batch coupling inside one launch.

  Calls that carry several reference batches (SfmXcorrDesc.group) must give
  bit-identical fields to one call per batch, including the batch-coupled
  second-peak suppression (Q1) and masked tolerances (Q2).
  """
  from sofima_amd import flow_field as ff
  rng = np.random.default_rng(11)
  base = rng.integers(0, 256, (260, 300)).astype(np.float32)
  from scipy import ndimage
  base = ndimage.gaussian_filter(base, 1.5)
  base = ((base - base.min()) / (base.max() - base.min()) * 255)
  pre = base[4:244, 6:286].astype(dtype)
  post = base[1:241, 9:289].astype(dtype)
  kw = {}
  if masked:
    kw = dict(pre_mask=rng.random(pre.shape) < 0.05,
              post_mask=rng.random(post.shape) < 0.05)
  calc = ff.JAXMaskedXCorrWithStatsCalculator(method=method)
  monkeypatch.setattr(ff, 'LAUNCH_PATCHES', 1)
  one = calc.flow_field(pre, post, 48, 8, batch_size=37, **kw)
  monkeypatch.setattr(ff, 'LAUNCH_PATCHES', 37 * 5)
  five = calc.flow_field(pre, post, 48, 8, batch_size=37, **kw)
  monkeypatch.setattr(ff, 'LAUNCH_PATCHES', 1 << 20)
  every = calc.flow_field(pre, post, 48, 8, batch_size=37, **kw)
  assert one.shape[1] * one.shape[2] > 37 * 6
  np.testing.assert_array_equal(one, five)
  np.testing.assert_array_equal(one, every)


# -- FFT form (3-D and large patches) ---------------------------------------------
FFT = 3


@pytest.mark.gpu
def test_fft_flow_golden_2d_and_3d(gpu, golden):
  """The hipFFT form reproduces the reference goldens like the direct kernel."""
  from sofima_amd import flow_field
  g = golden('flow3d')
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator(method=FFT)
  check_flow(calc.flow_field(g['pre'], g['post'], (16, 24, 24), 8, batch_size=4),
             g['plain'])
  g = golden('flow2d')
  # same batch size as the golden: the peak ratio depends on batch membership
  check_flow(calc.flow_field(g['pre'], g['post'], 48, 24, batch_size=8), g['plain'])
  check_flow(calc.flow_field(g['pre'], g['post'], 48, 24, pre_mask=g['pre_mask'],
                             post_mask=g['post_mask'], batch_size=8), g['masked'])


@pytest.mark.gpu
@pytest.mark.parametrize('masked', [False, True])
@pytest.mark.parametrize('shape,p,q', [((70, 90), (48, 64), (48, 64)),
                                       ((70, 90), (50, 33), (31, 20)),
                                       ((20, 30, 34), (12, 20, 16), (12, 20, 16)),
                                       ((20, 30, 34), (9, 14, 21), (5, 14, 10))])
def test_fft_surface_matches_direct_kernel(gpu, shape, p, q, masked):
  """masked_xcorr: FFT form == shift-by-shift kernel to float32 FFT accuracy."""
  from sofima_amd import flow_field
  rng = np.random.default_rng(17)
  b = 3
  prev = rng.integers(0, 256, (b,) + p).astype(np.float32)
  curr = rng.integers(0, 256, (b,) + q).astype(np.float32)
  prev -= prev.mean(axis=tuple(range(1, prev.ndim)), keepdims=True)
  curr -= curr.mean(axis=tuple(range(1, curr.ndim)), keepdims=True)
  kw = {}
  if masked:
    kw = dict(prev_mask=rng.random(prev.shape) < 0.1, curr_mask=rng.random(curr.shape) < 0.1)
  dim = len(p)
  want = flow_field.masked_xcorr(prev, curr, dim=dim, method=1, **kw)
  got = flow_field.masked_xcorr(prev, curr, dim=dim, method=FFT, **kw)
  assert got.shape == want.shape
  if masked:
    np.testing.assert_allclose(got, want, atol=2e-4)
  else:
    np.testing.assert_allclose(got, want, atol=2e-6 * np.abs(want).max())


@pytest.mark.gpu
def test_fft_flow_3d_production_patch_vs_oracle(gpu):
  """cfg-5 geometry (80^3 patches, step 40): exact shift recovered, oracle parity."""
  from scipy import ndimage
  from sofima_amd import flow_field
  rng = np.random.default_rng(23)
  base = ndimage.gaussian_filter(rng.standard_normal((140, 140, 140)), 1.5)
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  pre = base[8:128, 8:128, 8:128]
  post = base[10:130, 5:125, 12:132]
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()   # AUTO -> FFT for 80^3
  f = calc.flow_field(pre, post, (80, 80, 80), 40, batch_size=8)
  assert f.shape == (5, 2, 2, 2)
  # flow = position in pre - position in post of the same content, as (x, y, z)
  np.testing.assert_array_equal(f[0], 4)
  np.testing.assert_array_equal(f[1], -3)
  np.testing.assert_array_equal(f[2], 2)
  want = flow_oracle.flow_field(pre, post, (80, 80, 80), 40, batch_size=8)
  check_flow(f, want)


@pytest.mark.gpu
@pytest.mark.parametrize('shape,center,radius', [((3, 70, 72, 68), (30, 35, 33), (2, 3, 3)),
                                                 ((4, 610, 590), (300, 290), 5)])
def test_batched_peaks_large_surfaces_vs_oracle(gpu, shape, center, radius):
  """Surfaces >= 2^18 elements take the multi-workgroup first pass: same result."""
  from scipy import ndimage
  from sofima_amd import flow_field
  rng = np.random.default_rng(31)
  sig = (0,) + (2.5,) * (len(shape) - 1)
  img = ndimage.gaussian_filter(rng.standard_normal(shape), sig).astype(np.float32)
  img[1] = -np.abs(img[1])                  # no positive value: NaN row
  img[2].flat[0] = img[2].max() * 3         # peak at flat index 0 (Q4)
  got = flow_field._batched_peaks(img, center, 2, 0.5, radius)
  want = flow_oracle.batched_peaks(img, center, 2, 0.5, radius)
  np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
  nd = len(shape) - 1
  np.testing.assert_array_equal(got[:, :nd], want[:, :nd])
  ok = np.isfinite(want[:, nd])
  np.testing.assert_allclose(got[ok, nd], want[ok, nd], rtol=1e-5)
  ok = np.isfinite(want[:, nd + 1])
  np.testing.assert_allclose(got[ok, nd + 1], want[ok, nd + 1], rtol=1e-5)


@pytest.mark.gpu
def test_whole_overlap_offset_regime_vs_oracle(gpu):
  """The call pattern of stitch_rigid._estimate_offset (stitch_rigid.py:38-66):
  one patch = the whole overlap strip, dynamic-range masks, step 1, batch 1."""
  from scipy import ndimage
  from sofima_amd import flow_field
  rng = np.random.default_rng(2)
  base = ndimage.gaussian_filter(rng.standard_normal((1100, 260)), 2.0)
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  a = base[20:1044, 20:220]
  b = base[31:1055, 14:214]
  a_mask = (ndimage.maximum_filter(a, 10) - ndimage.minimum_filter(a, 10)) < 65
  b_mask = (ndimage.maximum_filter(b, 10) - ndimage.minimum_filter(b, 10)) < 65
  assert 0.02 < a_mask.mean() < 0.9
  mfc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  got = mfc.flow_field(a, b, pre_mask=a_mask, post_mask=b_mask, patch_size=a.shape,
                       step=(1, 1), batch_size=1)
  want = flow_oracle.flow_field(a, b, a.shape, (1, 1), pre_mask=a_mask,
                                post_mask=b_mask, batch_size=1)
  xo, yo, _, pr = got.squeeze()
  assert (xo, yo) == (-6.0, 11.0)
  check_flow(got, want)


@pytest.mark.gpu
def test_full_size_bench_workload_properties(gpu):
  """BASELINE configs[1] at full size (8192^2 pair, 40401 patches of 160^2).

  Size-independent properties: a rigid integer content shift (plus noise) is
  recovered EXACTLY by every patch; the field does not depend on how many
  reference batches share a launch; a sample of patches agrees with the
  oracle's FFT form.
  """
  from bench import synth_pair
  from sofima_amd import flow_field as ff
  pre, post = synth_pair(8192, 1002, shift=(3, -5))
  calc = ff.JAXMaskedXCorrWithStatsCalculator()
  f = calc.flow_field(pre, post, 160, 40, batch_size=1024)
  assert f.shape == (4, 201, 201)
  assert not np.isnan(f).any()
  np.testing.assert_array_equal(f[0], -5.0)   # x: pre - post position of the content
  np.testing.assert_array_equal(f[1], 3.0)
  assert np.isfinite(f[2]).all()
  old = ff.LAUNCH_PATCHES
  try:
    ff.LAUNCH_PATCHES = 1
    g = calc.flow_field(pre, post, 160, 40, batch_size=1024)
  finally:
    ff.LAUNCH_PATCHES = old
  np.testing.assert_array_equal(f, g)
  # oracle on a corner crop with the same batch membership for its 64 patches
  crop = (slice(0, 160 + 7 * 40), slice(0, 160 + 7 * 40))
  fc = calc.flow_field(pre[crop], post[crop], 160, 40, batch_size=64)
  want = flow_oracle.flow_field(pre[crop], post[crop], 160, 40, batch_size=64, workers=8)
  check_flow(fc, want)


def test_full_size_masked_pair_properties(gpu, monkeypatch):
  """BASELINE configs[1] at full size with masks used IN the correlation
  (mask_only_for_patch_selection=False): discs on both images, so the batch
  holds patches of all four classes (clean / pre / post / both masked).
  Size-independent properties: the rigid content shift is recovered exactly by
  every patch that is not dropped, and the class shortcuts give bit-identical
  fields to running all eight passes for every patch (on a 2048^2 crop with
  the same masks; the full pair takes the shortcuts)."""
  from bench import synth_pair
  from sofima_amd import flow_field as ff
  pre, post = synth_pair(8192, 1002, shift=(3, -5))
  rng = np.random.default_rng(77)
  yy, xx = np.mgrid[-90:91, -90:91]
  disc = yy ** 2 + xx ** 2 <= 90 ** 2
  masks = []
  for _ in range(2):
    m = np.zeros(pre.shape, bool)
    for _ in range(120):
      y, x = rng.integers(90, 8192 - 91, 2)
      m[y - 90:y + 91, x - 90:x + 91] |= disc
    masks.append(m)
  calc = ff.JAXMaskedXCorrWithStatsCalculator()
  kw = dict(batch_size=1024, mask_only_for_patch_selection=False)
  f = calc.flow_field(pre, post, 160, 40, pre_mask=masks[0], post_mask=masks[1], **kw)
  assert f.shape == (4, 201, 201)
  ok = ~np.isnan(f[0])
  assert ok.mean() > 0.95
  # (a patch that is mostly masked can lock onto a few-pixel overlap: that is
  # the method, the reference does the same; the property is stated for
  # patches with less than a quarter of their pixels masked on either side)
  frac = np.maximum(ff._masked_counts(masks[0], (160, 160), (40, 40)),
                    ff._masked_counts(masks[1], (160, 160), (40, 40))) / 160.0 ** 2
  light = ok & (frac < 0.25)
  assert light.mean() > 0.85
  np.testing.assert_array_equal(f[0][light], -5.0)
  np.testing.assert_array_equal(f[1][light], 3.0)
  assert ((f[0][ok] == -5.0) & (f[1][ok] == 3.0)).mean() > 0.99
  assert np.isfinite(f[2][light]).all()   # (sharpness of an NCC surface may be negative)
  c = (slice(3000, 5048), slice(2000, 4048))
  args = (pre[c], post[c], 160, 40)
  ckw = dict(pre_mask=masks[0][c], post_mask=masks[1][c], **kw)
  fast = calc.flow_field(*args, **ckw)
  monkeypatch.setenv('SFM_MASKED_FAST', '0')
  eight = calc.flow_field(*args, **ckw)
  monkeypatch.delenv('SFM_MASKED_FAST')
  np.testing.assert_array_equal(fast, eight)
  cnt = ff._masked_counts(masks[0][c], (160, 160), (40, 40)) > 0
  cnt2 = ff._masked_counts(masks[1][c], (160, 160), (40, 40)) > 0
  # all four classes occur in the crop
  assert (cnt & cnt2).any() and (cnt & ~cnt2).any() and (~cnt & cnt2).any() and (~cnt & ~cnt2).any()


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(8))
def test_mfma_random_geometry_fuzz(gpu, seed):
  """Random patch / post-patch sizes (odd, non multiples of 16, P != Q), image
  sizes and starts that touch every image border: int8 MFMA == direct kernel."""
  from scipy import ndimage
  from sofima_amd import flow_field
  rng = np.random.default_rng(1000 + seed)
  py, px = int(rng.integers(17, 201)), int(rng.integers(17, 161))
  if py * px > 32768:
    py = 32768 // px
  qy, qx = int(rng.integers(9, py + 1)), int(rng.integers(9, px + 1))
  if rng.random() < 0.4:
    qy, qx = py, px
  h, w = py + int(rng.integers(0, 90)), px + int(rng.integers(0, 90))
  base = ndimage.gaussian_filter(rng.standard_normal((h + 8, w + 8)), 1.3)
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  pre = np.ascontiguousarray(base[4:4 + h, 4:4 + w])
  post = np.ascontiguousarray(base[5:5 + h, 2:2 + w])
  b = 9
  starts = np.stack([rng.integers(-5, h - py + 6, b), rng.integers(-5, w - px + 6, b)],
                    axis=1)
  starts[0] = (0, 0)
  starts[1] = (h - py, w - px)              # last rows / columns of the image
  post_starts = starts + np.array([(py - qy) // 2, (px - qx) // 2])
  kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5,
            post_patch_size=(qy, qx), post_starts=post_starts)
  ref = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), starts, None,
                                       method=1, **kw)
  got = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), starts, None,
                                       method=2, **kw)
  np.testing.assert_array_equal(np.isnan(got), np.isnan(ref), err_msg=str((py, px, qy, qx)))
  np.testing.assert_array_equal(got[:, :2], ref[:, :2], err_msg=str((py, px, qy, qx)))
  ok = np.isfinite(ref[:, 2])
  check_sharpness(got[ok, 2], ref[ok, 2], **VS_F32_KERNEL)
  np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(10))
def test_search_window_random_geometry_fuzz(gpu, seed):
  """Random search-window geometries on the wide variants of the int8 kernel -- pre patch
  161 .. 320 wide and 32 .. 320 tall (any parity, non multiples of 16), post patch up to 160
  wide and no larger than the pre patch, images barely larger than the patch, starts on and
  beyond every border (clamped), per-patch and fixed means, batches of 1 .. 40 -- against the
  oracle (float64 surfaces: the exact arbiter of near-equal candidates): vectors and NaN
  pattern identical, ratio to 1e-4, sharpness by SURVEY 8c's criterion; and the FFT form of
  the same call agrees with the matrix-core form on the vectors."""
  from scipy import ndimage
  from sofima_amd import flow_field
  rng = np.random.default_rng(7000 + seed)
  px = int(rng.integers(161, 321))
  py = int(rng.integers(32, 321))
  if py * px > 140 * 1024:
    py = (140 * 1024) // px
  qx = int(rng.integers(16, 161))
  qy = int(rng.integers(16, min(py, 160) + 1))
  h, w = py + int(rng.integers(0, 120)), px + int(rng.integers(0, 120))
  base = ndimage.gaussian_filter(rng.standard_normal((h + 12, w + 12)), float(rng.uniform(1.2, 2.5)))
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  dy, dx = int(rng.integers(-5, 6)), int(rng.integers(-5, 6))
  pre = np.ascontiguousarray(base[6:6 + h, 6:6 + w])
  post = np.ascontiguousarray(base[6 + dy:6 + dy + h, 6 + dx:6 + dx + w])
  b = int(rng.integers(1, 41))
  starts = np.stack([rng.integers(-8, h - py + 9, b), rng.integers(-8, w - px + 9, b)], axis=1)
  starts[0] = (0, 0)
  starts[-1] = (h - py, w - px)
  post_starts = starts + np.array([(py - qy) // 2, (px - qx) // 2])
  mean = None if seed % 2 == 0 else 120.0
  geo = str((py, px, qy, qx, h, w, b, mean))
  want = flow_oracle.batched_xcorr_peaks(pre, post, None, None, (py, px), starts, mean, 2, 0.5, 5,
                                         (qy, qx), post_starts, workers=8)
  kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5, post_patch_size=(qy, qx),
            post_starts=post_starts)
  got = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), starts, mean,
                                       method=2, **kw)
  fft = flow_field.batched_xcorr_peaks(pre, post, None, None, (py, px), starts, mean,
                                       method=3, **kw)
  np.testing.assert_array_equal(np.isnan(got), np.isnan(want), err_msg=geo)
  np.testing.assert_array_equal(got[:, :2], want[:, :2], err_msg=geo)
  np.testing.assert_allclose(got[:, 3], want[:, 3], rtol=1e-4, atol=1e-6, err_msg=geo)
  ok = np.isfinite(want[:, 2])
  check_sharpness(got[ok, 2], want[ok, 2])
  np.testing.assert_array_equal(fft[:, :2], got[:, :2], err_msg=geo)


@pytest.mark.gpu
def test_concurrent_python_threads(gpu):
  """The boundary is re-entrant: flow on the matrix cores, the FFT
  form and a mesh relaxation called from four threads at once give the results
  of the same calls made one after the other."""
  import threading
  from scipy import ndimage
  from sofima_amd import flow_field, mesh
  rng = np.random.default_rng(12)
  base = ndimage.gaussian_filter(rng.standard_normal((400, 420)), 1.5)
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  vol = ndimage.gaussian_filter(rng.standard_normal((60, 70, 72)), 1.2)
  vol = ((vol - vol.min()) / (vol.max() - vol.min()) * 255).astype(np.uint8)
  prev = (rng.standard_normal((2, 1, 60, 70)) * 3).astype(np.float32)
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40),
                               num_iters=50, max_iters=50, stop_v_max=1e-9, dt_max=100,
                               start_cap=0.1, final_cap=10.0)
  jobs = [
      lambda: flow_field.JAXMaskedXCorrWithStatsCalculator().flow_field(
          base[:384, :400], base[3:387, 5:405], 96, 24, batch_size=64),
      lambda: flow_field.JAXMaskedXCorrWithStatsCalculator().flow_field(
          base[8:392, 4:404], base[5:389, 9:409], 64, 16, batch_size=128),
      lambda: flow_field.JAXMaskedXCorrWithStatsCalculator(method=3).flow_field(
          vol[:56, :64, :64], vol[2:58, 3:67, 1:65], (32, 40, 40), 8, batch_size=4),
      lambda: np.array(mesh.relax_mesh(np.zeros_like(prev), prev, cfg)[0]),
  ]
  want = [j() for j in jobs]
  for _ in range(3):
    got = [None] * len(jobs)
    errs = []

    def run(i):
      try:
        got[i] = jobs[i]()
      except Exception as e:  # pylint: disable=broad-except
        errs.append(e)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
    for t in threads:
      t.start()
    for t in threads:
      t.join()
    assert not errs, errs
    for g, w in zip(got, want):
      np.testing.assert_array_equal(g, w)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(6))
def test_masked_paths_random_geometry_fuzz(gpu, seed):
  """Random sizes, P != Q, random masks (incl. a fully masked corner): the
  masked int8 MFMA path and the FFT form against the direct f32 kernel."""
  from scipy import ndimage
  from sofima_amd import flow_field
  rng = np.random.default_rng(2000 + seed)
  py, px = int(rng.integers(20, 150)), int(rng.integers(20, 150))
  qy, qx = (py, px) if rng.random() < 0.5 else (int(rng.integers(12, py + 1)),
                                                 int(rng.integers(12, px + 1)))
  b = 5
  base = ndimage.gaussian_filter(rng.standard_normal((b, py + 8, px + 8)), (0, 1.3, 1.3))
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  prev = np.ascontiguousarray(base[:, 4:4 + py, 4:4 + px]).astype(np.float32)
  curr = np.ascontiguousarray(base[:, 5:5 + qy, 3:3 + qx]).astype(np.float32)
  pm = rng.random(prev.shape) < rng.uniform(0.0, 0.2)
  cm = rng.random(curr.shape) < rng.uniform(0.0, 0.2)
  pm[0, :py // 3, :px // 3] = True
  # float64 FFT restatement of the reference as the arbiter: the exact-integer
  # MFMA path must be the closest, the f32 direct kernel and the f32 FFT form
  # carry float32 accumulation error of un-centred 8-bit data
  ref = flow_oracle.xcorr_surface(prev, curr, pm, cm, dtype=np.float64)
  for method, atol in ((2, 5e-5), (3, 2e-3), (1, 2e-3)):
    got = flow_field.masked_xcorr(prev.astype(np.uint8), curr.astype(np.uint8), pm, cm,
                                  method=method)
    assert got.shape == ref.shape
    # entries at the 0.3 max-overlap cut can flip between 0 and a value
    bad = ~np.isclose(got, ref, atol=atol, rtol=0)
    assert bad.mean() < 2e-3, (method, py, px, qy, qx, bad.mean(),
                               np.abs(got - ref)[bad].max() if bad.any() else 0)


@pytest.mark.gpu
@pytest.mark.parametrize('nd', [2, 3])
def test_device_plan_equals_host_plan(gpu, nd):
  """sfm_flow_starts (start coordinates, targeting lookups, clamping) and the
  device-side selection == the host planning of flow_field.py:557-680."""
  from sofima_amd import flow_field as ff
  rng = np.random.default_rng(40 + nd)
  for trial in range(6):
    shape = tuple(int(v) for v in rng.integers(60, 110, nd))
    if nd == 3:
      shape = (int(rng.integers(20, 30)),) + shape[1:]
    patch = tuple(int(v) for v in rng.integers(8, 20, nd))
    post_patch = patch if trial % 2 else tuple(max(4, p - int(rng.integers(0, 6)))
                                               for p in patch)
    if trial == 0:
      # post patch LARGER than the pre patch by an odd amount: the pre offset is
      # NumPy's floor division, -1 // 2 == -1 (flow_field.py:620)
      post_patch = tuple(p + 1 + 2 * int(rng.integers(0, 2)) for p in patch)
    step = tuple(int(v) for v in rng.integers(3, 9, nd))
    pre = rng.integers(0, 256, shape).astype(np.uint8)
    post = rng.integers(0, 256, shape).astype(np.uint8)
    kw = {}
    if trial >= 2:
      post_patch = patch      # mask grids of both sides must agree (as in the reference)
      kw['pre_mask'] = rng.random(shape) < 0.44
      kw['post_mask'] = rng.random(tuple(s + 3 for s in shape)) < 0.3
      kw['max_masked'] = 0.45
    if trial >= 3:
      tsh = tuple(int(v) for v in rng.integers(3, 7, nd))
      f = (rng.standard_normal((nd,) + tsh) * 25).astype(np.float32)
      f[:, tuple(0 for _ in tsh)] = np.nan
      kw['pre_targeting_field'], kw['pre_targeting_step'] = f, tuple(
          int(v) for v in rng.integers(10, 30, nd))
      kw['post_targeting_field'] = (rng.standard_normal((nd,) + tsh) * 25).astype(np.float32)
      kw['post_targeting_step'] = tuple(int(v) for v in rng.integers(10, 30, nd))
    if trial == 5:
      out_shape = (np.array(shape) - (np.array(post_patch) - step)) // step
      kw['selection_mask'] = rng.random(tuple(out_shape + 2)) < 0.7
    calc = ff.JAXMaskedXCorrWithStatsCalculator()
    host = calc.plan(shape, shape, patch, step, batch_size=7, post_patch_size=post_patch,
                     **kw)
    res = ff._Resident(pre, post, kw.get('pre_mask'), kw.get('post_mask'), gpu)
    dev = calc.device_plan(res, patch, step, kw.get('selection_mask'),
                           kw.get('max_masked', 0.75), 7, post_patch,
                           kw.get('pre_targeting_field'), kw.get('pre_targeting_step'),
                           kw.get('post_targeting_field'), kw.get('post_targeting_step'))
    n = dev['n']
    assert n == host['positions'].shape[0]
    seen = locals().get('seen', 0) + (n > 0)
    if n == 0:
      continue
    np.testing.assert_array_equal(dev['out_shape'], host['out_shape'])
    np.testing.assert_array_equal(dev['positions'][:n].cpu().numpy(), host['positions'])
    st = dev['starts'].cpu().numpy()
    np.testing.assert_array_equal(st[0], host['pre_starts'])
    np.testing.assert_array_equal(st[1], host['post_starts'])
    for dk, hk in (('tg', 'tg_offsets'), ('po', 'post_offsets')):
      assert (dev[dk] is None) == (host[hk] is None)
      if dev[dk] is not None:
        np.testing.assert_array_equal(dev[dk].cpu().numpy(), host[hk])
  assert seen >= 5


@pytest.mark.gpu
def test_flow_stays_on_device_through_clean_flow(gpu, golden):
  """flow_field(device_output=True) -> clean_flow without a host round trip ==
  the host path (and the reference's clean_flow on the golden flow)."""
  from oracle import flow_utils_oracle
  from sofima_amd import _dev, flow_field, flow_utils
  g = golden('flow2d')
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  host = calc.flow_field(g['pre'], g['post'], 48, 24, batch_size=8)
  devf = calc.flow_field(g['pre'], g['post'], 48, 24, batch_size=8, device_output=True)
  assert isinstance(devf, _dev.DeviceArray)
  np.testing.assert_array_equal(np.asarray(devf), host)
  cleaned = flow_utils.clean_flow(devf.tensor[:, None], 1.4, 1.4, 0, 5)
  want = flow_utils_oracle.clean_flow(host[:, None], 1.4, 1.4, 0, 5)
  np.testing.assert_array_equal(np.asarray(cleaned), want)


@pytest.mark.gpu
def test_sharpness_numerator_and_window_min_separately(gpu):
  """the sharpness quotient is ill-conditioned (window minimum near
  0 on raw surfaces), so its two ingredients are pinned separately on
  production-size patches: the peak value of the matrix-core surface to 1e-6
  relative and the minimum of the 11 x 11 window to 2e-6 of the peak, against
  the float64 evaluation of the same sums."""
  from sofima_amd import flow_field
  pre, post = _em_pair(5, 640, 720, warp=3.0)
  rng = np.random.default_rng(3)
  ys = rng.integers(0, 640 - 160, 6)
  xs = rng.integers(0, 720 - 160, 6)
  a = np.stack([pre[y:y + 160, x:x + 160] for y, x in zip(ys, xs)])
  b = np.stack([post[y:y + 160, x:x + 160] for y, x in zip(ys, xs)])
  got = flow_field.masked_xcorr(a, b, method=2, mean=None)   # int8 matrix cores
  a0 = a.astype(np.float64) - a.mean(axis=(1, 2), keepdims=True)
  b0 = b.astype(np.float64) - b.mean(axis=(1, 2), keepdims=True)
  want = flow_oracle.xcorr_surface(a0, b0, dtype=np.float64)
  for g, w in zip(got, want):
    iy, ix = np.unravel_index(np.argmax(w), w.shape)
    assert (iy, ix) == np.unravel_index(np.argmax(g), g.shape)
    peak = w[iy, ix]
    np.testing.assert_allclose(g[iy, ix], peak, rtol=1e-6)
    y0 = min(max(iy - 5, 0), w.shape[0] - 11)
    x0 = min(max(ix - 5, 0), w.shape[1] - 11)
    wg, ww = g[y0:y0 + 11, x0:x0 + 11], w[y0:y0 + 11, x0:x0 + 11]
    np.testing.assert_allclose(wg.min(), ww.min(), atol=2e-6 * peak)
    np.testing.assert_allclose(wg, ww, atol=2e-6 * peak)


def _masked_patch_batch(seed, b, py, px, qy, qx):
  """Textured uint8 patch batch; every third patch has no masked pixel, the
  others have masked pixels on the pre side, the post side or both."""
  from scipy import ndimage
  rng = np.random.default_rng(seed)
  base = ndimage.gaussian_filter(rng.standard_normal((b, py + 8, px + 8)), (0, 1.5, 1.5))
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  prev = base[:, 2:2 + py, 3:3 + px].copy()
  oy, ox = (py - qy) // 2, (px - qx) // 2
  curr = base[:, 4 + oy:4 + oy + qy, 1 + ox:1 + ox + qx].copy()
  prev[0, :3] = 0        # full-range pixels: a - centre = -128
  prev[0, 3:6] = 255
  pm = np.zeros(prev.shape, bool)
  cm = np.zeros(curr.shape, bool)
  for k in range(b):
    kind = k % 3           # 0: clean
    if kind == 0:
      continue
    if kind == 1 or k % 5 == 0:
      pm[k, py // 4:py // 2, px // 3:px // 3 + 9] = True
      pm[k][rng.random((py, px)) < 0.02] = True
    if kind == 2:
      cm[k, :qy // 5] = True
      cm[k][rng.random((qy, qx)) < 0.01] = True
  return prev, curr, pm, cm


@pytest.mark.parametrize('py,px,qy,qx', [(160, 160, 160, 160), (64, 80, 40, 64),
                                         (48, 48, 48, 48), (96, 96, 96, 80),
                                         (50, 46, 37, 31), (33, 3, 20, 2)])
def test_masked_clean_patch_shortcut_is_bit_identical(gpu, monkeypatch, py, px, qy, qx):
  """Patches without masked pixels take ONE matrix pass + box sums instead of
  eight passes: same surface, bit for bit, as the eight-pass form; both agree
  with the oracle."""
  from sofima_amd import flow_field
  prev, curr, pm, cm = _masked_patch_batch(py + qx, 13, py, px, qy, qx)
  monkeypatch.setenv('SFM_MASKED_FAST', '0')
  eight = flow_field.masked_xcorr(prev, curr, pm, cm, mean=None)
  monkeypatch.delenv('SFM_MASKED_FAST')
  fast = flow_field.masked_xcorr(prev, curr, pm, cm, mean=None)
  np.testing.assert_array_equal(fast, eight)
  assert np.abs(fast[0]).max() > 0.5       # a clean patch with a real peak
  a0 = prev.astype(np.float32) - np.array(
      [prev[k][~pm[k]].mean(dtype=np.float64) for k in range(len(prev))],
      np.float32)[:, None, None]
  b0 = curr.astype(np.float32) - np.array(
      [curr[k][~cm[k]].mean(dtype=np.float64) for k in range(len(curr))],
      np.float32)[:, None, None]
  want = flow_oracle.xcorr_surface(a0, b0, pm, cm, dtype=np.float64, workers=4)
  # elements whose denominator sits at the tolerance may flip to 0 on one side
  flipped = (fast == 0) != (want == 0)
  assert flipped.mean() < 1e-3, flipped.mean()
  bad = (np.abs(fast - want) > 3e-5) & ~flipped
  assert bad.mean() < 2e-5, (bad.mean(), np.abs(fast - want)[bad].max())
  # all patches clean / all dirty / one-sided mask arrays
  none = np.zeros_like(pm)
  for masks in ((none, np.zeros_like(cm)), (pm, None), (None, cm)):
    monkeypatch.setenv('SFM_MASKED_FAST', '0')
    eight = flow_field.masked_xcorr(prev, curr, masks[0], masks[1], mean=None)
    monkeypatch.delenv('SFM_MASKED_FAST')
    fast = flow_field.masked_xcorr(prev, curr, masks[0], masks[1], mean=None)
    np.testing.assert_array_equal(fast, eight)


@pytest.mark.parametrize('p', [64, 160, 48])
def test_masked_clean_pair_maxima_from_the_axes_are_exact(gpu, monkeypatch, p):
  """The batch maxima of clean same-size pairs come from the two shift axes plus
  the candidate cross product (masked_axis_max_kernel) instead of a sweep over
  every shift: same surfaces, bit for bit, as the sweep (SFM_MASKED_AXISMAX=0)
  and as the eight-pass form -- on textured patches (one candidate), a patch
  whose border rows / columns sit at the mean of the rest (several shifts come
  within 1e-5 of the zero shift), patches that are flat outside a small block
  (thousands of candidates: they take the sweep) and completely flat patches
  (every denominator zero)."""
  from sofima_amd import flow_field
  prev, curr, pm, cm = _masked_patch_batch(700 + p, 12, p, p, p, p)
  rng = np.random.default_rng(p)
  # clean patches are k % 3 == 0: 0, 3, 6, 9
  for arr in (prev, curr):
    blk = arr[6, p // 2 - 6:p // 2 + 6, p // 2 - 6:p // 2 + 6].copy()
    arr[6] = 77; arr[6, p // 2 - 6:p // 2 + 6, p // 2 - 6:p // 2 + 6] = blk   # flat but a block
    arr[9] = 200                                                 # completely flat
  # pair 3: the batch maximum of the denominator (contrast scaled up), with border rows /
  # columns AT the mean of the rest -- dropping them changes the sum of squared deviations
  # by less than 1e-5, so several shifts are candidates next to the zero shift
  # (tests/test_prune_bounds.py shows the same construction on the CPU)
  for arr in (prev, curr):
    inner = arr[3, 2:-3, 1:].astype(np.float64)
    inner = np.clip(np.round((inner - inner.mean()) * 3 + 128), 0, 255)
    arr[3, 2:-3, 1:] = inner.astype(np.uint8)
    m = np.uint8(np.round(inner.mean()))
    arr[3, :2] = m; arr[3, -3:] = m; arr[3, :, :1] = m
  runs = {}
  for name, env in (('axes', {}), ('sweep', {'SFM_MASKED_AXISMAX': '0'}),
                    ('eight', {'SFM_MASKED_FAST': '0'})):
    for k, v in env.items():
      monkeypatch.setenv(k, v)
    runs[name] = flow_field.masked_xcorr(prev, curr, pm, cm, mean=None)
    for k in env:
      monkeypatch.delenv(k)
  np.testing.assert_array_equal(runs['axes'], runs['sweep'])
  np.testing.assert_array_equal(runs['axes'], runs['eight'])
  assert np.abs(runs['axes'][0]).max() > 0.5
  # clean patches only (every maximum from the axes kernel), incl. an all-flat batch
  none_a, none_b = np.zeros_like(pm), np.zeros_like(cm)
  for sl in (slice(None), slice(9, 10)):
    monkeypatch.setenv('SFM_MASKED_AXISMAX', '0')
    sweep = flow_field.masked_xcorr(prev[sl], curr[sl], none_a[sl], none_b[sl], mean=None)
    monkeypatch.delenv('SFM_MASKED_AXISMAX')
    axes = flow_field.masked_xcorr(prev[sl], curr[sl], none_a[sl], none_b[sl], mean=None)
    np.testing.assert_array_equal(axes, sweep)


def test_masked_flow_mostly_clean_vs_oracle(gpu, monkeypatch):
  """flow_field with mask_only_for_patch_selection=False and a localised mask
  (most patches clean, the ones around the blob dirty) vs the oracle, and vs
  the eight-pass form (identical output)."""
  from sofima_amd import flow_field
  pre, post = _em_pair(21, 600, 640, warp=2.0)
  pre_mask = np.zeros(pre.shape, bool)
  pre_mask[250:300, 280:360] = True
  post_mask = np.zeros(post.shape, bool)
  post_mask[255:310, 270:350] = True
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  kw = dict(pre_mask=pre_mask, post_mask=post_mask, batch_size=64,
            mask_only_for_patch_selection=False)
  got = calc.flow_field(pre, post, 160, 40, **kw)
  monkeypatch.setenv('SFM_MASKED_FAST', '0')
  eight = calc.flow_field(pre, post, 160, 40, **kw)
  monkeypatch.delenv('SFM_MASKED_FAST')
  np.testing.assert_array_equal(got, eight)
  want = flow_oracle.flow_field(pre, post, 160, 40, workers=4, **kw)
  check_flow(got, want)


def test_masked_groups_per_round_do_not_change_the_field(gpu, monkeypatch):
  """The masked matrix-core path runs several reference batches per round of
  launches (maxima, overlap bounds and peak bitmaps are kept per batch):
  bit-identical to one batch per round, for ragged last batches and for a
  mean given by the caller."""
  from sofima_amd import flow_field
  pre, post = _em_pair(33, 700, 760, warp=2.0)
  rng = np.random.default_rng(5)
  pre_mask = np.zeros(pre.shape, bool)
  pre_mask[200:330, 250:420] = True
  pre_mask |= rng.random(pre.shape) < 0.002
  post_mask = np.zeros(post.shape, bool)
  post_mask[400:520, 100:300] = True
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  for bs, mean in ((16, None), (25, 120.0)):
    kw = dict(pre_mask=pre_mask, post_mask=post_mask, batch_size=bs,
              mask_only_for_patch_selection=False, max_masked=1.01)
    if mean is not None:
      calc = flow_field.JAXMaskedXCorrWithStatsCalculator(mean=mean)
    monkeypatch.setenv('SFM_MASKED_GROUPS', '1')
    one = calc.flow_field(pre, post, 160, 40, **kw)
    for n in ('3', '8'):
      monkeypatch.setenv('SFM_MASKED_GROUPS', n)
      np.testing.assert_array_equal(calc.flow_field(pre, post, 160, 40, **kw), one)
    monkeypatch.delenv('SFM_MASKED_GROUPS')
    np.testing.assert_array_equal(calc.flow_field(pre, post, 160, 40, **kw), one)
    # the peak sweep that skips row blocks without anything above its threshold
    monkeypatch.setenv('SFM_MASKED_BLKMAX', '0')
    np.testing.assert_array_equal(calc.flow_field(pre, post, 160, 40, **kw), one)
    monkeypatch.delenv('SFM_MASKED_BLKMAX')
    assert np.isfinite(one[:2]).mean() > 0.5


def _prune_images(kind, seed, h, w):
  """Image pairs that stress the tile pruning in different ways."""
  from scipy import ndimage
  rng = np.random.default_rng(seed)
  if kind == 'em':            # high NCC peak near the centre: most pruning
    return _em_pair(seed, h, w, shift=(3, -5), warp=2.0)
  if kind == 'far':           # true shift of 100+ px: the peak sits in an outer tile
    base = ndimage.gaussian_filter(rng.standard_normal((h + 260, w + 260)), 2.0)
    base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
    return (base[130:130 + h, 130:130 + w].copy(),
            base[130 + 118:130 + 118 + h, 130 - 97:130 - 97 + w].copy())
  if kind == 'periodic':      # many peaks of similar height all over the surface
    yy, xx = np.mgrid[:h, :w]
    tex = 100 + 60 * np.sin(2 * np.pi * yy / 23.0) * np.cos(2 * np.pi * xx / 31.0)
    pre = np.clip(tex + rng.normal(0, 6, tex.shape), 0, 255).astype(np.uint8)
    post = np.clip(np.roll(tex, (4, -7), (0, 1)) + rng.normal(0, 6, tex.shape), 0, 255)
    return pre, post.astype(np.uint8)
  if kind == 'smooth':        # broad peak: thousands of elements above half the
    base = ndimage.gaussian_filter(rng.standard_normal((h + 40, w + 40)), 30.0)  # maximum
    base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
    return base[20:20 + h, 20:20 + w].copy(), base[24:24 + h, 13:13 + w].copy()
  if kind == 'fine':          # period-3 lattice: thousands of local maxima above half
    yy, xx = np.mgrid[:h, :w]
    tex = 128 + 100 * np.sin(2 * np.pi * yy / 3.0) * np.sin(2 * np.pi * xx / 3.0)
    pre = np.clip(tex + rng.normal(0, 3, tex.shape), 0, 255).astype(np.uint8)
    post = np.clip(np.roll(tex, (1, 2), (0, 1)) + rng.normal(0, 3, tex.shape), 0, 255)
    return pre, post.astype(np.uint8)
  if kind == 'noise':         # unrelated images: low, scattered maxima, little pruning
    return (rng.integers(0, 256, (h, w)).astype(np.uint8),
            rng.integers(0, 256, (h, w)).astype(np.uint8))
  # 'edges': half the image flat, features only in a band (energies concentrated
  # in a few rows, so the bounds fall off abruptly)
  pre, post = _em_pair(seed, h, w, shift=(-6, 9))
  pre[: h // 2] = 90
  post[: h // 2 + 10] = 90
  pre[:, ::64] = 255
  return pre, post


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['em', 'far', 'periodic', 'noise', 'edges', 'smooth', 'fine'])
def test_pruned_correlation_tiles_are_bit_identical(gpu, monkeypatch, kind):
  """The fused-peaks kernel skips dy tiles whose Cauchy-Schwarz bound is below
  threshold_rel x the running maximum: results equal the un-pruned run bit for
  bit, for every patch size class, peak radius and min_distance."""
  from sofima_amd import flow_field
  h, w = 460, 500
  pre, post = _prune_images(kind, 31, h, w)
  rng = np.random.default_rng(5)
  for (py, px), radius, md, thr in [((160, 160), 5, 2, 0.5), ((96, 96), 5, 2, 0.5),
                                    ((64, 64), 12, 6, 0.5), ((50, 70), 5, 2, 0.3),
                                    ((160, 160), 30, 2, 0.9), ((128, 112), 3, 1, 0.5),
                                    ((160, 160), 5, 2, 0.2)]:  # (0.2: 'smooth' / 'fine' overflow the lists)
    b = 24
    starts = np.stack([rng.integers(-10, h - py + 10, b),
                       rng.integers(-10, w - px + 10, b)], axis=1)
    kw = dict(min_distance=md, threshold_rel=thr, peak_radius=radius,
              post_patch_size=(py, px), post_starts=starts)
    for mean in (None, 100.0):
      args = (pre, post, None, None, (py, px), starts, mean)
      pruned = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
      monkeypatch.setenv('SFM_MFMA_PRUNE', '0')
      full = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
      monkeypatch.delenv('SFM_MFMA_PRUNE')
      np.testing.assert_array_equal(pruned, full)
      # lazy surface stores (only tiles that may hold hot elements, and their guard
      # bands, reach memory) against the kernel that stores every computed tile,
      # with and without the pruning
      monkeypatch.setenv('SFM_MFMA_LAZY', '0')
      eager = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
      monkeypatch.setenv('SFM_MFMA_PRUNE', '0')
      eager_full = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
      monkeypatch.delenv('SFM_MFMA_PRUNE')
      monkeypatch.delenv('SFM_MFMA_LAZY')
      np.testing.assert_array_equal(pruned, eager)
      np.testing.assert_array_equal(pruned, eager_full)
      # tiles abandoned inside their row loop (proved cold by the energy of the rows
      # still to come): tested after every row group, and never
      for period in ('1', '0'):
        monkeypatch.setenv('SFM_MFMA_EARLY', period)
        np.testing.assert_array_equal(
            pruned, flow_field.batched_xcorr_peaks(*args, method=2, **kw))
      monkeypatch.delenv('SFM_MFMA_EARLY')
      monkeypatch.setenv('SFM_MFMA_WIDEN', '1')   # wider initial store requests
      np.testing.assert_array_equal(
          pruned, flow_field.batched_xcorr_peaks(*args, method=2, **kw))
      monkeypatch.delenv('SFM_MFMA_WIDEN')
      # row loops that drop provably cold outer column tiles in flight (default: down
      # to the four central tiles; 'far' and 'periodic' have their hot columns where
      # the previous patch does not predict them): never, and to the a-priori widths
      for widest in ('0', '4'):
        monkeypatch.setenv('SFM_MFMA_NARROW', widest)
        np.testing.assert_array_equal(
            pruned, flow_field.batched_xcorr_peaks(*args, method=2, **kw))
      monkeypatch.delenv('SFM_MFMA_NARROW')
      # (the hot-list / candidate overflow fall-backs sweep surfaces with pruned,
      # never stored tiles: 'smooth' and 'fine' take them at threshold 0.2.  The
      # lattice of 'fine' has many peaks of nearly equal height, which the float
      # direct kernel and the exact integer one may rank differently: no reference
      # comparison there)
      if ((py, px) == (96, 96) and kind != 'fine') or kind == 'smooth':
        ref = flow_field.batched_xcorr_peaks(*args, method=1, **kw)
        np.testing.assert_array_equal(np.isnan(pruned), np.isnan(ref))
        np.testing.assert_array_equal(pruned[:, :2], ref[:, :2])
        np.testing.assert_allclose(pruned[:, 3], ref[:, 3], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_abandoned_tiles_are_counted_and_save_matrix_instructions(gpu, monkeypatch):
  """SfmProfile.tiles_abandoned / mfma_issued: on an EM-like pair the in-loop cold
  test gives up row tiles and the kernel issues fewer matrix instructions than
  with SFM_MFMA_EARLY=0, for identical peak statistics."""
  import ctypes as C
  from sofima_amd import flow_field, _abi
  lib = _abi.load()
  pre, post = _prune_images('em', 7, 460, 500)
  rng = np.random.default_rng(2)
  starts = np.stack([rng.integers(0, 300, 64), rng.integers(0, 340, 64)], axis=1)
  kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5,
            post_patch_size=(160, 160), post_starts=starts)
  args = (pre, post, None, None, (160, 160), starts, None)

  def run():
    pf = _abi.SfmProfile()
    lib.sfm_profile_read(C.byref(pf))
    lib.sfm_profile_enable(1)
    try:
      out = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
      lib.sfm_profile_read(C.byref(pf))
    finally:
      lib.sfm_profile_enable(0)
    return out, int(pf.tiles_abandoned[0]), int(pf.mfma_issued[0]), int(pf.tiles_drawn[0])

  out_e, abandoned_e, issued_e, drawn_e = run()
  monkeypatch.setenv('SFM_MFMA_EARLY', '0')
  out_0, abandoned_0, issued_0, drawn_0 = run()
  np.testing.assert_array_equal(out_e, out_0)
  assert drawn_e == drawn_0 > 0
  assert abandoned_0 == 0 and abandoned_e > 0
  assert issued_e < issued_0


@pytest.mark.gpu
@pytest.mark.parametrize('py,px,qy,qx', [(160, 160, 160, 160), (96, 96, 96, 96),
                                         (64, 80, 40, 64), (50, 46, 50, 46)])
def test_masked_overlap_rule_skips_are_bit_identical(gpu, monkeypatch, py, px, qy, qx):
  """What the reference zeroes by its overlap rule (flow_field.py:151-155) is not
  computed: dead rows in the final assembly and in the peak sweeps, dead row /
  column tiles in the a' * b' pass (bounded through the valid-pixel counts).
  Surfaces and peak statistics equal the run without the skips, with a clean
  patch in the batch (threshold 0.3 Py Px), without one, and heavily masked."""
  from sofima_amd import flow_field
  prev, curr, pm, cm = _masked_patch_batch(3 * py + qx, 12, py, px, qy, qx)
  rng = np.random.default_rng(py)
  heavy_p, heavy_c = pm.copy(), cm.copy()
  heavy_p[:, : py // 2] = True                      # no clean patch, overlap maximum ~ half a patch
  heavy_c[rng.random(cm.shape) < 0.3] = True
  some_p = pm.copy()
  some_p[::3][:, 5:9, 5:9] = True                   # no clean patch, but nearly clean ones
  for masks in ((pm, cm), (some_p, cm), (heavy_p, heavy_c), (pm, None)):
    full_kw = dict(mean=None)
    fast = flow_field.masked_xcorr(prev, curr, masks[0], masks[1], **full_kw)
    monkeypatch.setenv('SFM_MASKED_DEADROWS', '0')
    ref = flow_field.masked_xcorr(prev, curr, masks[0], masks[1], **full_kw)
    monkeypatch.delenv('SFM_MASKED_DEADROWS')
    np.testing.assert_array_equal(fast, ref)
  # whole flow fields (peaks through the live-row sweeps), production patch size
  if (py, px) == (160, 160):
    pre, post = _em_pair(11, 420, 470, warp=2.0)
    m0 = np.zeros(pre.shape, bool)
    m1 = np.zeros(pre.shape, bool)
    m0[100:180, 200:330] = True
    m1[rng.random(pre.shape) < 0.01] = True
    calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
    kw = dict(batch_size=64, mask_only_for_patch_selection=False)
    for masks in ((m0, m1), (m0, None)):
      a = calc.flow_field(pre, post, 160, 40, pre_mask=masks[0], post_mask=masks[1], **kw)
      monkeypatch.setenv('SFM_MASKED_DEADROWS', '0')
      b = calc.flow_field(pre, post, 160, 40, pre_mask=masks[0], post_mask=masks[1], **kw)
      monkeypatch.delenv('SFM_MASKED_DEADROWS')
      np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_workspace_budget_only_changes_the_call_size(gpu):
  """LAUNCH_PATCHES is an upper bound: when the workspace of a whole pair does
  not fit the free-memory budget the call is cut into smaller ones (whole
  reference batches), and the field is the same."""
  from sofima_amd import flow_field as ff
  rng = np.random.default_rng(77)
  base = em_texture(rng, (420, 460))
  pre, post = base[8:400, 10:440].copy(), base[5:397, 14:444].copy()
  calc = ff.JAXMaskedXCorrWithStatsCalculator()
  want = calc.flow_field(pre, post, 48, 12, batch_size=32)
  calls = []
  orig = ff._abi.load().sfm_xcorr_peaks
  frac, small = ff.WORKSPACE_FRACTION, ff.SMALL_WORKSPACE
  try:
    ff.WORKSPACE_FRACTION = 1e-12          # nothing fits: one batch per call
    ff.SMALL_WORKSPACE = 0                 # (small calls are not exempt here)
    lib = ff._abi.load()

    class Spy:
      def __getattr__(self, name):
        if name == 'sfm_xcorr_peaks':
          def f(desc, out):
            calls.append(int(desc._obj.batch))
            return orig(desc, out)
          return f
        return getattr(lib, name)

    load = ff._abi.load
    ff._abi.load = lambda: Spy()
    try:
      got = calc.flow_field(pre, post, 48, 12, batch_size=32)
    finally:
      ff._abi.load = load
  finally:
    ff.WORKSPACE_FRACTION, ff.SMALL_WORKSPACE = frac, small
  assert len(calls) > 1 and set(calls) == {32}
  np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
def test_xcd_sharded_patch_queue_is_bit_identical(gpu):
  """SFM_MFMA_XCD=1 (one patch queue per XCD, prep blocks mapped to the same
  partitions; opt-in, measured without gain) only changes the processing order."""
  from sofima_amd import _abi, flow_field as ff
  rng = np.random.default_rng(78)
  base = em_texture(rng, (700, 760))
  pre, post = base[8:680, 10:740].copy(), base[5:677, 14:744].copy()
  calc = ff.JAXMaskedXCorrWithStatsCalculator()
  for kw in (dict(patch_size=96, step=24, batch_size=64),
             dict(patch_size=64, step=16, batch_size=37, post_patch_size=48)):
    want = calc.flow_field(pre, post, **kw)
    with _abi.option('SFM_MFMA_XCD', 1):
      got = calc.flow_field(pre, post, **kw)
    np.testing.assert_array_equal(got, want)
