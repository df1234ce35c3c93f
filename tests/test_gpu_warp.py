"""warp.warp_subvolume on the HIP path (-m gpu): the reference's known-answer
tests (tests/warp_test.py:27-82) and the oracle.  OpenCV is not installable
here: parity beyond the KATs is unpinned (see oracle/warp_oracle.py)."""
import types

import numpy as np
import pytest

from oracle import warp_oracle

pytestmark = pytest.mark.gpu


def _box(start, size):
  return types.SimpleNamespace(start=np.array(start), size=np.array(size))


def test_kat_segmentation_translate(gpu):
  from sofima_amd import warp
  image = np.zeros((1, 2, 100, 100), dtype=np.uint64)
  image[0, 0, 40, 30] = 42
  image[0, 1, 50, 40] = 2**40
  image_box = _box((0, 0, 0), (100, 100, 2))
  coord_map = np.zeros((2, 2, 15, 15))       # larger than the requested output
  coord_map[0, 0, :, :] = 10
  coord_map[1, 1, :, :] = 17
  map_box = _box((0, 0, 0), (15, 15, 2))
  out_box = _box((10, 20, 0), (90, 80, 2))   # at an offset relative to the input
  warped = warp.warp_subvolume(image, image_box, coord_map, map_box, 10, out_box)
  expected = np.zeros((1, 2, 80, 90))
  expected[0, 0, 20, 10] = 42
  expected[0, 1, 13, 30] = 2**40
  assert warped.dtype == np.uint64
  np.testing.assert_array_equal(warped, expected)


def test_kat_rotate(gpu):
  from sofima_amd import warp
  hy, hx = np.mgrid[-50:50, -50:50]
  image = np.zeros((1, 1, 100, 100), dtype=np.uint8)
  image[0, 0, ...][np.abs(hy) + np.abs(hx) < 25] = 255      # rhombus
  box = _box((0, 0, 0), (100, 100, 1))
  angle = np.pi / 4
  coord_map = np.zeros((2, 1, 10, 10))
  coord_map[0, 0] = (np.cos(angle) * hx[::10, ::10] - np.sin(angle) * hy[::10, ::10]
                     ) - hx[::10, ::10]
  coord_map[1, 0] = (np.sin(angle) * hx[::10, ::10] + np.cos(angle) * hy[::10, ::10]
                     ) - hy[::10, ::10]
  warped = warp.warp_subvolume(image, box, coord_map, _box((0, 0, 0), (10, 10, 1)), 10,
                               box)
  mask = np.zeros((1, 1, 100, 100), dtype=bool)
  mask[0, 0, 33:68, 33:68] = True
  assert warped.dtype == np.uint8
  assert np.all(warped[mask] > 128)
  assert np.all(warped[~mask] < 64)


@pytest.mark.parametrize('map_dtype', [np.float64, np.float32])
@pytest.mark.parametrize('dtype', [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize('interp', ['nearest', 'linear', 'cubic', None])
def test_warp_vs_oracle(gpu, dtype, interp, map_dtype):
  """Smooth map with extrapolated borders, a NaN section, an output box that
  hangs over the image: identical to the oracle (integer types exactly), for
  float32 and float64 maps (the dense coordinates are interpolated in the
  map's dtype, warp.py:125-150)."""
  from sofima_amd import warp
  from tests.util import em_texture
  rng = np.random.default_rng(5)
  img = np.stack([em_texture(rng, (3, 120, 150))]).astype(dtype)
  if dtype != np.uint8:
    img = img * dtype(37 if dtype == np.uint16 else 0.37)
  yy, xx = np.mgrid[:9, :11]
  cm = np.zeros((2, 3, 9, 11))
  cm[0] = 4 * np.sin(yy / 3.0) + 0.3 * xx + rng.standard_normal((3, 9, 11))
  cm[1] = 3 * np.cos(xx / 4.0) - 0.2 * yy
  cm[:, 1] = np.nan                               # skipped section
  cm = cm.astype(map_dtype)
  boxes = dict(image_box=((5, 8, 0), (150, 120, 3)), map_box=((0, 0, 0), (11, 9, 3)),
               out_box=((-3, 2, 0), (140, 110, 3)))
  want = warp_oracle.warp_subvolume(img, boxes['image_box'], cm, boxes['map_box'], 16.0,
                                    boxes['out_box'], interp)
  got = warp.warp_subvolume(img, _box(*boxes['image_box']), cm, _box(*boxes['map_box']),
                            16.0, _box(*boxes['out_box']), interp)
  assert got.shape == want.shape == (1, 3, 110, 140) and got.dtype == dtype
  assert not got[:, 1].any()
  if dtype == np.float32:
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-4)
  else:
    np.testing.assert_array_equal(got, want)
  assert got[:, 0].std() > 1


def test_warp_z_extents_must_agree(gpu):
  """The reference indexes the map and the output with every z of the image
  (warp.py:141-176): a shorter map or output box is an error, not a silently
  unwarped section."""
  from sofima_amd import warp
  img = np.zeros((1, 3, 20, 20), np.uint8)
  cm = np.zeros((2, 2, 3, 3), np.float32)
  with pytest.raises(ValueError):
    warp.warp_subvolume(img, _box((0, 0, 0), (20, 20, 3)), cm, _box((0, 0, 0), (3, 3, 2)), 10.0,
                        _box((0, 0, 0), (20, 20, 3)))
  cm = np.zeros((2, 3, 3, 3), np.float32)
  with pytest.raises(ValueError):
    warp.warp_subvolume(img, _box((0, 0, 0), (20, 20, 3)), cm, _box((0, 0, 0), (3, 3, 3)), 10.0,
                        _box((0, 0, 0), (20, 20, 2)))
  # an output box with more sections than the image: the extra ones stay zero
  out = warp.warp_subvolume(img + 7, _box((0, 0, 0), (20, 20, 3)), cm, _box((0, 0, 0), (3, 3, 3)),
                            10.0, _box((0, 0, 0), (20, 20, 4)))
  assert out.shape == (1, 4, 20, 20) and not out[:, 3].any() and out[:, :3].any()


def test_ndimage_warp_vs_reference_output(gpu, golden):
  """warp.ndimage_warp (HIP, one kernel, double arithmetic in SciPy's operation
  order) == the reference function's own output on every fixture: 2-d / 3-d,
  uint8 / uint16 / float32, linear / nearest, boxes, out_scale -- bit for bit,
  floats included."""
  from sofima_amd import warp
  from tests.util import ndimage_warp_case
  g = golden('ndimage_warp')
  for name in g['names']:
    img, cmap, stride, work, ov, order, boxes, scale, want = ndimage_warp_case(g, str(name))
    kw = {}
    if boxes is not None:
      kw = {k + '_box': types.SimpleNamespace(start=boxes[k][0], size=boxes[k][1])
            for k in ('image', 'map', 'out')}
    if scale is not None:
      kw['out_scale'] = scale
    got = warp.ndimage_warp(img, cmap, stride, work, ov, order=order, parallelism=2, **kw)
    assert got.dtype == want.dtype and got.shape == want.shape, name
    np.testing.assert_array_equal(got, want, err_msg=str(name))


@pytest.mark.parametrize('dim', [2, 3])
def test_ndimage_warp_vs_oracle_random(gpu, dim):
  """Larger random cases against the SciPy restatement (pinned by the fixtures):
  maps that leave the image and the node grid, fractional strides."""
  from sofima_amd import warp
  rng = np.random.default_rng(50 + dim)
  if dim == 2:
    img = rng.integers(0, 256, (333, 517)).astype(np.uint8)
    cmap = (rng.standard_normal((2, 12, 18)) * 25).astype(np.float32)
    stride, work, ov = (31.5, 30.25), (128, 128), (16, 16)
  else:
    img = (rng.standard_normal((21, 90, 77)) * 50).astype(np.float32)
    cmap = (rng.standard_normal((3, 6, 11, 9)) * 6).astype(np.float32)
    stride, work, ov = (4.5, 9.0, 9.5), (64, 64, 16), (8, 8, 2)
  for order in (0, 1):
    got = warp.ndimage_warp(img, cmap, stride, work, ov, order=order)
    want = warp_oracle.ndimage_warp(img, cmap, stride, order=order)
    np.testing.assert_array_equal(got, want)
  with pytest.raises(NotImplementedError):
    warp.ndimage_warp(img, cmap, stride, work, ov, order=3)
  with pytest.raises(ValueError):
    warp.ndimage_warp(img[0], cmap, stride, work, ov)
