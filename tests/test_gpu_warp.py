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
