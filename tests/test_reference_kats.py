"""The reference's own known-answer tests, re-typed against the CPU oracle.

Sources: /root/reference/tests/flow_field_test.py:24-125 and
tests/mesh_test.py:25-144.  The same cases run against the HIP path in
tests/test_gpu_flow.py / tests/test_gpu_mesh.py.
"""
import types

import numpy as np

from oracle import flow_oracle as fo
from oracle import mesh_oracle as mo


def test_delta_images_and_mask():
  pre = np.zeros((120, 120), np.uint8)
  post = np.zeros((120, 120), np.uint8)
  pre[60, 60] = 255
  post[70, 53] = 255
  f = fo.flow_field(pre, post, 80, 40, batch_size=4)
  assert f.shape == (4, 2, 2)
  np.testing.assert_array_equal(f[0], 7)
  np.testing.assert_array_equal(f[1], -10)
  np.testing.assert_array_equal(f[3], 0)
  post[54, 68] = 255
  mask = np.zeros((128, 128), bool)
  mask[:55, :70] = 1
  f = fo.flow_field(pre, post, 80, 40, post_mask=mask, batch_size=4)
  np.testing.assert_array_equal(f[0], 7)
  np.testing.assert_array_equal(f[1], -10)
  np.testing.assert_array_equal(f[3], 0)


def test_3d():
  pre = np.zeros((50, 100, 100), np.uint8)
  post = np.zeros((50, 100, 100), np.uint8)
  pre[25, 50, 50] = 255
  post[22, 45, 54] = 255
  f = fo.flow_field(pre, post, (40, 80, 80), 10, batch_size=1)
  assert f.shape == (5, 2, 3, 3)
  np.testing.assert_array_equal(f[0], -4)
  np.testing.assert_array_equal(f[1], 5)
  np.testing.assert_array_equal(f[2], 3)


def test_peak():
  hy, hx = np.mgrid[:50, :50]
  cy, cx = 20, 28
  r = np.sqrt(2 * (cx - hx)**2 + (cy - hy)**2)
  xcorr = (10 * np.exp(-r / 4)).astype(np.float32)
  p = fo.batched_peaks(xcorr[None], (25, 25), 2, 0.5, (2, 3))
  assert p.shape == (1, 4)
  assert p[0, 0] == 3 and p[0, 1] == -5 and p[0, 3] == 0
  assert p[0, 2] == xcorr[cy, cx] / xcorr[cy - 2:cy + 3, cx - 3:cx + 4].min()


def test_post_targeting():
  pre = np.zeros((120, 120), np.uint8)
  post = np.zeros((120, 120), np.uint8)
  pre[50, 55] = 255
  post[100, 100] = 255
  f = fo.flow_field(pre, post, 80, 40, batch_size=4)
  assert np.isnan(f[:, 0, 0]).all()
  tg = np.full((2, 2, 2), 40.0, np.float32)
  f = fo.flow_field(pre, post, 80, 40, batch_size=4, post_targeting_field=tg,
                    post_targeting_step=40)
  np.testing.assert_array_equal(f[0], -45)
  np.testing.assert_array_equal(f[1], -50)


def _cfg(**kw):
  base = dict(f_alpha=0.99, f_inc=1.1, f_dec=0.5, alpha=0.1, n_min=5,
              dt_max=10.0, start_cap=1e6, final_cap=1e6, cap_scale=1.1,
              cap_upscale_every=100, prefer_orig_order=False,
              remove_drift=False, fire=True)
  base.update(kw)
  return types.SimpleNamespace(**base)


def test_relaxation_fire_and_damped():
  for fire, gamma in ((True, 0.0), (False, 0.9 * np.sqrt(4 * 0.1))):
    x = np.zeros((2, 1, 50, 50))
    x[0, 0, 20:30, 10] = 3
    x[0, 0, 20:30, 40] = -4
    x[1, 0, 30, 10:20] = 2
    cfg = _cfg(dt=0.01, gamma=gamma, k0=0.1, k=0.1, stride=(10, 10),
               num_iters=100, max_iters=10000, stop_v_max=0.001, fire=fire)
    new_x, _, _ = mo.relax_mesh(x, np.zeros_like(x), cfg)
    np.testing.assert_array_almost_equal(new_x, np.zeros_like(x), decimal=3)


def test_equilibrium():
  x = np.zeros((2, 1, 10, 10))
  np.testing.assert_array_equal(x, mo.inplane_force(x, 1.0, (40.0, 40.0)))
  x = np.zeros((3, 10, 10, 10))
  np.testing.assert_array_equal(x, mo.elastic_mesh_3d(x, 1.0, 40.0))
  x = np.zeros((3, 5, 10, 10, 10))
  np.testing.assert_array_equal(x, mo.elastic_mesh_3d(x, 1.0, 40.0))


def test_force_and_consistency():
  x = np.zeros((2, 1, 10, 10))
  dx, dy = 4, -3
  x[0, 0, 5, 5] = dx
  x[1, 0, 5, 5] = dy
  k, l0 = 0.1, 10.0
  f = mo.inplane_force(x, k, (l0, 10))
  l = np.sqrt((l0 + dx)**2 + dy**2)
  np.testing.assert_allclose(
      [k * (l - l0) * (l0 + dx) / l, k * (l - l0) * dy / l], f[:, 0, 5, 4],
      rtol=1e-6)
  l = np.sqrt((l0 - dx)**2 + (l0 - dy)**2)
  l2, k2 = l0 * np.sqrt(2.0), k / np.sqrt(2.0)
  np.testing.assert_allclose(
      [-k2 * (l - l2) * (l0 - dx) / l, -k2 * (l - l2) * (l0 - dy) / l],
      f[:, 0, 6, 6], rtol=1e-5)
  planar = ((1, 0, 0), (0, 1, 0), (1, 1, 0), (-1, 1, 0))
  rng = np.random.default_rng(42)
  x = rng.random((3, 1, 50, 50))
  x[2] = 0
  for poo in (False, True):
    np.testing.assert_allclose(
        mo.inplane_force(x[:2], 0.01, (40.0, 40.0), poo)[:2],
        mo.elastic_mesh_3d(x, 0.01, (40.0, 40.0, 14.0), poo, links=planar)[:2],
        atol=1e-5)


def _clean_flow_kat():
  """Inputs / expected output of tests/flow_utils_test.py:38-64."""
  flow = np.zeros((4, 1, 50, 40))
  flow[2, ...] = 2.0
  flow[2, 0, 10, 20] = 1.2
  flow[3, 0, 10, 22] = 1.2
  flow[3, 0, 10, 24] = 1.6
  flow[0, 0, 5, 4] = 12
  flow[1, 0, 5, 6] = -14
  flow[:, 0, 5, 10] = 2
  flow[:, 0, 15, 10] = 7
  expected = np.zeros((2, 1, 50, 40))
  expected[:, 0, 5, 10] = 2
  expected[:, 0, 15, 10] = np.nan  # median filter
  expected[:, 0, 10, 20] = np.nan  # peak sharpness
  expected[:, 0, 10, 22] = np.nan  # peak ratio
  expected[:, 0, 5, 4] = np.nan  # magnitude
  expected[:, 0, 5, 6] = np.nan  # magnitude
  return flow, dict(min_peak_ratio=1.4, min_peak_sharpness=1.6, max_magnitude=10,
                    max_deviation=5), expected


def test_clean_flow():
  """tests/flow_utils_test.py:38-64 against the oracle."""
  from oracle import flow_utils_oracle
  flow, kw, expected = _clean_flow_kat()
  np.testing.assert_array_equal(flow_utils_oracle.clean_flow(flow, **kw), expected)


def test_mask_irregular():
  """tests/map_utils_test.py:323-333 against the oracle."""
  from oracle import maps_oracle
  coord_map = np.zeros([2, 50, 50])
  coord_map[0, 40, 10] = 10
  masked, bad = maps_oracle.mask_irregular(coord_map, (40, 40), 0.25, 1.1)
  expected = np.zeros([2, 50, 50])
  expected[:, 39:42, 8:11] = np.nan
  np.testing.assert_array_equal(expected, masked)
  np.testing.assert_array_equal(np.isnan(expected[0, ...]), bad)


# -- tests/warp_test.py:27-82 against the warp oracle ---------------------------
def test_warp_subvolume_segmentation_translate_oracle():
  from oracle import warp_oracle
  image = np.zeros((1, 2, 100, 100), dtype=np.uint64)
  image[0, 0, 40, 30] = 42
  image[0, 1, 50, 40] = 2**40
  coord_map = np.zeros((2, 2, 15, 15))
  coord_map[0, 0, :, :] = 10
  coord_map[1, 1, :, :] = 17
  warped = warp_oracle.warp_subvolume(
      image, ((0, 0, 0), (100, 100, 2)), coord_map, ((0, 0, 0), (15, 15, 2)), 10,
      ((10, 20, 0), (90, 80, 2)))
  expected = np.zeros((1, 2, 80, 90))
  expected[0, 0, 20, 10] = 42
  expected[0, 1, 13, 30] = 2**40
  np.testing.assert_array_equal(warped, expected)


def test_warp_subvolume_rotate_oracle():
  from oracle import warp_oracle
  hy, hx = np.mgrid[-50:50, -50:50]
  image = np.zeros((1, 1, 100, 100), dtype=np.uint8)
  image[0, 0, ...][np.abs(hy) + np.abs(hx) < 25] = 255
  angle = np.pi / 4
  coord_map = np.zeros((2, 1, 10, 10))
  coord_map[0, 0] = (np.cos(angle) * hx[::10, ::10] - np.sin(angle) * hy[::10, ::10]
                     ) - hx[::10, ::10]
  coord_map[1, 0] = (np.sin(angle) * hx[::10, ::10] + np.cos(angle) * hy[::10, ::10]
                     ) - hy[::10, ::10]
  box = ((0, 0, 0), (100, 100, 1))
  warped = warp_oracle.warp_subvolume(image, box, coord_map, ((0, 0, 0), (10, 10, 1)),
                                      10, box)
  mask = np.zeros((1, 1, 100, 100), dtype=bool)
  mask[0, 0, 33:68, 33:68] = True
  assert np.all(warped[mask] > 128)
  assert np.all(warped[~mask] < 64)
