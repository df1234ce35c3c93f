"""stitch_rigid drop-ins on the HIP path vs the reference's outputs (-m gpu).

stitch_cfg1.npz holds what the unmodified reference produced for the 2 x 2
montage of BASELINE configs[0] (tests/golden/make_golden.py: gen_stitch).
"""
import json

import numpy as np
import pytest

from oracle import mesh_oracle
from oracle import stitch_oracle
from tests.util import cfg_from

pytestmark = pytest.mark.gpu


def _tiles(g):
  return {tuple(int(v) for v in k): t for k, t in zip(g['tile_keys'], g['tiles'])}


@pytest.mark.parametrize('dtype', [np.uint8, np.float32])
@pytest.mark.parametrize('size,limit', [(10, 40), (7, 25.5), (3, 8), (1, 1)])
def test_range_mask_matches_scipy(gpu, dtype, size, limit):
  from sofima_amd import stitch_rigid
  rng = np.random.default_rng(size)
  from tests.util import em_texture
  img = em_texture(rng, (97, 131)).astype(dtype)
  if dtype == np.float32:
    img += rng.random(img.shape, dtype=np.float32)
  extra = rng.random(img.shape) < 0.1
  want = stitch_oracle.range_mask(img, limit, size)
  got = stitch_rigid.range_mask(img, limit, size).cpu().numpy().astype(bool)
  np.testing.assert_array_equal(got, want)
  got = stitch_rigid.range_mask(img, limit, size, extra).cpu().numpy().astype(bool)
  np.testing.assert_array_equal(got, want | extra)


def test_estimate_offset_vs_reference_output(gpu, golden):
  from sofima_amd import stitch_rigid
  g = golden('stitch_cfg1')
  tiles = _tiles(g)
  a, b = tiles[(0, 0)][:, -96:], tiles[(1, 0)][:, :96]
  off, pr = stitch_rigid._estimate_offset(a, b, 10)
  np.testing.assert_array_equal(off + [pr], g['eo0'])
  off, pr = stitch_rigid._estimate_offset(a, b, 40, filter_size=7,
                                          masks=(g['eo1_ma'], g['eo1_mb']))
  np.testing.assert_array_equal(off + [pr], g['eo1'])
  off, pr = stitch_rigid._estimate_offset(tiles[(0, 0)][-128:, :],
                                          tiles[(0, 1)][:128, :], 0)
  np.testing.assert_array_equal(off + [pr], g['eo2'])
  off, pr = stitch_rigid._estimate_offset(a, b, 250, filter_size=7,
                                          masks=(g['eo1_ma'], g['eo1_mb']))
  assert np.isnan(off).all() and np.isnan(pr)


def test_compute_coarse_offsets_vs_reference_output(gpu, golden):
  from sofima_amd import stitch_rigid
  g = golden('stitch_cfg1')
  cx, cy = stitch_rigid.compute_coarse_offsets(
      (2, 2), _tiles(g), overlaps_xy=((96, 128), (96, 128)), min_overlap=32)
  np.testing.assert_array_equal(cx, g['cx'])
  np.testing.assert_array_equal(cy, g['cy'])


def test_estimate_offset_large_strip_vs_oracle(gpu):
  """A 4096 x 300 overlap (cfg-3 tile size) with a second, weaker match so that
  the peak ratio is non-trivial."""
  from sofima_amd import stitch_rigid
  from tests.util import em_texture
  rng = np.random.default_rng(77)
  base = em_texture(rng, (4096 + 40, 340))
  a = np.ascontiguousarray(base[20:20 + 4096, 20:320])
  b = np.ascontiguousarray(base[9:9 + 4096, 26:326])
  b[2000:2600] = base[20 + 2000 + 30:20 + 2600 + 30, 20 + 11:320 + 11]  # distractor
  off, pr = stitch_rigid._estimate_offset(a, b, 30)
  woff, wpr = stitch_oracle.estimate_offset(a, b, 30)
  assert off == woff == [6.0, -11.0]
  np.testing.assert_allclose(pr, wpr, rtol=1e-4)


def test_tile_mesh_forces_vs_reference_output(gpu, golden):
  from sofima_amd import stitch_rigid
  g = golden('stitch_cfg1')
  f = np.array(stitch_rigid.elastic_tile_mesh(g['tm_x'], g['tm_cx'], g['tm_cy']))
  np.testing.assert_allclose(f, g['tm_f'], rtol=1e-6, atol=1e-5)
  f = np.array(stitch_rigid.elastic_tile_mesh_3d(g['tm3_x'], g['tm3_cx'], g['tm3_cy']))
  np.testing.assert_allclose(f, g['tm3_f'], rtol=1e-6, atol=1e-5)
  # at the oracle: identical operation order -> bit-identical
  np.testing.assert_array_equal(
      np.array(stitch_rigid.elastic_tile_mesh(g['tm_x'], g['tm_cx'], g['tm_cy'])),
      stitch_oracle.elastic_tile_mesh(g['tm_x'], g['tm_cx'], g['tm_cy']))
  # non-finite pair terms: nan -> 0, inf -> float max (jnp.nan_to_num defaults)
  cx = g['tm_cx'].copy()
  cx[0, 0, 0, 0] = np.inf
  np.testing.assert_array_equal(
      np.array(stitch_rigid.elastic_tile_mesh(g['tm_x'], cx, g['tm_cy'])),
      stitch_oracle.elastic_tile_mesh(g['tm_x'], cx, g['tm_cy']))


def test_optimize_coarse_mesh_vs_reference_output(gpu, golden):
  """relax_mesh with the tile-mesh force (the boundary hole of round 1):
  stitch_rigid.optimize_coarse_mesh reproduces the reference's tile layout."""
  from sofima_amd import mesh, stitch_rigid
  g = golden('stitch_cfg1')
  cfg = cfg_from(json.loads(str(g['tm_cfg'])), mesh.IntegrationConfig)
  got = stitch_rigid.optimize_coarse_mesh(g['tm_cx'], g['tm_cy'], cfg)
  assert got.shape == g['tm_relaxed'].shape and got.dtype == np.float32
  np.testing.assert_allclose(got, g['tm_relaxed'], atol=1e-3)
  got = stitch_rigid.optimize_coarse_mesh(g['tm3_cx'], g['tm3_cy'], cfg,
                                          mesh_fn=stitch_rigid.elastic_tile_mesh_3d)
  np.testing.assert_allclose(got, g['tm3_relaxed'], atol=1e-3)
  got = stitch_rigid.optimize_coarse_mesh(g['cx'], g['cy'])   # default config
  np.testing.assert_allclose(got, g['coarse'], atol=1e-3)
  # same number of steps and energy trace as the oracle
  force = mesh.TileMeshForce(g['tm_cx'], g['tm_cy'])
  gx, ge, gt = mesh.relax_mesh(np.zeros_like(g['tm_cx']), None, cfg, mesh_force=force)
  wx, we, wt = mesh_oracle.relax_mesh(
      np.zeros_like(g['tm_cx']), None, cfg,
      mesh_force=lambda x, *a, **k: stitch_oracle.elastic_tile_mesh(
          x, g['tm_cx'], g['tm_cy']))
  assert gt == wt and len(ge) == len(we)
  # converged: the last kinetic energies are round-off of ~0 on both sides
  np.testing.assert_allclose(ge[:-1], we[:-1], rtol=1e-3)
  assert ge[-1] < 1e-5 and we[-1] < 1e-5
  np.testing.assert_allclose(np.array(gx), wx, atol=1e-3)


def test_arbitrary_mesh_force_callables(gpu, golden):
  """Any callable f(x, k, stride, prefer_orig_order) works as `mesh_force`
  (mesh.py:427-428): evaluated per step on the device-resident state."""
  import torch
  from sofima_amd import mesh, stitch_rigid
  g = golden('stitch_cfg1')
  cfg = cfg_from(json.loads(str(g['tm_cfg'])), mesh.IntegrationConfig)
  cfg = mesh.IntegrationConfig(**{**cfg.to_dict(), 'num_iters': 200, 'max_iters': 600})
  cx, cy = g['tm_cx'], g['tm_cy']
  native = mesh.relax_mesh(np.zeros_like(cx), None, cfg,
                           mesh_force=mesh.TileMeshForce(cx, cy))
  calls = []

  def closure(x, *args, **kwargs):          # the reference's _mesh_force shape
    calls.append(args)
    return stitch_rigid.elastic_tile_mesh(x, cx, cy, *args, **kwargs)

  via_cb = mesh.relax_mesh(np.zeros_like(cx), None, cfg, mesh_force=closure)
  assert via_cb[2] == native[2] and len(calls) == (cfg.num_iters + 1) * (native[2] // 200)
  assert calls[0] == (cfg.k, cfg.stride, cfg.prefer_orig_order)
  np.testing.assert_array_equal(np.array(via_cb[0]), np.array(native[0]))
  assert via_cb[1] == native[1]

  # a torch-written force, and a NumPy-returning one, on a spring mesh with prev
  rng = np.random.default_rng(3)
  prev = (rng.standard_normal((2, 2, 20, 24)) * 3).astype(np.float32)
  scfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.05, k=0.1, stride=(10, 10),
                                num_iters=30, max_iters=30, stop_v_max=1e-9, dt_max=100,
                                start_cap=1.0, final_cap=10.0, remove_drift=True)
  ref = mesh.relax_mesh(np.zeros_like(prev), prev, scfg)
  a = mesh.relax_mesh(np.zeros_like(prev), prev, scfg,
                      mesh_force=lambda x, k, s, p: mesh.inplane_force(x, k, s, p).tensor)
  b = mesh.relax_mesh(np.zeros_like(prev), prev, scfg,
                      mesh_force=lambda x, k, s, p: mesh_oracle.inplane_force(
                          np.asarray(x), k, s, p))
  np.testing.assert_allclose(np.array(a[0]), np.array(ref[0]), atol=2e-4)
  np.testing.assert_allclose(np.array(b[0]), np.array(ref[0]), atol=2e-4)
  vv = mesh.velocity_verlet(np.zeros_like(prev), np.zeros_like(prev), prev, scfg, 1.0,
                            mesh_force=lambda x, k, s, p: -0.01 * x.tensor)
  assert isinstance(vv[0].tensor, torch.Tensor) and len(vv) == 7

  class Boom(RuntimeError):
    pass

  def bad(x, *a):
    raise Boom('from the callback')

  with pytest.raises(Boom):
    mesh.relax_mesh(np.zeros_like(prev), prev, scfg, mesh_force=bad)
  with pytest.raises(ValueError):
    mesh.relax_mesh(np.zeros_like(prev), prev, scfg,
                    mesh_force=lambda x, *a: np.zeros((2, 3)))
