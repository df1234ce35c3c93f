"""Host-side logic of the drop-in modules (no GPU): patch planning, config."""
import json

import numpy as np
import pytest

from oracle import flow_oracle as fo
from sofima_amd import flow_field, mesh


def _plan(calc, pre_shape, post_shape, patch, step, **kw):
  nd = len(pre_shape)
  post_patch = kw.pop('post_patch_size', None) or patch
  return calc.plan(pre_shape, post_shape, patch, step,
                   post_patch_size=post_patch,
                   counts_fn=flow_field._host_masked_counts, **kw)


def test_plan_matches_oracle_selection():
  rng = np.random.default_rng(0)
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  pre_mask = rng.random((200, 180)) < 0.4
  pre_mask[:80, :90] = True
  post_mask = np.zeros((210, 190), bool)
  post_mask[120:, 100:] = True
  sel = rng.random((8, 7)) < 0.7
  for kw in (dict(), dict(pre_mask=pre_mask), dict(post_mask=post_mask),
             dict(pre_mask=pre_mask, post_mask=post_mask, max_masked=0.5),
             dict(selection_mask=sel)):
    p = _plan(calc, (200, 180), (200, 180), (48, 40), (24, 20), batch_size=7,
              **kw)
    out_shape, grid = fo.plan_patches((200, 180), (200, 180), (48, 40),
                                      (24, 20), (48, 40), kw.get('pre_mask'),
                                      kw.get('post_mask'),
                                      kw.get('selection_mask'),
                                      kw.get('max_masked', 0.75))
    np.testing.assert_array_equal(p['out_shape'], out_shape)
    np.testing.assert_array_equal(p['positions'], grid)
    n = len(grid)
    assert p['pre_starts'].shape[0] == p['n_batches'] * 7 >= n
    # padding repeats the last position
    np.testing.assert_array_equal(p['post_starts'][n:],
                                  np.repeat(p['post_starts'][n - 1:n],
                                            p['n_batches'] * 7 - n, 0))


def test_plan_3d_and_post_patch():
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  p = _plan(calc, (50, 100, 100), (50, 100, 100), (40, 80, 80), (10, 10, 10),
            batch_size=1)
  np.testing.assert_array_equal(p['out_shape'], [2, 3, 3])
  assert p['positions'].shape == (18, 3)
  p = _plan(calc, (192, 160), (192, 160), (48, 48), (24, 24), batch_size=8,
            post_patch_size=(32, 32))
  np.testing.assert_array_equal(p['out_shape'], [7, 6])
  # pre patches are shifted by (48 - 32) // 2 and clipped at 0
  np.testing.assert_array_equal(p['pre_starts'][0], [0, 0])
  np.testing.assert_array_equal(p['pre_starts'][6], [24 - 8, 0])
  np.testing.assert_array_equal(p['post_starts'][6], [24, 0])


def test_targeting_offsets_match_oracle():
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  tg = np.zeros((2, 4, 4), np.float32)
  tg[0], tg[1] = 60.0, -40.0
  tg[0, 0, 0] = np.nan
  p = _plan(calc, (192, 160), (192, 160), (48, 48), (24, 24), batch_size=8,
            pre_targeting_field=tg, pre_targeting_step=(48, 48))
  starts = np.array(p['positions']) * 24
  want = fo._target_offsets(tg, (48, 48), starts, (48, 48), (192, 160))
  np.testing.assert_array_equal(p['tg_offsets'][:len(starts)], want)
  moved = p['pre_starts'][:len(starts)]
  assert (moved >= 0).all()
  assert (moved + 48 <= np.array([192, 160])).all()


def test_integration_config_roundtrip_and_hash():
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1,
                               stride=[40, 40], num_iters=1000,
                               max_iters=100000, stop_v_max=0.005,
                               dt_max=1000, start_cap=0.01, final_cap=10,
                               prefer_orig_order=True)
  assert cfg.stride == (40, 40)          # list -> tuple, like the reference
  hash(cfg)                              # usable as a static/hashable argument
  d = cfg.to_dict()
  assert d['fire'] is True and d['f_alpha'] == 0.99 and d['n_min'] == 5
  again = mesh.IntegrationConfig.from_json(cfg.to_json())
  assert again == cfg
  assert json.loads(cfg.to_json())['cap_upscale_every'] == 100
  with pytest.raises(Exception):
    cfg.dt = 1.0                         # frozen


def test_public_names_of_the_reference_exist():
  for name in ('masked_xcorr', '_batched_peaks', 'batched_xcorr_peaks',
               'JAXMaskedXCorrWithStatsCalculator', '_integral_image'):
    assert hasattr(flow_field, name)
  assert flow_field.JAXMaskedXCorrWithStatsCalculator.non_spatial_flow_channels == 2
  for name in ('inplane_force', 'elastic_mesh_3d', 'MESH_LINK_DIRECTIONS',
               'IntegrationConfig', 'velocity_verlet', 'relax_mesh'):
    assert hasattr(mesh, name)
  assert len(mesh.MESH_LINK_DIRECTIONS) == 13


def test_validation_errors_precede_device_work():
  x = np.zeros((2, 1, 8, 8), np.float32)
  base = dict(dt=0.01, gamma=0.0, k0=0.1, k=0.1, stride=(10, 10), num_iters=5,
              max_iters=10, stop_v_max=0.001)
  with pytest.raises(NotImplementedError):
    mesh.relax_mesh(x, x, mesh.IntegrationConfig(fire=False, start_cap=1.0,
                                                 final_cap=2.0, **base))
  with pytest.raises(ValueError):
    mesh.relax_mesh(x, x, mesh.IntegrationConfig(start_cap=1.0, final_cap=2.0,
                                                 cap_scale=1.0, **base))
  with pytest.raises(ValueError):
    mesh.relax_mesh(x, x, mesh.IntegrationConfig(**base), prev_fn=lambda a: a)
  with pytest.raises(ValueError):
    mesh.inplane_force(x, 0.1, (10, 10, 10))


def test_integral_image_query():
  rng = np.random.default_rng(1)
  m = rng.random((37, 41)) < 0.3
  svt = flow_field._integral_image(m)
  got = flow_field._query_integral_image(svt, (8, 6), (4, 5))
  want = np.array([[m[y:y + 8, x:x + 6].sum() for x in range(0, 41 - 6 + 1, 5)]
                   for y in range(0, 37 - 8 + 1, 4)])
  np.testing.assert_array_equal(got, want)
  m3 = rng.random((9, 11, 13)) < 0.3
  got = flow_field._query_integral_image(flow_field._integral_image(m3),
                                         (4, 5, 6), (2, 3, 4))
  want = np.array([[[m3[z:z + 4, y:y + 5, x:x + 6].sum()
                     for x in range(0, 13 - 6 + 1, 4)]
                    for y in range(0, 11 - 5 + 1, 3)]
                   for z in range(0, 9 - 4 + 1, 2)])
  np.testing.assert_array_equal(got, want)


def test_overlap_strips_match_oracle():
  """Strip selection of stitch_elastic.compute_flow_map (host logic)."""
  from oracle import stitch_oracle as so
  from sofima_amd import stitch_elastic
  rng = np.random.default_rng(0)
  for _ in range(200):
    h, w = rng.integers(100, 200, 2)
    pre = rng.integers(0, 255, (h, w)).astype(np.uint8)
    post = rng.integers(0, 255, (h, w)).astype(np.uint8)
    axis = int(rng.integers(0, 2))
    stride = (int(rng.integers(4, 20)), int(rng.integers(4, 20)))
    off = np.zeros(2)
    off[axis] = -rng.integers(20, 60)
    off[1 - axis] = rng.integers(-30, 30) + rng.random()
    a, b, o = stitch_elastic._overlap_strips(pre, post, off, axis, stride)
    wa, wb, wo = so.flow_map_strips(pre, post, off, axis, stride)
    np.testing.assert_array_equal(a, wa)
    np.testing.assert_array_equal(b, wb)
    assert tuple(o) == tuple(wo)


def test_interpolate_missing_offsets():
  from sofima_amd import stitch_rigid
  conn = np.zeros((2, 1, 3, 5))
  conn[0, 0] = np.arange(15).reshape(3, 5)
  conn[1, 0] = -np.arange(15).reshape(3, 5)
  conn[:, 0, 1, 2] = np.inf          # neighbours at distance 1 on both sides
  conn[:, 0, 0, 0] = np.inf          # only a right neighbour
  conn[:, 0, 2, 3:] = np.inf         # (2, 3): left at r=1; (2, 4): left at r=2
  out = stitch_rigid.interpolate_missing_offsets(conn.copy(), -1)
  np.testing.assert_array_equal(out[:, 0, 1, 2], [(6 + 8) / 2, -(6 + 8) / 2])
  np.testing.assert_array_equal(out[:, 0, 0, 0], [1, -1])
  np.testing.assert_array_equal(out[:, 0, 2, 3], [12, -12])
  # the search runs on the array being modified: (2, 3) is finite by now
  np.testing.assert_array_equal(out[:, 0, 2, 4], [12, -12])
  out = stitch_rigid.interpolate_missing_offsets(conn.copy(), -2, max_r=2)
  np.testing.assert_array_equal(out[:, 0, 1, 2], [(2 + 12) / 2, -(2 + 12) / 2])
  with pytest.raises(ValueError):
    stitch_rigid.interpolate_missing_offsets(np.zeros((2, 3, 5)), -1)


def test_force_spec_resolution():
  import functools
  spec = mesh._resolve_force(mesh.inplane_force)
  assert (spec.kind, spec.ncomp) == (0, 2)
  spec = mesh._resolve_force(functools.partial(mesh.elastic_mesh_3d,
                                               links=((1, 0, 0),)))
  assert (spec.kind, spec.ncomp, spec.links) == (0, 3, ((1, 0, 0),))
  spec = mesh._resolve_force(lambda x, *a, **k: x)
  assert spec.kind == 2 and spec.ncomp is None
  with pytest.raises(TypeError):
    mesh._resolve_force(3)


def test_flow_map3d_overlap_alignment_matches_reference_offsets(golden):
  """The host arithmetic of compute_flow_map3d (aligned overlap boxes,
  stitch_elastic.py:125-180): recorded offsets == the reference's, the flow grid
  sizes follow from the overlap sizes."""
  from sofima_amd import stitch_elastic
  g = golden('flow_map3d')
  shape = tuple(int(v) for v in g['tile_shape'])
  stride = tuple(int(v) for v in g['stride'])
  patch = tuple(int(v) for v in g['patch'])
  for name, om, axis in (('fx', g['ox'], 0), ('fy', g['oy'], 1)):
    for i, k in enumerate(g[name + '_keys']):
      x, y = int(k[0]), int(k[1])
      cur, nb, size, rec = stitch_elastic._aligned_overlap3d(shape, om[:, 0, y, x], axis,
                                                             stride)
      assert rec == tuple(g[name + '_offsets'][i])
      s = stride[2 - axis]
      assert np.all(cur % s == 0) and np.all(nb % s == 0) and np.all(size > 0)
      grid = [(n - (p - st)) // st + p // 2 // st * 2 - 1
              for n, p, st in zip(size[::-1], patch, stride)]
      assert tuple(g[f'{name}_{i}'].shape[1:]) == tuple(grid)
