"""Pins the CPU oracle against the golden vectors made from the reference.

xcorr_np.npz comes from the reference's genuine NumPy path
(masked_xcorr(use_jax=False)); the other files from the reference source
executed over the NumPy stand-in for jax (tests/golden/make_golden.py).
"""
import numpy as np
import pytest

from oracle import flow_oracle as fo
from oracle import mesh_oracle as mo
from tests.util import cfg_from, check_flow, load_cfgs


def test_xcorr_surface_vs_reference_numpy_path(golden):
  g = golden('xcorr_np')
  for fn in (fo.xcorr_surface, fo.xcorr_surface_direct):
    s = np.abs(g['unmasked']).max()
    np.testing.assert_allclose(fn(g['a'], g['b']), g['unmasked'], atol=2e-6 * s)
    np.testing.assert_allclose(fn(g['a'], g['b'], g['am'], g['bm']),
                               g['masked'], atol=1e-6)
    s = np.abs(g['unmasked3']).max()
    np.testing.assert_allclose(fn(g['a3'], g['b3'], dim=3), g['unmasked3'],
                               atol=2e-6 * s)
    np.testing.assert_allclose(fn(g['a3'], g['b3'], g['am3'], g['bm3'], dim=3),
                               g['masked3'], atol=1e-6)
  # One-sided mask; the reference's NumPy path runs in float64 there.
  np.testing.assert_allclose(
      fo.xcorr_surface(g['a'], g['b'], g['am'], None, dtype=np.float64),
      g['masked_prev_only'], atol=1e-6)
  np.testing.assert_allclose(
      fo.xcorr_surface_direct(g['a'], g['b'], g['am'], None),
      g['masked_prev_only'], atol=1e-6)


def test_batched_peaks_vs_golden(golden):
  g = golden('peaks')
  np.testing.assert_array_equal(
      fo.batched_peaks(g['imgs'], (15, 15), 2, 0.5, 5), g['out_all'])
  np.testing.assert_array_equal(
      fo.batched_peaks(g['imgs'][:1], (15, 15), 2, 0.5, 5), g['out_0'])
  np.testing.assert_array_equal(
      fo.batched_peaks(g['imgs'][5:], (15, 15), 2, 0.5, (2, 3)), g['out_r'])
  np.testing.assert_array_equal(
      fo.batched_peaks(g['vol'], (4, 5, 6), 1, 0.5, (1, 2, 2)), g['out_3d'])
  # The batch-coupled quirks are in the vectors: image 0 alone has ratio 1.25,
  # batched with image 1 its second peak is suppressed.
  assert g['out_0'][0, 3] == np.float32(1.25)
  assert g['out_all'][0, 3] == 0
  assert g['out_all'][2, 3] == 1.0          # single peak at flat index 0
  assert np.isnan(g['out_all'][3]).all()    # all-zero surface


def test_flow_field_vs_golden(golden):
  g = golden('flow2d')
  pre, post = g['pre'], g['post']
  check_flow(fo.flow_field(pre, post, 48, 24, batch_size=8), g['plain'])
  check_flow(fo.flow_field(pre, post, 48, 24, pre_mask=g['pre_mask'],
                           post_mask=g['post_mask'], batch_size=8), g['masked'])
  check_flow(fo.flow_field(pre, post, 48, 24, pre_mask=g['pre_mask'],
                           post_mask=g['post_mask'],
                           mask_only_for_patch_selection=True, max_masked=0.5,
                           batch_size=16), g['masksel'])
  check_flow(fo.flow_field(pre, post, 48, 24, batch_size=8, post_patch_size=32),
             g['postpatch'])
  check_flow(fo.flow_field(pre, post, (48, 32), (24, 16),
                           selection_mask=np.pad(g['sel'], ((0, 0), (0, 4))),
                           batch_size=5), g['selected'])
  check_flow(fo.flow_field(pre, post, 48, 24, batch_size=8,
                           pre_targeting_field=g['tg_pre'],
                           pre_targeting_step=48,
                           post_targeting_field=g['tg_post'],
                           post_targeting_step=64), g['targeted'])
  check_flow(fo.flow_field(g['pre_f'], g['post_f'], 40, 20, batch_size=64,
                           mean=120.0, min_distance=3, peak_radius=(3, 4)),
             g['float_mean'])
  # the vectors exercise what they claim to
  assert np.isnan(g['masksel'][0]).sum() > np.isnan(g['plain'][0]).sum()
  assert (g['plain'][0] == -5).all() and (g['plain'][1] == 3).all()


def test_flow_field_3d_vs_golden(golden):
  g = golden('flow3d')
  check_flow(fo.flow_field(g['pre'], g['post'], (16, 24, 24), 8, batch_size=4),
             g['plain'])


def test_forces_vs_golden(golden):
  g = golden('mesh_force')
  tol = dict(rtol=1e-5, atol=2e-6)
  for poo in (0, 1):
    np.testing.assert_allclose(
        mo.inplane_force(g['x2'], 0.1, (40.0, 30.0), bool(poo)),
        g[f'f2_{poo}'], **tol)
    np.testing.assert_allclose(
        mo.elastic_mesh_3d(g['x3'], 0.1, (20.0, 25.0, 14.0), bool(poo)),
        g[f'f3_{poo}'], **tol)
    np.testing.assert_allclose(
        mo.elastic_mesh_3d(g['x3b'], 0.05, 16.0, bool(poo)), g[f'f3b_{poo}'],
        **tol)
    np.testing.assert_allclose(
        mo.inplane_force(g['xf'], 0.1, (40.0, 40.0), bool(poo)),
        g[f'ff_{poo}'], **tol)
  planar = ((1, 0, 0), (0, 1, 0), (1, 1, 0), (-1, 1, 0))
  np.testing.assert_allclose(
      mo.elastic_mesh_3d(g['x3'], 0.1, (20.0, 25.0, 14.0), False, links=planar),
      g['f3_planar'], **tol)


@pytest.mark.parametrize('tag', ['fire1', 'fire10', 'fire100', 'fire_cap',
                                 'fire_drift', 'damped10', 'noprev10',
                                 'fire3d_20'])
def test_velocity_verlet_vs_golden(golden, tag):
  g = golden('mesh_vv')
  c = load_cfgs(g)[tag]
  cap = c['_force_cap']
  cfg = cfg_from(c)
  is3d = '3d' in tag
  x0 = g['x30'] if is3d else g['x0']
  prev = None if 'noprev' in tag else (g['prev3'] if is3d else g['prev'])
  force = mo.elastic_mesh_3d if is3d else mo.inplane_force
  st = mo.velocity_verlet(x0, np.zeros_like(x0), prev, cfg, cap,
                          mesh_force=force)
  for i, name in enumerate('xva'):
    np.testing.assert_allclose(st[i], g[f'{tag}_{name}'], rtol=1e-4, atol=5e-5)
  if cfg.fire:
    np.testing.assert_allclose([float(s) for s in st[3:]], g[f'{tag}_scal'],
                               rtol=1e-6)


def test_relax_vs_golden(golden):
  g = golden('mesh_relax')
  cfgs = load_cfgs(g)
  xs, ek, t = mo.relax_mesh(g['xr'], np.zeros_like(g['xr']),
                            cfg_from(cfgs['fire']))
  assert t == int(g['fire_t'])
  np.testing.assert_allclose(xs, g['fire_x'], atol=1e-3)
  np.testing.assert_allclose(ek, g['fire_ekin'], rtol=1e-2, atol=1e-9)
  xs, ek, t = mo.relax_mesh(g['xe'], g['pe'], cfg_from(cfgs['em2d']))
  assert t == int(g['em2d_t'])
  np.testing.assert_allclose(xs, g['em2d_x'], atol=5e-3)
  np.testing.assert_allclose(ek, g['em2d_ekin'], rtol=5e-2, atol=1e-6)


def test_compose_maps_vs_golden(golden):
  from oracle import maps_oracle as mpo
  g = golden('compose_maps')
  for mode in ('nearest', 'constant'):
    got = mpo.compose_maps_fast(g['m1'], (0, 10, 20), (16, 16), g['m2'],
                                (0, -5, 8), (20, 20), mode=mode)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(g[f'c2_{mode}']))
    np.testing.assert_allclose(got, g[f'c2_{mode}'], rtol=1e-5, atol=2e-5)
    got = mpo.compose_maps_fast(g['n1'], (1, 2, 3), (8, 10, 10), g['n2'],
                                (0, 1, 2), (8, 10, 10), mode=mode)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(g[f'c3_{mode}']))
    np.testing.assert_allclose(got, g[f'c3_{mode}'], rtol=1e-5, atol=2e-5)
  # reference KAT (tests/map_utils_test.py:266-301)
  coord_map = np.zeros([2, 1, 60, 60])
  flow = np.zeros([2, 1, 50, 50])
  flow[0, 0, :, 10:25] = -5
  flow[0, 0, :, 25:40] = 65
  flow[:, 0, :, 4] = np.nan
  coord_map[0, :, :, 7:] = -10
  upd = mpo.compose_maps_fast(flow, (64, 58, 42), 40, coord_map, (64, 50, 40), 40)
  flow[0, 0, :, 5:10] = -10
  flow[0, 0, :, 10:25] = -15
  flow[0, 0, :, 25:40] = 55
  flow[0, 0, :, 40:] = -10
  np.testing.assert_array_equal(upd, flow)


def test_target_mesh_and_montage_relax_vs_golden(golden):
  import json
  from oracle import maps_oracle as mpo
  g = golden('montage')
  stride = tuple(g['stride'])
  for xin, want in ((g['x'], g['tg0']), (g['xs'], g['tg1'])):
    got = mpo.target_mesh_all(g['nbors'], xin, g['fx'], g['fy'], stride)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-4)
  cfg = cfg_from(json.loads(str(g['cfg'])))
  prev_fn = lambda xx: mpo.target_mesh_all(g['nbors'], xx, g['fx'], g['fy'], stride)
  xs, ek, t = mo.relax_mesh(g['x'], None, cfg, prev_fn=prev_fn)
  assert t == int(g['t'])
  np.testing.assert_allclose(xs, g['relaxed'], atol=2e-3)
  np.testing.assert_allclose(ek, g['ekin'], rtol=2e-2)


def test_clean_flow_vs_golden(golden):
  """flow_utils.clean_flow restatement == the reference's output, exactly."""
  from oracle import flow_utils_oracle as fu
  g = golden('clean_flow')
  p2 = [float(v) for v in g['p2']]
  np.testing.assert_array_equal(fu.clean_flow(g['f2'], *p2), g['c2'])
  np.testing.assert_array_equal(
      fu.clean_flow(g['f2'], p2[0], p2[1], 0.0, p2[3]), g['c2_nomag'])
  np.testing.assert_array_equal(
      fu.clean_flow(g['f2'], p2[0], p2[1], p2[2], 0.0), g['c2_nodev'])
  np.testing.assert_array_equal(fu.clean_flow(g['f2'][:2], *p2), g['c2_2ch'])
  p3 = [float(v) for v in g['p3']]
  np.testing.assert_array_equal(fu.clean_flow(g['f3'], *p3, dim=3), g['c3'])
  np.testing.assert_array_equal(fu.clean_flow(g['f3'][:3], *p3, dim=3), g['c3_3ch'])
  # the filter does something in every variant
  assert np.isnan(g['c2']).sum() > np.isnan(g['f2'][:2]).sum()
  assert np.isnan(g['c3']).sum() > np.isnan(g['f3'][:3]).sum()


_IRREG = (('a', dict(frac=0.25, max_frac=1.1)), ('b', dict(frac=0.4)),
          ('c', dict(frac=0.25, max_frac=1.5, dilation_iters=0)),
          ('d', dict(frac=0.3, max_frac=1.3, dilation_iters=3)))


def test_mask_irregular_vs_golden(golden):
  from oracle import maps_oracle
  g = golden('mask_irregular')
  for tag, kw in _IRREG:
    mm, bad = maps_oracle.mask_irregular(g['m'], (20.0, 16.0), **kw)
    np.testing.assert_array_equal(bad, g['bad_' + tag])
    np.testing.assert_array_equal(mm, g['map_' + tag])
    assert 0 < bad.sum() < bad.size


def test_target_mesh_3d_vs_golden(golden):
  """Volumetric compute_target_mesh restatement vs the reference (11 fields)."""
  from oracle import maps_oracle
  g = golden('montage3d')
  got = maps_oracle.target_mesh_all(g['nbors'], g['x'], g['fx'], g['fy'],
                                    tuple(g['stride']))
  np.testing.assert_array_equal(np.isnan(got), np.isnan(g['tg']))
  np.testing.assert_allclose(got, g['tg'], rtol=1e-5, atol=1e-4)
  assert np.isfinite(g['tg']).mean() > 0.1


def _stitch_tiles(g):
  return {tuple(int(v) for v in k): t for k, t in zip(g['tile_keys'], g['tiles'])}


def test_stitch_rigid_oracle_vs_golden(golden):
  """_estimate_offset, elastic_tile_mesh[_3d], optimize_coarse_mesh."""
  from oracle import stitch_oracle as so
  g = golden('stitch_cfg1')
  tiles = _stitch_tiles(g)
  a, b = tiles[(0, 0)][:, -96:], tiles[(1, 0)][:, :96]
  off, pr = so.estimate_offset(a, b, 10)
  np.testing.assert_array_equal(off + [pr], g['eo0'])
  off, pr = so.estimate_offset(a, b, 40, filter_size=7,
                               masks=(g['eo1_ma'], g['eo1_mb']))
  np.testing.assert_array_equal(off + [pr], g['eo1'])
  off, pr = so.estimate_offset(tiles[(0, 0)][-128:, :], tiles[(0, 1)][:128, :], 0)
  np.testing.assert_array_equal(off + [pr], g['eo2'])
  off, pr = so.estimate_offset(a, b, 250, filter_size=7,
                               masks=(g['eo1_ma'], g['eo1_mb']))
  assert np.isnan(off).all() and np.isnan(pr) and np.isnan(g['eo3']).all()

  np.testing.assert_allclose(so.elastic_tile_mesh(g['tm_x'], g['tm_cx'], g['tm_cy']),
                             g['tm_f'], rtol=1e-6, atol=1e-5)
  np.testing.assert_allclose(
      so.elastic_tile_mesh_3d(g['tm3_x'], g['tm3_cx'], g['tm3_cy']), g['tm3_f'],
      rtol=1e-6, atol=1e-5)
  cfg = cfg_from(__import__('json').loads(str(g['tm_cfg'])))
  np.testing.assert_allclose(so.optimize_coarse_mesh(g['tm_cx'], g['tm_cy'], cfg),
                             g['tm_relaxed'], atol=2e-3)
  np.testing.assert_allclose(
      so.optimize_coarse_mesh(g['tm3_cx'], g['tm3_cy'], cfg,
                              mesh_fn=so.elastic_tile_mesh_3d),
      g['tm3_relaxed'], atol=2e-3)
  np.testing.assert_allclose(so.optimize_coarse_mesh(g['cx'], g['cy']), g['coarse'],
                             atol=2e-3)


def test_flow_map_cfg1_oracle_vs_golden(golden):
  """configs[0] flow leg: the 512^2 tile strips, patch 64, step 32."""
  from oracle import stitch_oracle as so
  g = golden('stitch_cfg1')
  tiles = _stitch_tiles(g)
  for name, conn, axis in (('fx', g['cx'][:, 0], 0), ('fy', g['cy'][:, 0], 1)):
    flows, offs = so.compute_flow_map(tiles, conn, axis, (64, 64), (32, 32), 64)
    keys = [tuple(int(v) for v in k) for k in g[name + '_keys']]
    assert sorted(flows) == sorted(keys)
    for i, k in enumerate(keys):
      assert tuple(offs[k]) == tuple(g[name + '_offsets'][i])
      check_flow(flows[k], g[f'{name}_{i}'])


@pytest.mark.parametrize('case', ['regularized', 'regular', 'prep_failed', 'masked', 'median'])
def test_three_pass_driver_oracle_vs_reference_output(golden, case):
  """mesh_oracle.relax_mesh_passes against RelaxMesh.relax_mesh of the reference
  (processor/mesh.py:428-513, run through the stand-in on an instance carrying
  only `_config`): status, step count, NaN pattern, mesh, energy trace."""
  import json
  import types
  g = golden('relax_passes')
  cfg = types.SimpleNamespace(**json.loads(str(g['cfg'])))
  for k, v in dict(fire=True, f_alpha=0.99, f_inc=1.1, f_dec=0.5, alpha=0.1, n_min=5,
                   cap_scale=1.1, cap_upscale_every=100, remove_drift=False).items():
    if not hasattr(cfg, k):
      setattr(cfg, k, v)
  cfg.stride = tuple(cfg.stride)
  prev = g[f'{case}_prev']
  mask = g[f'{case}_mask']
  mask = None if mask.size == 0 else mask
  start_fn = None
  if bool(g[f'{case}_median']):
    def start_fn(x, p):      # maybe_update_init_state, PREV_MEDIAN (processor/mesh.py:387-398)
      x[0, ...] = np.nanmedian(p[0, ...])
      x[1, ...] = np.nanmedian(p[1, ...])
      return np.nan_to_num(x)
  x, ek, steps, status = mo.relax_mesh_passes(np.zeros_like(prev), prev, cfg, mask,
                                              float(g[f'{case}_frac']), start_fn)
  assert status == int(g[f'{case}_status']) and steps == int(g[f'{case}_steps'])
  want = g[f'{case}_x']
  np.testing.assert_array_equal(np.isnan(x), np.isnan(want))
  scale = np.nanmax(np.abs(want))
  np.testing.assert_allclose(np.nan_to_num(x), np.nan_to_num(want), atol=1e-4 * scale)
  np.testing.assert_allclose(ek, g[f'{case}_ekin'], rtol=1e-3, atol=1e-7)


def test_ndimage_warp_oracle_vs_reference_output(golden):
  """oracle.warp_oracle.ndimage_warp == the arrays the reference's
  warp.ndimage_warp produced (work boxes with overlap, boxes, out_scale,
  uint8 / uint16 / float32, linear and nearest), bit for bit."""
  from oracle import warp_oracle
  from tests.util import ndimage_warp_case
  g = golden('ndimage_warp')
  for name in g['names']:
    img, cmap, stride, _, _, order, boxes, scale, want = ndimage_warp_case(g, str(name))
    kw = {}
    if boxes is not None:
      kw = dict(image_start=boxes['image'][0], map_start=boxes['map'][0],
                out_start=boxes['out'][0], out_size=boxes['out'][1])
    if scale is not None:
      kw['out_scale'] = scale
    got = warp_oracle.ndimage_warp(img, cmap, stride, order=order, **kw)
    assert got.dtype == want.dtype and got.shape == want.shape, name
    np.testing.assert_array_equal(got, want, err_msg=str(name))
