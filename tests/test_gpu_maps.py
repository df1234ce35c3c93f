"""compose_maps_fast and the montage target mesh on the GPU (-m gpu)."""
import json

import numpy as np
import pytest

from tests.util import cfg_from

pytestmark = pytest.mark.gpu


def test_compose_maps_fast_kat(gpu):
  """tests/map_utils_test.py:266-301 of the reference (exact piecewise case)."""
  from sofima_amd import map_utils
  coord_map = np.zeros([2, 1, 60, 60])
  flow = np.zeros([2, 1, 50, 50])
  flow[0, 0, :, 10:25] = -5
  flow[0, 0, :, 25:40] = 65
  flow[:, 0, :, 4] = np.nan
  stride = 40
  start1, start2 = (64, 58, 42), (64, 50, 40)   # zyx = box.start[::-1]
  updated = np.array(map_utils.compose_maps_fast(flow, start1, stride, coord_map,
                                                 start2, stride))
  np.testing.assert_array_equal(updated, flow)
  coord_map[0, :, :, 7:] = -10
  updated = np.array(map_utils.compose_maps_fast(flow, start1, stride, coord_map,
                                                 start2, stride))
  flow[0, 0, :, 5:10] = -10
  flow[0, 0, :, 10:25] = -15
  flow[0, 0, :, 25:40] = 55
  flow[0, 0, :, 40:] = -10
  np.testing.assert_array_equal(updated, flow)


def test_compose_maps_fast_golden(gpu, golden):
  from sofima_amd import map_utils
  g = golden('compose_maps')
  for mode in ('nearest', 'constant'):
    got = np.array(map_utils.compose_maps_fast(
        g['m1'], (0, 10, 20), (16, 16), g['m2'], (0, -5, 8), (20, 20), mode=mode))
    want = g[f'c2_{mode}']
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-5)
    got = np.array(map_utils.compose_maps_fast(
        g['n1'], (1, 2, 3), (8, 10, 10), g['n2'], (0, 1, 2), (8, 10, 10),
        mode=mode))
    want = g[f'c3_{mode}']
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-5)


def test_target_mesh_golden(gpu, golden):
  """compute_target_mesh of a 2 x 2 montage (reference chain over the jax
  stand-in: stitch_rigid -> stitch_elastic -> aggregate_arrays)."""
  from sofima_amd import stitch_elastic
  g = golden('montage')
  fn = stitch_elastic.TargetMeshFn(g['nbors'], g['fx'], g['fy'],
                                   tuple(g['stride']))
  for xin, want in ((g['x'], g['tg0']), (g['xs'], g['tg1'])):
    got = np.array(fn(xin))
    assert got.shape == want.shape
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-4)
  # something was actually pasted for every tile
  assert (np.isfinite(g['tg0'][0]).mean(axis=(1, 2)) > 0.05).all()
  # single-tile form of the reference
  one = stitch_elastic.compute_target_mesh(g['nbors'][1], g['xs'], g['fx'],
                                           g['fy'], tuple(g['stride']))
  np.testing.assert_allclose(one, g['tg1'][:, 1], rtol=1e-5, atol=1e-4)


def test_montage_relaxation_golden(gpu, golden):
  """relax_mesh(x, None, cfg, prev_fn=target mesh) with remove_drift."""
  from sofima_amd import mesh, stitch_elastic
  g = golden('montage')
  cfg = cfg_from(json.loads(str(g['cfg'])), mesh.IntegrationConfig)
  fn = stitch_elastic.TargetMeshFn(g['nbors'], g['fx'], g['fy'],
                                   tuple(g['stride']))
  xs, ek, t = mesh.relax_mesh(g['x'], None, cfg, prev_fn=fn)
  assert t == int(g['t'])
  np.testing.assert_allclose(np.array(xs), g['relaxed'], atol=2e-3)
  np.testing.assert_allclose(ek, g['ekin'], rtol=5e-3)
  with pytest.raises(ValueError):
    mesh.relax_mesh(g['x'], g['x'], cfg, prev_fn=fn)
  # ANY callable is a prev_fn (mesh.py:429-430): the same target mesh behind a
  # plain Python function takes the generic callback path and returns the same
  # trajectory bit for bit (same kernels: advance, target mesh, integrate)
  calls = []

  def wrapped(x):
    calls.append(1)
    return fn(x)

  xs2, ek2, t2 = mesh.relax_mesh(g['x'], None, cfg, prev_fn=wrapped)
  assert t2 == t and ek2 == ek
  np.testing.assert_array_equal(np.array(xs2), np.array(xs))
  # one evaluation per force evaluation: the initial one of every chunk + one per step
  assert len(calls) == t + t // cfg.num_iters


def test_generic_prev_fn_callables(gpu):
  """prev_fn given as arbitrary Python callables (NumPy in / NumPy out, device
  in / device out) follows the oracle driven by the same callable; errors raised
  inside it surface unchanged."""
  from oracle import mesh_oracle
  from sofima_amd import mesh
  rng = np.random.default_rng(5)
  x0 = (rng.standard_normal((2, 2, 20, 24)) * 0.5).astype(np.float32)
  anchor = (rng.standard_normal((2, 2, 20, 24)) * 3).astype(np.float32)
  anchor[:, 0, 3:5, 2:9] = np.nan
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.05, k=0.1, stride=(20, 20),
                               num_iters=40, max_iters=120, stop_v_max=1e-9, dt_max=1000,
                               start_cap=0.1, final_cap=10, prefer_orig_order=True)

  def host_fn(x):        # NumPy in, NumPy out: pulls towards a mix of anchor and mirror image
    x = np.asarray(x)
    return (0.5 * (anchor + x[:, ::-1])).astype(np.float32)

  def dev_fn(x):         # stays on the device
    import torch
    t = x.tensor
    return 0.5 * (torch.from_numpy(anchor).to(t.device) + torch.flip(t, dims=(1,)))

  wx, we, wt = mesh_oracle.relax_mesh(x0, None, cfg, prev_fn=host_fn)
  for f in (host_fn, dev_fn):
    gx, ge, gt = mesh.relax_mesh(x0, None, cfg, prev_fn=f)
    assert gt == wt
    np.testing.assert_allclose(np.array(gx), wx, atol=1e-3 * np.abs(wx).max())
    np.testing.assert_allclose(ge, we, rtol=1e-3)
  out = mesh.velocity_verlet(x0, np.zeros_like(x0), None, cfg, 0.5, prev_fn=host_fn)
  ref = mesh_oracle.velocity_verlet(x0, np.zeros_like(x0), None, cfg, 0.5, prev_fn=host_fn)
  np.testing.assert_allclose(np.array(out[0]), ref[0], atol=1e-4)
  assert out[5] == ref[5]

  class Boom(RuntimeError):
    pass

  def bad(x):
    raise Boom('inside prev_fn')

  with pytest.raises(Boom):
    mesh.relax_mesh(x0, None, cfg, prev_fn=bad)
  with pytest.raises(ValueError):
    mesh.relax_mesh(x0, None, cfg, prev_fn=lambda x: np.zeros((2, 1, 3, 3), np.float32))
  with pytest.raises(ValueError):
    mesh.relax_mesh(x0, x0, cfg, prev_fn=host_fn)


@pytest.mark.gpu
def test_clean_flow_kat(gpu):
  """tests/flow_utils_test.py:38-64 against the HIP kernel."""
  from sofima_amd import flow_utils
  from tests.test_reference_kats import _clean_flow_kat
  flow, kw, expected = _clean_flow_kat()
  np.testing.assert_array_equal(np.asarray(flow_utils.clean_flow(flow, **kw)), expected)


@pytest.mark.gpu
def test_clean_flow_golden(gpu, golden):
  """sfm_clean_flow == reference clean_flow, bit for bit (values pass through)."""
  from sofima_amd import flow_utils
  g = golden('clean_flow')
  p2 = [float(v) for v in g['p2']]
  cases = [
      (g['f2'], p2, 2, g['c2']),
      (g['f2'], [p2[0], p2[1], 0.0, p2[3]], 2, g['c2_nomag']),
      (g['f2'], [p2[0], p2[1], p2[2], 0.0], 2, g['c2_nodev']),
      (g['f2'][:2], p2, 2, g['c2_2ch']),
      (g['f3'], [float(v) for v in g['p3']], 3, g['c3']),
      (g['f3'][:3], [float(v) for v in g['p3']], 3, g['c3_3ch']),
  ]
  for flow, p, dim, want in cases:
    got = np.asarray(flow_utils.clean_flow(flow, *p, dim=dim))
    np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
def test_clean_flow_on_flow_field_output(gpu):
  """flow_field -> clean_flow stays consistent with the oracle chain."""
  from oracle import flow_utils_oracle
  from sofima_amd import flow_field, flow_utils
  rng = np.random.default_rng(3)
  from scipy import ndimage
  base = ndimage.gaussian_filter(rng.standard_normal((400, 400)), 2.0)
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  pre, post = base[8:392, 8:392], base[5:389, 11:395].copy()
  post[100:160, 100:200] = 7  # featureless region: unreliable vectors
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  flow = calc.flow_field(pre, post, 64, 16, batch_size=256)[:, np.newaxis]
  got = np.asarray(flow_utils.clean_flow(flow, 1.4, 1.6, 20, 4))
  want = flow_utils_oracle.clean_flow(flow, 1.4, 1.6, 20, 4)
  np.testing.assert_array_equal(got, want)
  assert 0 < np.isnan(got[0]).sum() < got[0].size


@pytest.mark.gpu
def test_mask_irregular_kat_and_golden(gpu, golden):
  """tests/map_utils_test.py:323-333 and the reference goldens, in-place contract."""
  from sofima_amd import map_utils
  from tests.test_oracle_golden import _IRREG
  coord_map = np.zeros([2, 50, 50])
  coord_map[0, 40, 10] = 10
  bad = map_utils.mask_irregular(coord_map, (40, 40), 0.25, 1.1)
  expected = np.zeros([2, 50, 50])
  expected[:, 39:42, 8:11] = np.nan
  np.testing.assert_array_equal(expected, coord_map)
  np.testing.assert_array_equal(np.isnan(expected[0, ...]), bad)
  g = golden('mask_irregular')
  for tag, kw in _IRREG:
    mm = g['m'].copy()
    bad = map_utils.mask_irregular(mm, (20.0, 16.0), **kw)
    np.testing.assert_array_equal(bad, g['bad_' + tag])
    np.testing.assert_array_equal(mm, g['map_' + tag])
  # device-resident input is modified in place on the device
  import torch
  t = torch.from_numpy(g['m'].copy()).cuda()
  bad = map_utils.mask_irregular(t, (20.0, 16.0), frac=0.25, max_frac=1.1)
  np.testing.assert_array_equal(t.cpu().numpy(), g['map_a'])


@pytest.mark.gpu
def test_target_mesh_3d_golden_and_relax(gpu, golden):
  """Volumetric montage: sfm_target_mesh vs the reference, and as the native
  prev_fn of a 3-D relaxation vs the oracle."""
  from oracle import maps_oracle, mesh_oracle
  from sofima_amd import mesh, stitch_elastic
  g = golden('montage3d')
  stride = tuple(float(v) for v in g['stride'])
  fn = stitch_elastic.TargetMeshFn(g['nbors'], g['fx'], g['fy'], stride)
  got = np.array(fn(g['x']))
  np.testing.assert_array_equal(np.isnan(got), np.isnan(g['tg']))
  np.testing.assert_allclose(got, g['tg'], rtol=1e-5, atol=1e-4)
  one = stitch_elastic.compute_target_mesh(g['nbors'][2], g['x'], g['fx'], g['fy'], stride)
  np.testing.assert_allclose(one, g['tg'][:, 2], rtol=1e-5, atol=1e-4)
  # relaxation with the volumetric target mesh as prev_fn
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.05, k=0.1,
                               stride=stride[::-1], num_iters=30, max_iters=30,
                               stop_v_max=1e-9, dt_max=100, start_cap=1.0,
                               final_cap=10.0, prefer_orig_order=False)
  x0 = np.nan_to_num(g['x']).astype(np.float32)
  gx, ge, gt = mesh.relax_mesh(x0, None, cfg, mesh_force=mesh.elastic_mesh_3d,
                               prev_fn=fn)
  wx, we, wt = mesh_oracle.relax_mesh(
      x0, None, cfg, mesh_force=mesh_oracle.elastic_mesh_3d,
      prev_fn=lambda xx: maps_oracle.target_mesh_all(g['nbors'], xx, g['fx'], g['fy'],
                                                     stride))
  assert gt == wt == 30
  np.testing.assert_allclose(np.array(gx), wx, atol=2e-3 * np.abs(wx).max())
  np.testing.assert_allclose(ge, we, rtol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('variant', ['fire_drift', 'fire', 'verlet'])
def test_fused_target_step_is_bit_identical(gpu, variant):
  """Tiled in-plane montage with the native prev_fn: target mesh sampled from the
  advanced positions on the overlap strips + fused integrator (2 launches per
  step) == advance + full target mesh + integrate (SFM_MESH_FUSE_TARGET=0), bit
  for bit, over several chunks; and it follows the oracle."""
  from oracle import maps_oracle, mesh_oracle
  from sofima_amd import _abi, mesh, stitch_elastic
  from tests.util import synth_montage
  rng = np.random.default_rng(31)
  nb, fx, fy, x0 = synth_montage(rng, 3, 2, (52, 70), 12)
  stride = (20.0, 20.0)
  kw = dict(dt=0.001, gamma=0.0, k0=0.02, k=0.1, stride=stride, num_iters=40,
            max_iters=120, stop_v_max=1e-9, dt_max=100, prefer_orig_order=True,
            start_cap=0.1, final_cap=10.0, remove_drift=(variant == 'fire_drift'))
  if variant == 'verlet':
    kw.update(fire=False, gamma=0.5, dt=0.05, start_cap=10.0)
  cfg = mesh.IntegrationConfig(**kw)
  fn = stitch_elastic.TargetMeshFn(nb, fx, fy, stride)
  a = mesh.relax_mesh(x0, None, cfg, prev_fn=fn)
  with _abi.option('SFM_MESH_FUSE_TARGET', 0):
    b = mesh.relax_mesh(x0, None, cfg, prev_fn=fn)
  np.testing.assert_array_equal(np.array(a[0]), np.array(b[0]))
  assert a[1] == b[1] and a[2] == b[2]
  want = mesh_oracle.relax_mesh(
      x0, None, cfg, prev_fn=lambda xx: maps_oracle.target_mesh_all(nb, xx, fx, fy, stride))
  assert a[2] == want[2]
  np.testing.assert_allclose(np.array(a[0]), want[0], atol=1e-3 * np.abs(want[0]).max())


