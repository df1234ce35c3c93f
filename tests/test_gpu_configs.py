"""BASELINE.json configs at their workload sizes on the HIP path (-m gpu).

configs[0]  2 x 2 grid of 512^2 tiles, patch 64 step 32: flow leg vs the
            reference's own output (stitch_cfg1.npz), mesh leg in
            test_gpu_maps.py (montage.npz)
configs[1]  8192^2 pair, patch 160 step 40 batch 1024: warped pair, the WHOLE
            field (all 40 reference batches, 40401 patches) vs the oracle; the
            masked pair on four whole batches (the rigid-shift properties are
            in test_gpu_flow.py)
configs[2]  8 x 8 montage of 4096^2 tiles: a 4096 x 400 overlap strip, patch 120
            step 20 batch 256, and the [2, 64, 204, 204] mesh with the native
            target-mesh prev_fn + remove_drift, 100 FIRE steps, vs the oracle
configs[4]  3-D: 80^3 patches step 40 on the overlap strip of two 512^3 tiles
            and the [3, 64, 12, 12, 12] mesh with the volumetric target mesh,
            vs the oracle
configs[3] (64 sections over 8 GPUs) is the section-parallel form of
configs[1]; its block chain is covered by the gloo tests.
"""
import numpy as np
import pytest

from oracle import flow_oracle, maps_oracle, mesh_oracle
from tests.util import check_flow, check_sharpness, em_texture, synth_montage

pytestmark = pytest.mark.gpu


def test_cfg0_flow_leg_vs_reference_output(gpu, golden):
  """stitch_elastic.compute_flow_map on the four 512^2 tiles == the arrays the
  reference's compute_flow_map produced (flow vectors exact)."""
  from sofima_amd import stitch_elastic
  g = golden('stitch_cfg1')
  tiles = {tuple(int(v) for v in k): t for k, t in zip(g['tile_keys'], g['tiles'])}
  for name, conn, axis in (('fx', g['cx'][:, 0], 0), ('fy', g['cy'][:, 0], 1)):
    flows, offs = stitch_elastic.compute_flow_map(
        tiles, conn, axis, patch_size=(64, 64), stride=(32, 32), batch_size=64)
    keys = [tuple(int(v) for v in k) for k in g[name + '_keys']]
    assert sorted(flows) == sorted(keys)
    for i, k in enumerate(keys):
      assert tuple(offs[k]) == tuple(g[name + '_offsets'][i])
      check_flow(flows[k], g[f'{name}_{i}'])


def _warp(img, amp, lam):
  """Samples img through a smooth displacement field."""
  from scipy import ndimage
  h, w = img.shape
  out = np.empty_like(img)
  rows = 1024
  xx = np.arange(w, dtype=np.float32)[None, :]
  for y0 in range(0, h, rows):
    yy = np.arange(y0, min(y0 + rows, h), dtype=np.float32)[:, None]
    d = amp * np.sin(2 * np.pi * xx / lam) * np.cos(2 * np.pi * yy / lam)
    blk = ndimage.map_coordinates(img, [yy + d, xx - d], order=1, mode='nearest',
                                  output=np.float32)
    out[y0:y0 + rows] = np.clip(np.rint(blk), 0, 255).astype(np.uint8)
  return out


def test_cfg1_full_size_warped_pair_whole_field_vs_oracle(gpu, tmp_path):
  """8192^2 pair whose second image is sampled through a smooth 6 px warp: the
  WHOLE [4, 201, 201] field -- all 40 reference batches of 1024 (the last one
  ragged: 465 patches), 40401 patches, same batch membership as the reference
  -- agrees with the oracle vector for vector (the oracle runs one process per
  batch on the host cores)."""
  from sofima_amd import flow_field as ff
  from tests.util import oracle_flow_batches
  rng = np.random.default_rng(1002)
  size = 8192
  base = em_texture(rng, (size + 32, size + 32))
  pre = np.ascontiguousarray(base[16:16 + size, 16:16 + size])
  post = _warp(np.ascontiguousarray(base[19:19 + size, 11:11 + size]), 6.0, 2048.0)
  calc = ff.JAXMaskedXCorrWithStatsCalculator()
  got = calc.flow_field(pre, post, 160, 40, batch_size=1024)
  assert got.shape == (4, 201, 201)
  parts = oracle_flow_batches(tmp_path, pre, post, 160, 40, 1024, range(40))
  assert sorted(parts) == list(range(40))
  want = np.concatenate([parts[b] for b in range(40)], axis=1)
  assert want.shape == (4, 201 * 201)
  want = want.reshape(4, 201, 201)
  assert np.isfinite(want[:2]).all()
  np.testing.assert_array_equal(got[:2], want[:2])
  np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
  np.testing.assert_allclose(got[3], want[3], rtol=1e-4, atol=1e-6)
  # sharpness = peak / min(11 x 11 window) of a RAW surface: ill conditioned where
  # the window minimum is close to 0 (3 of the 40401 patches differ by up to 8 %
  # in the quotient): value to 2e-4 or window minimum to 2e-5 x |peak|
  ok = np.isfinite(want[2])
  check_sharpness(got[2][ok], want[2][ok])
  # content shift (-5, 3) plus the warp, |d| <= 6 px
  valid = np.isfinite(got[0])
  assert valid.mean() > 0.99
  assert np.abs(got[0][valid] + 5).max() <= 7 and np.abs(got[1][valid] - 3).max() <= 7
  assert np.ptp(got[0][valid]) >= 8          # the warp is visible in the field
  # Exact tile pruning at full size: 40401 patches through 512 resident
  # workgroups whose seed probes carry state from patch to patch -- the field
  # with every tile computed (SFM_MFMA_PRUNE=0) is the same, bit for bit, and so
  # is a second pruned run (the decisions depend on wave timing, the result not).
  from sofima_amd import _abi
  with _abi.option('SFM_MFMA_PRUNE', 0):
    full = calc.flow_field(pre, post, 160, 40, batch_size=1024)
  np.testing.assert_array_equal(got, full)
  np.testing.assert_array_equal(calc.flow_field(pre, post, 160, 40, batch_size=1024), full)


def test_cfg1_full_size_masked_pair_batches_vs_oracle(gpu, tmp_path):
  """The same geometry with masks used IN the correlation
  (mask_only_for_patch_selection=False, discs on both images, max_masked > 1 so
  that no patch is dropped and batch membership is the plain row-major one):
  four whole reference batches -- first, two from the middle, the ragged last --
  hold patches of all four mask classes and agree with the oracle's Padfield
  surface statistics (batch-global tolerances included)."""
  from bench import synth_pair
  from sofima_amd import flow_field as ff
  from tests.util import oracle_flow_batches
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
  kw = dict(batch_size=1024, mask_only_for_patch_selection=False, max_masked=1.01)
  got = calc.flow_field(pre, post, 160, 40, pre_mask=masks[0], post_mask=masks[1], **kw)
  batches = (0, 13, 26, 39)
  parts = oracle_flow_batches(tmp_path, pre, post, 160, 40, 1024, batches,
                              pre_mask=masks[0], post_mask=masks[1],
                              mask_only_for_patch_selection=False, max_masked=1.01)
  g = got.reshape(4, -1)
  cnt_a = (ff._masked_counts(masks[0], (160, 160), (40, 40)) > 0).ravel()
  cnt_b = (ff._masked_counts(masks[1], (160, 160), (40, 40)) > 0).ravel()
  seen = set()
  n = 0
  for b in batches:
    w = parts[b]
    sl = slice(b * 1024, b * 1024 + w.shape[1])
    n += w.shape[1]
    seen |= set(zip(cnt_a[sl].tolist(), cnt_b[sl].tolist()))
    np.testing.assert_array_equal(np.isnan(g[:, sl]), np.isnan(w), err_msg=str(b))
    np.testing.assert_array_equal(g[:2, sl], w[:2], err_msg=str(b))
    np.testing.assert_allclose(g[3, sl], w[3], rtol=1e-4, atol=1e-6, err_msg=str(b))
    ok = np.isfinite(w[2])
    check_sharpness(g[2, sl][ok], w[2][ok])
  assert n == 3 * 1024 + 465
  assert len(seen) == 4, seen          # clean / pre only / post only / both masked


def test_cfg2_montage_strip_vs_oracle(gpu):
  """One 4096 x 400 overlap strip of the 8 x 8 montage of 4096^2 tiles:
  patch 120, step 20, batch 256 (stitch_elastic.compute_flow_map defaults)."""
  from sofima_amd import flow_field as ff
  rng = np.random.default_rng(1003)
  base = em_texture(rng, (4096 + 24, 400 + 24))
  pre = np.ascontiguousarray(base[12:12 + 4096, 12:12 + 400])
  post = _warp(np.ascontiguousarray(base[10:10 + 4096, 16:16 + 400]), 3.0, 900.0)
  calc = ff.JAXMaskedXCorrWithStatsCalculator()
  got = calc.flow_field(pre, post, (120, 120), (20, 20), batch_size=256)
  assert got.shape == (4, 199, 15)
  want = flow_oracle.flow_field(pre, post, (120, 120), (20, 20), batch_size=256,
                                workers=16)
  check_flow(got, want)


def test_cfg2_montage_mesh_vs_oracle(gpu):
  """[2, 64, 204, 204] montage mesh, native target-mesh prev_fn, remove_drift,
  prefer_orig_order: 100 FIRE steps follow the oracle (same branch sequence:
  identical dt, alpha, n_pos, cap)."""
  from sofima_amd import mesh, stitch_elastic
  rng = np.random.default_rng(1004)
  nb, fx, fy, x0 = synth_montage(rng, 8, 8, (204, 204), 20)
  stride = (20.0, 20.0)
  cfg = mesh.IntegrationConfig(
      dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=stride, num_iters=100,
      max_iters=100, stop_v_max=0.001, dt_max=100, prefer_orig_order=True,
      start_cap=0.1, final_cap=10.0, remove_drift=True)
  fn = stitch_elastic.TargetMeshFn(nb, fx, fy, stride)
  got = mesh.velocity_verlet(x0, np.zeros_like(x0), None, cfg, cfg.start_cap,
                             prev_fn=fn)
  want = mesh_oracle.velocity_verlet(
      x0, np.zeros_like(x0), None, cfg, cfg.start_cap,
      prev_fn=lambda xx: maps_oracle.target_mesh_all(nb, xx, fx, fy, stride))
  assert got[5] == want[5]                                   # n_pos
  np.testing.assert_allclose([got[3], got[4], got[6]],
                             [want[3], want[4], want[6]], rtol=1e-6)
  scale = np.abs(want[0]).max()
  np.testing.assert_allclose(np.array(got[0]), want[0], atol=1e-3 * scale)
  np.testing.assert_allclose(np.array(got[1]), want[1],
                             atol=2e-3 * np.abs(want[1]).max())


def test_cfg4_3d_overlap_strip_flow_vs_oracle(gpu):
  """80^3 patches, step 40 (liconn notebook) on the overlap strip of two
  512^3 tiles (z 512, y cropped to 200 to bound the oracle's time, x 120)."""
  from sofima_amd import flow_field as ff
  rng = np.random.default_rng(1005)
  base = em_texture(rng, (512 + 8, 200 + 8, 120 + 8), 1.5)
  pre = np.ascontiguousarray(base[4:516, 4:204, 4:124])
  post = np.ascontiguousarray(base[6:518, 1:201, 7:127])
  calc = ff.JAXMaskedXCorrWithStatsCalculator()
  got = calc.flow_field(pre, post, (80, 80, 80), 40, batch_size=16)
  assert got.shape == (5, 11, 4, 2)
  # flow = position in pre - position in post of the same content, as (x, y, z)
  np.testing.assert_array_equal(got[0], 3)
  np.testing.assert_array_equal(got[1], -3)
  np.testing.assert_array_equal(got[2], 2)
  want = flow_oracle.flow_field(pre, post, (80, 80, 80), 40, batch_size=16,
                                workers=16)
  check_flow(got, want)


def test_cfg4_3d_montage_mesh_vs_oracle(gpu):
  """[3, 64, 12, 12, 12] volumetric montage mesh: elastic_mesh_3d, volumetric
  target mesh as prev_fn, remove_drift (per x column, reference quirk)."""
  from sofima_amd import mesh, stitch_elastic
  rng = np.random.default_rng(1006)
  nb, fx, fy, x0 = synth_montage(rng, 8, 8, (12, 12, 12), 3, amp=5.0)
  stride = (40.0, 40.0, 40.0)
  cfg = mesh.IntegrationConfig(
      dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=stride, num_iters=100,
      max_iters=100, stop_v_max=0.001, dt_max=100, prefer_orig_order=False,
      start_cap=0.1, final_cap=10.0, remove_drift=True)
  fn = stitch_elastic.TargetMeshFn(nb, fx, fy, stride)
  got = mesh.velocity_verlet(x0, np.zeros_like(x0), None, cfg, cfg.start_cap,
                             mesh_force=mesh.elastic_mesh_3d, prev_fn=fn)
  want = mesh_oracle.velocity_verlet(
      x0, np.zeros_like(x0), None, cfg, cfg.start_cap,
      mesh_force=mesh_oracle.elastic_mesh_3d,
      prev_fn=lambda xx: maps_oracle.target_mesh_all(nb, xx, fx, fy, stride))
  assert got[5] == want[5]
  np.testing.assert_allclose([got[3], got[4], got[6]],
                             [want[3], want[4], want[6]], rtol=1e-6)
  scale = np.abs(want[0]).max()
  np.testing.assert_allclose(np.array(got[0]), want[0], atol=1e-3 * scale)


@pytest.mark.parametrize('grid,mesh_shape,drift', [((8, 8), (12, 12, 12), True),
                                                  ((8, 8), (12, 12, 12), False),
                                                  ((2, 2), (12, 12, 12), True),
                                                  ((3, 2), (8, 16, 12), True),
                                                  ((2, 3), (4, 8, 10), True)])
def test_cfg4_montage_mesh_single_launch_is_bit_identical(gpu, grid, mesh_shape, drift):
  """mesh_persist3d_kernel (every step of a chunk of the volumetric montage in ONE launch:
  the blocks of advance / target mesh / integrate / column means with grid barriers in
  between; opt-in, SFM_MESH_PERSIST3D=1) against the four-launch step: positions, velocities,
  accelerations, FIRE scalars, e_kin and step counts are the same bits -- per-column drift
  means and none, 64 and 4 tiles, a tile size 64 does not divide (the last case: the
  single launch declines it, both runs take the four-launch step), chunks of 1, 2 and 37
  steps, two chunks in a row through relax_mesh."""
  import dataclasses
  from sofima_amd import _abi, mesh, stitch_elastic
  rng = np.random.default_rng(sum(mesh_shape) + grid[0])
  nb, fx, fy, x0 = synth_montage(rng, grid[0], grid[1], mesh_shape, 3, amp=5.0)
  stride = (40.0, 40.0, 40.0)
  base = mesh.IntegrationConfig(
      dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=stride, num_iters=37,
      max_iters=74, stop_v_max=1e-9, dt_max=100, prefer_orig_order=False,
      start_cap=0.1, final_cap=10.0, remove_drift=drift)
  fn = stitch_elastic.TargetMeshFn(nb, fx, fy, stride)
  for iters in (1, 2, 37):
    cfg = dataclasses.replace(base, num_iters=iters, max_iters=iters)
    runs = []
    for single in (1, 0):
      with _abi.option('SFM_MESH_PERSIST3D', single):
        runs.append(mesh.velocity_verlet(x0, np.zeros_like(x0), None, cfg, cfg.start_cap,
                                         mesh_force=mesh.elastic_mesh_3d, prev_fn=fn))
    a, b = runs
    for i in range(3):
      np.testing.assert_array_equal(np.array(a[i]), np.array(b[i]), err_msg=f'{iters} steps, array {i}')
    assert tuple(a[3:]) == tuple(b[3:]), (iters, a[3:], b[3:])
  res = []
  for single in (1, 0):
    with _abi.option('SFM_MESH_PERSIST3D', single):
      res.append(mesh.relax_mesh(x0, None, base, mesh_force=mesh.elastic_mesh_3d, prev_fn=fn))
  np.testing.assert_array_equal(np.array(res[0][0]), np.array(res[1][0]))
  assert res[0][1] == res[1][1] and res[0][2] == res[1][2] == 74


def test_cfg4_flow_map3d_vs_reference_output(gpu, golden):
  """stitch_elastic.compute_flow_map3d (stitch_elastic.py:85-194) on a 2 x 2 grid
  of 3-d tiles == the flow arrays and offsets the reference's function produced
  (flow_map3d.npz): the 3-d flow leg of the volumetric montage, pinned to the
  reference's own output."""
  from sofima_amd import stitch_elastic
  g = golden('flow_map3d')
  tiles = {tuple(int(v) for v in k): t for k, t in zip(g['tile_keys'], g['tiles'])}
  shape = tuple(int(v) for v in g['tile_shape'])
  for name, om, axis in (('fx', g['ox'], 0), ('fy', g['oy'], 1)):
    flows, offs = stitch_elastic.compute_flow_map3d(
        tiles, shape, om, axis, patch_size=tuple(g['patch']), stride=tuple(g['stride']),
        batch_size=8)
    keys = [tuple(int(v) for v in k) for k in g[name + '_keys']]
    assert sorted(flows) == sorted(keys)
    for i, k in enumerate(keys):
      assert tuple(offs[k]) == tuple(g[name + '_offsets'][i])
      check_flow(flows[k], g[f'{name}_{i}'])
