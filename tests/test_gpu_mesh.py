"""HIP mesh kernels vs golden vectors and the CPU oracle (-m gpu)."""
import functools

import numpy as np
import pytest

from oracle import mesh_oracle
from tests.util import cfg_from, load_cfgs

pytestmark = pytest.mark.gpu


def test_equilibrium_is_exactly_zero(gpu):
  from sofima_amd import mesh
  x = np.zeros((2, 1, 10, 10))
  f = np.array(mesh.inplane_force(x, k=1.0, stride=(40.0, 40.0)))
  np.testing.assert_array_equal(x, f)
  x = np.zeros((3, 10, 10, 10))
  f = np.array(mesh.elastic_mesh_3d(x, k=1.0, stride=40.0))
  np.testing.assert_array_equal(x, f)
  x = np.zeros((3, 5, 10, 10, 10))
  f = np.array(mesh.elastic_mesh_3d(x, k=1.0, stride=40.0))
  np.testing.assert_array_equal(x, f)
  # arbitrary strides stay exactly at rest too
  x = np.zeros((3, 4, 5, 6))
  f = np.array(mesh.elastic_mesh_3d(x, k=0.3, stride=(14.3, 9.7, 31.1)))
  np.testing.assert_array_equal(x, f)


def test_force_known_answer(gpu):
  """tests/mesh_test.py:82-120 of the reference, re-typed."""
  from sofima_amd import mesh
  x = np.zeros((2, 1, 10, 10))
  dx, dy = 4, -3
  x[0, 0, 5, 5] = dx
  x[1, 0, 5, 5] = dy
  k, l0 = 0.1, 10.0
  f = np.array(mesh.inplane_force(x, k=k, stride=(l0, 10)))
  l = np.sqrt((l0 + dx) ** 2 + dy**2)
  np.testing.assert_allclose(
      [k * (l - l0) * (l0 + dx) / l, k * (l - l0) * dy / l], f[:, 0, 5, 4],
      rtol=1e-6)
  l = np.sqrt(dx**2 + (l0 + dy) ** 2)
  np.testing.assert_allclose(
      [k * (l - l0) * dx / l, k * (l - l0) * (l0 + dy) / l], f[:, 0, 4, 5],
      rtol=1e-6)
  l = np.sqrt((l0 - dx) ** 2 + (l0 - dy) ** 2)
  l2 = l0 * np.sqrt(2.0)
  k2 = k / np.sqrt(2.0)
  np.testing.assert_allclose(
      [-k2 * (l - l2) * (l0 - dx) / l, -k2 * (l - l2) * (l0 - dy) / l],
      f[:, 0, 6, 6], rtol=1e-5)
  l = np.sqrt((l0 + dx) ** 2 + (l0 - dy) ** 2)
  np.testing.assert_allclose(
      [k2 * (l - l2) * (l0 + dx) / l, -k2 * (l - l2) * (l0 - dy) / l],
      f[:, 0, 6, 4], rtol=1e-5)


def test_2d_3d_consistency(gpu):
  """tests/mesh_test.py:122-144."""
  from sofima_amd import mesh
  planar = ((1, 0, 0), (0, 1, 0), (1, 1, 0), (-1, 1, 0))
  rng = np.random.default_rng(42)
  x = rng.random((3, 1, 50, 50))
  x[2, ...] = 0.0
  for poo in (False, True):
    f2 = np.array(mesh.inplane_force(x[:2], 0.01, (40.0, 40.0), poo))
    f3 = np.array(mesh.elastic_mesh_3d(x, 0.01, (40.0, 40.0, 14.0), poo,
                                       links=planar))
    np.testing.assert_allclose(f2[:2], f3[:2], atol=1e-5)


def test_forces_match_golden(gpu, golden):
  from sofima_amd import mesh
  g = golden('mesh_force')
  tol = dict(rtol=1e-5, atol=2e-6)
  for poo in (0, 1):
    np.testing.assert_allclose(
        np.array(mesh.inplane_force(g['x2'], 0.1, (40.0, 30.0), bool(poo))),
        g[f'f2_{poo}'], **tol)
    np.testing.assert_allclose(
        np.array(mesh.elastic_mesh_3d(g['x3'], 0.1, (20.0, 25.0, 14.0),
                                      bool(poo))), g[f'f3_{poo}'], **tol)
    np.testing.assert_allclose(
        np.array(mesh.elastic_mesh_3d(g['x3b'], 0.05, 16.0, bool(poo))),
        g[f'f3b_{poo}'], **tol)
    np.testing.assert_allclose(
        np.array(mesh.inplane_force(g['xf'], 0.1, (40.0, 40.0), bool(poo))),
        g[f'ff_{poo}'], **tol)
  planar = ((1, 0, 0), (0, 1, 0), (1, 1, 0), (-1, 1, 0))
  np.testing.assert_allclose(
      np.array(mesh.elastic_mesh_3d(g['x3'], 0.1, (20.0, 25.0, 14.0), False,
                                    links=planar)), g['f3_planar'], **tol)


def test_forces_match_oracle_bitwise_mostly(gpu):
  """Random large mesh: the HIP stencil and the oracle agree to float32
  round-off (same operation order, no FMA)."""
  from sofima_amd import mesh
  rng = np.random.default_rng(7)
  x = (rng.standard_normal((2, 3, 65, 131)) * 5).astype(np.float32)
  for poo in (False, True):
    got = np.array(mesh.inplane_force(x, 0.1, (40.0, 40.0), poo))
    want = mesh_oracle.inplane_force(x, 0.1, (40.0, 40.0), poo)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
  x = (rng.standard_normal((3, 2, 9, 33, 17)) * 3).astype(np.float32)
  for poo in (False, True):
    got = np.array(mesh.elastic_mesh_3d(x, 0.1, (40.0, 40.0, 30.0), poo))
    want = mesh_oracle.elastic_mesh_3d(x, 0.1, (40.0, 40.0, 30.0), poo)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize('tag', ['fire1', 'fire10', 'fire100', 'fire_cap',
                                 'fire_drift', 'damped10', 'noprev10',
                                 'fire3d_20'])
def test_velocity_verlet_matches_golden(gpu, golden, tag):
  from sofima_amd import mesh
  g = golden('mesh_vv')
  c = load_cfgs(g)[tag]
  cap = c['_force_cap']
  cfg = cfg_from(c, mesh.IntegrationConfig)
  is3d = '3d' in tag
  x0 = g['x30'] if is3d else g['x0']
  prev = None if 'noprev' in tag else (g['prev3'] if is3d else g['prev'])
  force = mesh.elastic_mesh_3d if is3d else mesh.inplane_force
  st = mesh.velocity_verlet(x0, np.zeros_like(x0), prev, cfg, cap,
                            mesh_force=force)
  for i, name in enumerate('xva'):
    np.testing.assert_allclose(np.array(st[i]), g[f'{tag}_{name}'], rtol=1e-4,
                               atol=5e-5, err_msg=f'{tag} {name}')
  if cfg.fire:
    scal = g[f'{tag}_scal']
    np.testing.assert_allclose([float(s) for s in st[3:]], scal, rtol=1e-6)
    assert int(st[5]) == int(scal[2])
  else:
    assert len(st) == 3


def test_relax_matches_golden(gpu, golden):
  from sofima_amd import mesh
  g = golden('mesh_relax')
  cfgs = load_cfgs(g)
  cfg = cfg_from(cfgs['fire'], mesh.IntegrationConfig)
  xs, ek, t = mesh.relax_mesh(g['xr'], np.zeros_like(g['xr']), cfg)
  assert t == int(g['fire_t'])
  # 300 steps; measured on MI355X: max |dx| 1.7e-6 (positions up to 6.5e-4),
  # kinetic energies within 4.3e-4 relative
  np.testing.assert_allclose(np.array(xs), g['fire_x'], atol=1e-5)
  np.testing.assert_allclose(ek, g['fire_ekin'], rtol=1e-3, atol=1e-9)
  cfg = cfg_from(cfgs['em2d'], mesh.IntegrationConfig)
  xs, ek, t = mesh.relax_mesh(g['xe'], g['pe'], cfg)
  assert t == int(g['em2d_t'])
  # 400 FIRE steps with force capping amplify the round-off of the first steps
  # (same step count and FIRE branch sequence): measured max |dx| 1.6e-3 on
  # positions up to 4.2 (3.8e-4 relative), kinetic energies within 2.5 %
  # (test_em2d_divergence_is_amplified_roundoff follows the two trajectories)
  np.testing.assert_allclose(np.array(xs), g['em2d_x'], atol=2.5e-3)
  np.testing.assert_allclose(ek, g['em2d_ekin'], rtol=3e-2, atol=1e-6)


def test_em2d_divergence_is_amplified_roundoff(gpu, golden, capsys):
  """Why `em2d` carries atol 2.5e-3 instead of SURVEY 8c's 1e-3: the HIP and the
  oracle trajectories of that case are followed chunk by chunk (20 steps each,
  the FIRE scalars handed over like relax_mesh does).  The FIRE scalars stay
  IDENTICAL (same branch at every step); the positions agree to round-off for
  the first 80 steps and separate in one burst while FIRE's time step is at its
  largest: the end-state difference is round-off amplified by the dynamics, not
  a different algorithm."""
  import dataclasses
  from sofima_amd import mesh
  g = golden('mesh_relax')
  cfg = cfg_from(load_cfgs(g)['em2d'], mesh.IntegrationConfig)
  cfg = dataclasses.replace(cfg, num_iters=20, max_iters=400)
  ocfg = cfg_from(dataclasses.asdict(cfg))
  x0, prev = g['xe'], g['pe']
  gx, gv = x0.copy(), np.zeros_like(x0)
  wx, wv = x0.copy(), np.zeros_like(x0)
  dt, alpha, cap = cfg.dt, cfg.alpha, cfg.start_cap
  scale = float(np.nanmax(np.abs(g['em2d_x'])))
  curve = []
  for chunk in range(cfg.max_iters // cfg.num_iters):
    go = mesh.velocity_verlet(gx, gv, prev, cfg, cap, dt, alpha)
    wo = mesh_oracle.velocity_verlet(wx, wv, prev, ocfg, cap, dt, alpha)
    gx, gv = np.array(go[0]), np.array(go[1])
    wx, wv = wo[0], wo[1]
    # identical FIRE state: dt, alpha (1 ulp), n_pos, cap
    np.testing.assert_allclose([go[3], go[4], go[6]], [wo[3], wo[4], wo[6]], rtol=1e-6)
    assert go[5] == wo[5]
    dt, alpha, cap = wo[3], wo[4], wo[6]
    v_max = float(np.sqrt((wv ** 2).sum(axis=0)).max())
    if v_max < cfg.stop_v_max and cap < cfg.final_cap:
      cap = min(cap * cfg.cap_scale, cfg.final_cap)
    curve.append(float(np.nanmax(np.abs(gx - wx))) / scale)
  with capsys.disabled():
    print('em2d |dx| / scale per 20-step chunk:', ' '.join('%.1e' % c for c in curve))
  # Measured on MI355X (|dx| / 4.2 px per 20-step chunk): 3e-13 1e-11 2e-10 5e-8
  # 1.3e-3 2e-4 9e-4 8e-4 1.1e-3 ... 5e-4: round-off for 80 steps, then ONE burst
  # of x 3e4 within 20 steps -- the phase in which FIRE has grown dt by 1.1 per
  # step until the mesh overshoots (power < 0 resets it) -- and a plateau around
  # 1e-3 afterwards.  Both implementations take that branch at the same step.
  assert max(curve[:4]) <= 1e-6          # 80 steps: still at round-off level
  assert max(curve) <= 2.5e-3 and curve[-1] <= 1e-3


def test_relaxation_known_answers(gpu):
  """tests/mesh_test.py:25-65: FIRE and damped VV relax a perturbed mesh to 0;
  inputs are float64 NumPy arrays like in the reference test."""
  from sofima_amd import mesh
  for fire, gamma in ((True, 0.0), (False, 0.9 * np.sqrt(4 * 0.1))):
    x = np.zeros((2, 1, 50, 50))
    x[0, 0, 20:30, 10] = 3
    x[0, 0, 20:30, 40] = -4
    x[1, 0, 30, 10:20] = 2
    config = mesh.IntegrationConfig(
        dt=0.01, gamma=gamma, k0=0.1, k=0.1, stride=(10, 10), num_iters=100,
        max_iters=10000, stop_v_max=0.001, fire=fire)
    new_x, _, _ = mesh.relax_mesh(x, np.zeros_like(x), config)
    new_x = np.array(new_x)
    assert new_x.dtype == np.float32
    np.testing.assert_array_almost_equal(new_x, np.zeros_like(x), decimal=3)


def test_relax_large_mesh_vs_oracle(gpu):
  """cfg-2 sized mesh, production-like config, a short chunk vs the oracle."""
  from sofima_amd import mesh
  rng = np.random.default_rng(3)
  from scipy import ndimage
  prev = ndimage.gaussian_filter(rng.standard_normal((2, 1, 205, 205)),
                                 (0, 0, 6, 6)).astype(np.float32) * 40
  prev[:, 0, :2] = np.nan
  prev[:, 0, -2:] = np.nan
  prev[:, 0, :, :2] = np.nan
  prev[:, 0, :, -2:] = np.nan
  cfg = mesh.IntegrationConfig(
      dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40), num_iters=100,
      max_iters=100, stop_v_max=0.005, dt_max=1000, start_cap=0.01,
      final_cap=10, prefer_orig_order=True)
  x = np.zeros_like(prev)
  gx, ge, gt = mesh.relax_mesh(x, prev, cfg)
  wx, we, wt = mesh_oracle.relax_mesh(x, prev, cfg)
  assert gt == wt == 100
  # Un-converged snapshot with dt grown to ~0.7: round-off differences are
  # amplified by the dynamics (identical FIRE branch sequence, see the scalar
  # checks in test_velocity_verlet_matches_golden), so compare at the scale of
  # the displacement field.
  np.testing.assert_allclose(np.array(gx), wx, atol=2e-3 * np.abs(wx).max())
  np.testing.assert_allclose(ge, we, rtol=1e-3)


@pytest.mark.parametrize('source', ['analytic', 'bench'])
def test_headline_mesh_leg_followed_through_all_1000_steps(gpu, capsys, source):
  """(source 'bench': the LITERAL input of the bench's mesh leg -- the flow field
  flow_field() returns for bench.synth_pair(8192, 1002, warp), through bench.mesh_inputs;
  'analytic': a field of the same geometry written down directly.)

  The mesh leg bench.py times -- ONE chunk of 1000 FIRE steps of the
  [2, 1, 205, 205] mesh pulled to a flow field of the 8192^2 geometry (mesh.py:
  448-499) -- followed through its whole length.

  FIRE with dt_max = 1000 drives the time step to the stability limit of the
  mesh, where its fastest modes amplify ANY perturbation by orders of magnitude
  within a few steps until `power < 0` resets dt; the fixed point is attracting,
  so the trajectories re-converge.  The oracle shows this against ITSELF: start
  positions perturbed by 1e-7 of the scale differ by ~3e-3 of the scale around
  step 220, take other FIRE branches from there on, and end 2e-6 apart.  So
  "identical FIRE scalars at every checkpoint of the free run" is not a
  property the reference has against itself, and the test states what IS:

  (a) Window by window: the HIP state after k steps (a run of k steps of the
  same persistent kernel from the same start; k = 0, 20, ..., 980) is handed to
  the oracle, which continues the chunk for 20 steps (`resume`: same
  acceleration, same n_pos).  The HIP state after k + 20 steps has n_pos
  IDENTICAL and dt / alpha / cap equal to 1e-6 in every one of the 50 windows --
  the same FIRE branch at each of the 1000 steps when both start from the same
  state -- and positions within 10 x what the oracle's own continuation moves
  when its start state is perturbed by 1e-7 of the scale (+ 1e-5).
  (b) Free running (the oracle alone from the start): FIRE scalars identical
  while the difference is at round-off level (first 60 steps), the difference
  curve within 10 x the oracle's own sensitivity curve (+ 1e-5), and the END
  state -- what relax_mesh returns to the caller -- within SURVEY 8c's 1e-3 px."""
  import dataclasses
  import bench
  from sofima_amd import mesh
  rng = np.random.default_rng(1002)
  # a flow field like the warped pair's: content shift + a smooth 6 px
  # deformation + integer quantisation, NaN border from mesh_inputs
  n = (8192 - (bench.PATCH - bench.STEP)) // bench.STEP
  yy, xx = np.mgrid[:n, :n].astype(np.float32) * bench.STEP + bench.PATCH / 2
  d = bench.WARP[0] * np.sin(2 * np.pi * xx / bench.WARP[1]) * np.cos(2 * np.pi * yy / bench.WARP[1])
  flow = np.stack([np.rint(-5 - d), np.rint(3 + d)]).astype(np.float32)
  flow[:, rng.random((n, n)) < 0.003] = np.nan       # a few invalid vectors
  if source == 'bench':
    from sofima_amd import flow_field
    pre, post = bench.synth_pair(8192, 1002, warp=bench.WARP)
    flow = flow_field.JAXMaskedXCorrWithStatsCalculator().flow_field(
        pre, post, bench.PATCH, bench.STEP, batch_size=bench.BATCH)
    assert flow.shape == (4, n, n)
  prev = bench.mesh_inputs(flow, bench.PATCH // 2 // bench.STEP)
  assert prev.shape == (2, 1, 205, 205)
  total, win = bench.MESH_ITERS, 20
  cfg = mesh.IntegrationConfig(
      dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(bench.STEP, bench.STEP),
      num_iters=total, max_iters=total, stop_v_max=0.005,
      dt_max=1000, start_cap=0.01, final_cap=10, prefer_orig_order=True)
  ocfg = cfg_from(dataclasses.asdict(cfg))
  wcfg = cfg_from(dataclasses.asdict(dataclasses.replace(cfg, num_iters=win)))
  x0 = np.zeros_like(prev)
  v0 = np.zeros_like(prev)

  def hip_state(k):
    go = mesh.velocity_verlet(x0, v0, prev, dataclasses.replace(cfg, num_iters=k),
                              cfg.start_cap)
    return [np.array(go[0]), np.array(go[1]), np.array(go[2])] + list(go[3:])

  states = {k: hip_state(k) for k in range(win, total + 1, win)}
  scale = float(np.abs(states[total][0]).max())
  eps = 1e-7 * scale

  # (a) every 20-step window of the HIP chunk against the oracle's continuation
  worst = (0.0, 0.0, 0)
  resets = 0
  for k in range(0, total, win):
    if k == 0:
      start = (x0, v0, cfg.start_cap, None, None, None)
    else:
      sx, sv, sa, sdt, salpha, snpos, scap = states[k]
      start = (sx, sv, scap, sdt, salpha, (sa, snpos))
    jitter = (rng.standard_normal(x0.shape) * eps).astype(np.float32)
    wo = mesh_oracle.velocity_verlet(start[0], start[1], prev, wcfg, start[2], start[3],
                                     start[4], resume=start[5])
    wp = mesh_oracle.velocity_verlet(start[0] + jitter, start[1], prev, wcfg, start[2],
                                     start[3], start[4], resume=start[5])
    gx, _, _, gdt, galpha, gnpos, gcap = states[k + win]
    assert gnpos == wo[5], (k, gnpos, wo[5])
    np.testing.assert_allclose([gdt, galpha, gcap], [wo[3], wo[4], wo[6]], rtol=1e-6,
                               err_msg=f'window {k}')
    resets += gnpos < (0 if k == 0 else states[k][5]) + win
    err = float(np.abs(gx - wo[0]).max()) / scale
    sens = float(np.abs(wp[0] - wo[0]).max()) / scale
    assert err <= 10 * sens + 1e-5, (k, err, sens)
    if err > worst[0]:
      worst = (err, sens, k)
  assert resets >= 3                         # the chunk does contain FIRE resets

  # (b) the free-running oracle, and the oracle against itself
  def free_run(start):
    snaps = []
    wo = mesh_oracle.velocity_verlet(start, v0, prev, ocfg, cfg.start_cap,
                                     snapshots=snaps, snapshot_every=win)
    return snaps + [(total, wo[0], wo[3], wo[4], wo[5], wo[6])]

  ora = free_run(x0)
  pert = free_run((rng.standard_normal(x0.shape) * eps).astype(np.float32))
  curve, own = [], []
  for so, sp in zip(ora, pert):
    k = so[0]
    g = states[k]
    if k <= 60:
      assert g[5] == so[4] and np.allclose([g[3], g[4], g[6]], [so[2], so[3], so[5]],
                                           rtol=1e-6), k
    curve.append(float(np.abs(g[0] - so[1]).max()) / scale)
    own.append(float(np.abs(sp[1] - so[1]).max()) / scale)
  with capsys.disabled():
    print(f'headline mesh: worst window {worst[2]}: {worst[0]:.1e} of the scale (oracle vs '
          f'its perturbed self there: {worst[1]:.1e}); free run |dx| / scale at every 100th '
          'step, HIP vs oracle:', ' '.join('%.1e' % c for c in curve[4::5]),
          '| oracle vs perturbed oracle:', ' '.join('%.1e' % c for c in own[4::5]))
  assert max(curve[:3]) <= 1e-6              # 60 steps: round-off level
  assert max(curve) <= 10 * max(own) + 1e-5, (max(curve), max(own))
  # the end state is what relax_mesh (the bench call) returns: SURVEY 8c, 1e-3 px
  gx, _, gt = mesh.relax_mesh(x0, prev, cfg)
  assert gt == total
  np.testing.assert_array_equal(np.array(gx), states[total][0])
  np.testing.assert_allclose(np.array(gx), ora[-1][1], atol=1e-3)


def test_error_behaviour(gpu):
  from sofima_amd import mesh
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
  with pytest.raises(ValueError):
    mesh.elastic_mesh_3d(np.zeros((3, 4, 4, 4)), 0.1, 10.0,
                         links=((2, 0, 0),))


def _with_env(env, fn):
  import os
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


@pytest.mark.gpu
@pytest.mark.parametrize('variant', ['fire', 'fire_drift', 'verlet', 'no_prev',
                                     'wide'])
def test_integrator_paths_agree(gpu, variant):
  """Persistent, LDS-tiled and multi-launch integrators give the same chunk.

  The three differ only in the order the per-step partial sums (FIRE power,
  drift means) are added, so the trajectories agree to round-off; each is also
  checked against the oracle.
  """
  from sofima_amd import mesh
  rng = np.random.default_rng(5)
  shape = (2, 3, 40, 70) if variant != 'wide' else (2, 1, 37, 300)
  from scipy import ndimage
  prev = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 3, 3))
  prev = (prev * 30).astype(np.float32)
  prev[:, :, :2] = np.nan
  prev[:, :, :, -3:] = np.nan
  kw = dict(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40), num_iters=40,
            max_iters=40, stop_v_max=1e-9, dt_max=1000, start_cap=0.01,
            final_cap=10, prefer_orig_order=True)
  if variant == 'fire_drift':
    kw['remove_drift'] = True
  if variant == 'verlet':
    kw.update(fire=False, gamma=0.5, dt=0.05, start_cap=10.0)
  cfg = mesh.IntegrationConfig(**kw)
  x0 = (rng.standard_normal(shape) * 0.5).astype(np.float32)
  pv = None if variant == 'no_prev' else prev
  run = lambda: mesh.relax_mesh(x0.copy(), None if pv is None else pv.copy(), cfg)
  res = {
      'default': run(),
      'tiled': _with_env({'SFM_MESH_PERSISTENT': '0'}, run),
      'multi': _with_env({'SFM_MESH_PERSISTENT': '0', 'SFM_MESH_TILED': '0'}, run),
  }
  wx, we, wt = mesh_oracle.relax_mesh(x0.copy(), None if pv is None else pv.copy(),
                                      cfg)
  scale = np.abs(wx).max()
  for name, (gx, ge, gt) in res.items():
    assert gt == wt, name
    np.testing.assert_allclose(np.array(gx), wx, atol=1e-3 * scale, err_msg=name)
    np.testing.assert_allclose(ge, we, rtol=1e-3, err_msg=name)
  for name in ('default', 'tiled'):
    np.testing.assert_allclose(np.array(res[name][0]), np.array(res['multi'][0]),
                               atol=2e-4 * scale, err_msg=name)


@pytest.mark.gpu
def test_montage_relaxation_tiled_equals_multi_launch(gpu, golden):
  """prev_fn path: advance + target mesh + tiled integrate == multi-launch."""
  import json
  from sofima_amd import mesh, stitch_elastic
  from tests.util import cfg_from
  g = golden('montage')
  cfg = cfg_from(json.loads(str(g['cfg'])), mesh.IntegrationConfig)
  fn = stitch_elastic.TargetMeshFn(g['nbors'], g['fx'], g['fy'], tuple(g['stride']))
  run = lambda: mesh.relax_mesh(g['x'], None, cfg, prev_fn=fn)
  a = run()
  b = _with_env({'SFM_MESH_TILED': '0'}, run)
  assert a[2] == b[2] == int(g['t'])
  np.testing.assert_allclose(np.array(a[0]), np.array(b[0]), atol=2e-3)
  np.testing.assert_allclose(np.array(a[0]), g['relaxed'], atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(3, 5, 6, 7, 9), (3, 1, 6, 7, 9), (3, 6, 7, 9)])
def test_remove_drift_axes_quirk_3d(gpu, shape):
  """remove_drift takes means over axes (1, 2, 3) (mesh.py:496-497): global for
  4-D states, per x column for 5-D ones (also with a batch of one)."""
  from sofima_amd import mesh
  rng = np.random.default_rng(9)
  x0 = (rng.standard_normal(shape) * 2).astype(np.float32)
  prev = (rng.standard_normal(shape) * 4).astype(np.float32)
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.05, k=0.1,
                               stride=(10, 10, 10), num_iters=25, max_iters=25,
                               stop_v_max=1e-9, dt_max=100, start_cap=1.0,
                               final_cap=10.0, remove_drift=True)
  gx, ge, gt = mesh.relax_mesh(x0.copy(), prev.copy(), cfg,
                               mesh_force=mesh.elastic_mesh_3d)
  wx, we, wt = mesh_oracle.relax_mesh(x0.copy(), prev.copy(), cfg,
                                      mesh_force=mesh_oracle.elastic_mesh_3d)
  assert gt == wt == 25
  np.testing.assert_allclose(np.array(gx), wx, atol=1e-3 * np.abs(wx).max())
  np.testing.assert_allclose(ge, we, rtol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['regularized', 'regular', 'prep_failed', 'masked'])
def test_three_pass_relaxation_driver_vs_oracle(gpu, case):
  """processor_mesh.relax_mesh (RelaxMesh.relax_mesh, processor/mesh.py:428-513):
  relax, fold test, soft relaxation towards the first solution, fold test,
  final relaxation -- the same status, step count and mesh as the oracle."""
  import types
  from scipy import ndimage
  from sofima_amd import mesh, processor_mesh
  rng = np.random.default_rng(8)
  shape = (2, 1, 40, 44)
  prev = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 4, 4)) * 60
  if case in ('regularized', 'masked'):
    prev[0, 0, 18:22, 20:24] += 60          # a local fold in the flow field
  frac = 0.7
  if case == 'prep_failed':
    # a 100 px tear: the free band of the soft pass is stretched beyond 1.1
    prev = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 4, 4)) * 20
    prev[0, 0, :, 22:] += 100
    frac = 0.9
  prev = prev.astype(np.float32)
  kw = dict(dt=0.001, gamma=0.0, k0=0.3, k=0.1, stride=(40, 40), num_iters=100,
            max_iters=400, stop_v_max=0.005, dt_max=1000, start_cap=1e6,
            final_cap=1e6, prefer_orig_order=True)
  cfg = mesh.IntegrationConfig(**kw)
  ocfg = types.SimpleNamespace(**cfg.to_dict())
  mask = None
  if case == 'masked':
    mask = np.zeros((1, 40, 44), bool)
    mask[0, :3, :] = True
  x0 = np.zeros(shape, np.float32)
  gx, ge, gs, gstat = processor_mesh.relax_mesh(x0.copy(), prev, cfg, mask, frac)
  wx, we, ws, wstat = mesh_oracle.relax_mesh_passes(x0.copy(), prev, ocfg, mask, frac)
  want = {'regularized': 2, 'regular': 0, 'prep_failed': 1, 'masked': 2}[case]
  assert int(gstat) == wstat == want, (gstat, wstat)
  assert gs == ws
  np.testing.assert_array_equal(np.isnan(gx), np.isnan(wx))
  scale = np.nanmax(np.abs(wx))
  np.testing.assert_allclose(np.nan_to_num(gx), np.nan_to_num(wx), atol=2e-3 * scale)
  np.testing.assert_allclose(ge, we, rtol=1e-3, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['regularized', 'regular', 'prep_failed', 'masked', 'median'])
def test_three_pass_relaxation_driver_vs_reference_output(gpu, golden, case):
  """processor_mesh.relax_mesh == what RelaxMesh.relax_mesh of the reference
  returned for the same inputs (relax_passes.npz, generated by running
  processor/mesh.py:428-513 through the stand-in): status, steps, NaN pattern,
  mesh, energies."""
  import json
  from sofima_amd import mesh, processor_mesh
  g = golden('relax_passes')
  cfg = mesh.IntegrationConfig(**json.loads(str(g['cfg'])))
  prev = g[f'{case}_prev']
  mask = g[f'{case}_mask']
  mask = None if mask.size == 0 else mask
  init = (processor_mesh.MeshInitState.PREV_MEDIAN if bool(g[f'{case}_median'])
          else processor_mesh.MeshInitState.ZEROS)
  gx, ge, gs, gstat = processor_mesh.relax_mesh(
      np.zeros_like(prev), prev.copy(), cfg, mask, float(g[f'{case}_frac']), init)
  assert int(gstat) == int(g[f'{case}_status']) and gs == int(g[f'{case}_steps'])
  want = g[f'{case}_x']
  np.testing.assert_array_equal(np.isnan(gx), np.isnan(want))
  scale = np.nanmax(np.abs(want))
  np.testing.assert_allclose(np.nan_to_num(gx), np.nan_to_num(want), atol=2e-3 * scale)
  np.testing.assert_allclose(ge, g[f'{case}_ekin'], rtol=1e-3, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 1, 205, 205), (2, 3, 40, 70), (2, 1, 33, 300)])
def test_speculative_fire_is_bit_identical(gpu, shape):
  """mesh_persist2d_spec_kernel runs every step on the downhill branch and
  redoes it when the power turns out negative: bit-identical to the kernel
  that waits for the power, including chunks with many uphill events."""
  from scipy import ndimage
  from sofima_amd import mesh
  rng = np.random.default_rng(21)
  prev = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 3, 3)) * 40
  prev = (prev + rng.standard_normal(shape) * 2).astype(np.float32)
  prev[:, :, :2] = np.nan
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.05, k=0.1, stride=(40, 40),
                               num_iters=150, max_iters=450, stop_v_max=1e-9, dt_max=1000,
                               start_cap=0.01, final_cap=10, prefer_orig_order=True)
  x0 = np.zeros(shape, np.float32)
  vv = lambda: mesh.velocity_verlet(x0, np.zeros_like(x0), prev, cfg, cfg.start_cap)
  a = _with_env({'SFM_MESH_SPECULATE': '1'}, vv)
  b = _with_env({'SFM_MESH_SPECULATE': '0'}, vv)
  for u, w in zip(a[:3], b[:3]):
    np.testing.assert_array_equal(np.array(u), np.array(w))
  assert a[3:] == b[3:]
  assert a[5] < 150          # the power went negative at least once in the chunk
  run = lambda: mesh.relax_mesh(x0, prev, cfg)
  c = _with_env({'SFM_MESH_SPECULATE': '1'}, run)
  d = _with_env({'SFM_MESH_SPECULATE': '0'}, run)
  np.testing.assert_array_equal(np.array(c[0]), np.array(d[0]))
  assert c[1] == d[1] and c[2] == d[2]
  wx, we, wt = mesh_oracle.relax_mesh(x0, prev, cfg)
  assert c[2] == wt
  np.testing.assert_allclose(np.array(c[0]), wx, atol=2e-3 * np.abs(wx).max())


@pytest.mark.gpu
def test_speculative_hand_off_fuzz(gpu):
  """Time-boxed (25 s) random fuzz of the persistent kernel's hand-off -- candidate
  positions instead of (x, v, a), uphill candidates only on a redo, 16-byte pair
  loads -- against the kernel that waits for the power: random shapes (one to four
  slices, ragged tiles), target fields from smooth to noise, spring constants and
  time steps that make uphill events from rare to every few steps, several chunks;
  every third case with another stream keeping the memory system busy (uneven
  load: the hand-off must not depend on arrival order).  Positions, velocities,
  FIRE scalars and step counts equal bit for bit."""
  import time
  import torch
  from scipy import ndimage
  from sofima_amd import mesh
  import os
  # (SFM_FUZZ_SECONDS / SFM_FUZZ_SEED: longer soaks, tools/measure/profile_round5.sh)
  box = float(os.environ.get('SFM_FUZZ_SECONDS', '25'))
  rng = np.random.default_rng(int(os.environ.get('SFM_FUZZ_SEED', '5')))
  t0 = time.time()
  cases = uphill_cases = 0
  junk = torch.zeros(64 << 20, device='cuda')
  side = torch.cuda.Stream()
  while time.time() - t0 < box or cases < 6:
    ny, nx = (int(v) for v in rng.integers(17, 211, 2))
    nz = int(rng.integers(1, 5))
    if ((ny + 15) // 16) * ((nx + 15) // 16) * nz > 256:
      nz = 1
    shape = (2, nz, ny, nx)
    sig = float(rng.choice([0.0, 1.0, 3.0, 8.0]))
    prev = rng.standard_normal(shape)
    if sig:
      prev = ndimage.gaussian_filter(prev, (0, 0, sig, sig))
    prev = (prev / np.abs(prev).max() * float(rng.choice([2, 20, 80]))).astype(np.float32)
    if rng.random() < 0.5:
      prev[:, :, : int(rng.integers(1, 4))] = np.nan
    iters = int(rng.integers(20, 200))
    cfg = mesh.IntegrationConfig(
        dt=float(rng.choice([0.001, 0.01, 0.1])), gamma=0.0, k0=float(rng.choice([0.01, 0.05, 0.3])),
        k=float(rng.choice([0.05, 0.1, 0.5])), stride=(40, 40), num_iters=iters,
        max_iters=iters * int(rng.integers(1, 4)), stop_v_max=1e-9,
        dt_max=float(rng.choice([10, 1000])), start_cap=float(rng.choice([0.01, 1.0])),
        final_cap=10, prefer_orig_order=bool(rng.integers(0, 2)))
    x0 = (rng.standard_normal(shape) * float(rng.choice([0, 0.5]))).astype(np.float32)
    vv = lambda: mesh.velocity_verlet(x0, np.zeros_like(x0), prev, cfg, cfg.start_cap)
    busy = cases % 3 == 2
    if busy:
      with torch.cuda.stream(side):
        for _ in range(40):
          junk.add_(1.0)
    a = _with_env({'SFM_MESH_SPECULATE': '1'}, vv)
    a2 = _with_env({'SFM_MESH_SPECULATE': '1'}, lambda: mesh.relax_mesh(x0, prev, cfg))
    torch.cuda.synchronize()
    b = _with_env({'SFM_MESH_SPECULATE': '0'}, vv)
    b2 = _with_env({'SFM_MESH_SPECULATE': '0'}, lambda: mesh.relax_mesh(x0, prev, cfg))
    msg = f'case {cases}: {shape} {cfg}'
    for u, w in zip(a[:3], b[:3]):
      np.testing.assert_array_equal(np.array(u), np.array(w), err_msg=msg)
    assert a[3:] == b[3:], msg
    np.testing.assert_array_equal(np.array(a2[0]), np.array(b2[0]), err_msg=msg)
    np.testing.assert_array_equal(np.array(a2[1]), np.array(b2[1]), err_msg=msg)   # (NaN == NaN: blown-up cases)
    assert a2[2] == b2[2], msg
    uphill_cases += a[5] < iters   # n_pos below the step count: the power went negative
    cases += 1
  assert uphill_cases >= 3, (cases, uphill_cases)
  print(f'hand-off fuzz: {cases} cases, {uphill_cases} with uphill events, 0 mismatches')


@pytest.mark.parametrize('case', ['tile2d', 'tile3d', 'vol3d', 'plane_vv'])
def test_small_mesh_single_launch_is_bit_identical(gpu, case):
  """mesh_small_kernel (meshes of at most one workgroup's worth of nodes: all
  steps of a chunk in one launch) == the kernel-per-phase path, bit for bit."""
  from sofima_amd import mesh, stitch_rigid
  rng = np.random.default_rng(33)
  if case in ('tile2d', 'tile3d'):
    nc = 2 if case == 'tile2d' else 3
    shape = (nc, 1, 5, 7)
    cx = (rng.standard_normal(shape) * 30).astype(np.float32)
    cy = (rng.standard_normal(shape) * 30).astype(np.float32)
    cx[0] -= 400
    cy[1] -= 400
    cx[:, 0, 2, 3] = np.nan          # a missing pair
    cx[:, :, :, -1] = np.inf
    cy[:, :, -1, :] = np.inf
    cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.0, k=0.1, stride=(1, 1),
                                 num_iters=300, max_iters=1500, stop_v_max=0.001, dt_max=100)
    fn = stitch_rigid.elastic_tile_mesh if nc == 2 else stitch_rigid.elastic_tile_mesh_3d
    run = lambda: stitch_rigid.optimize_coarse_mesh(cx, cy, cfg, mesh_fn=fn)
    a = _with_env({'SFM_MESH_SMALL': '1'}, run)
    b = _with_env({'SFM_MESH_SMALL': '0'}, run)
    np.testing.assert_array_equal(a, b)
    assert np.isfinite(a).all() and np.abs(a).max() > 100
    return
  if case == 'vol3d':
    shape = (3, 1, 6, 6, 6)
    stride = (40.0, 40.0, 40.0)
    kw = dict(mesh_force=mesh.elastic_mesh_3d)
    cfg_kw = dict(remove_drift=True, prefer_orig_order=False)
  else:
    shape = (2, 1, 9, 13)
    stride = (40.0, 40.0)
    kw = {}
    cfg_kw = dict(fire=False, prefer_orig_order=True)
    # the in-plane spring mesh would take the persistent kernel: switch it off too
  prev = (rng.standard_normal(shape) * 6).astype(np.float32)
  prev.reshape(shape[0], -1)[:, :3] = np.nan
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.5 if case == 'plane_vv' else 0.0, k0=0.05,
                               k=0.1, stride=stride, num_iters=120, max_iters=360,
                               stop_v_max=1e-9, dt_max=100,
                               start_cap=10 if case == 'plane_vv' else 0.05, final_cap=10,
                               **cfg_kw)
  x0 = np.zeros(shape, np.float32)
  vv = lambda: mesh.velocity_verlet(x0, np.zeros_like(x0), prev, cfg, cfg.start_cap, **kw)
  a = _with_env({'SFM_MESH_SMALL': '1', 'SFM_MESH_PERSISTENT': '0'}, vv)
  b = _with_env({'SFM_MESH_SMALL': '0', 'SFM_MESH_PERSISTENT': '0'}, vv)
  for u, w in zip(a[:3], b[:3]):
    np.testing.assert_array_equal(np.array(u), np.array(w))
  assert a[3:] == b[3:]
  run = lambda: mesh.relax_mesh(x0, prev, cfg, **kw)
  c = _with_env({'SFM_MESH_SMALL': '1', 'SFM_MESH_PERSISTENT': '0'}, run)
  d = _with_env({'SFM_MESH_SMALL': '0', 'SFM_MESH_PERSISTENT': '0'}, run)
  np.testing.assert_array_equal(np.array(c[0]), np.array(d[0]))
  assert c[1] == d[1] and c[2] == d[2]


@pytest.mark.parametrize('shape', [(2, 2, 17, 40), (2, 1, 33, 62), (2, 3, 16, 63), (2, 1, 5, 124),
                                   (2, 2, 31, 125), (2, 1, 100, 311)])
def test_shared_spring_step_is_bit_identical(gpu, shape):
  """integrate_shared2d_kernel (every spring once, far-side terms through DPP
  wave shifts) vs the per-node kernel in which both ends evaluate every spring: the
  forces are the same bit for bit, so without drift removal (FIRE depends on the
  SIGN of the power only) the whole chunk is bit-identical for tile widths that
  are not multiples of 62 / heights that are not multiples of 16; damped Verlet
  and the in-place (prev_fn-style) form are covered by the montage test."""
  from scipy import ndimage
  from sofima_amd import mesh
  rng = np.random.default_rng(shape[-1])
  prev = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 2, 2)) * 50
  prev = (prev + rng.standard_normal(shape)).astype(np.float32)
  prev[:, :, :2, :3] = np.nan
  prev[:, :, -1, -2:] = np.nan
  x0 = (rng.standard_normal(shape) * 0.5).astype(np.float32)
  for kw in (dict(), dict(fire=False, gamma=0.5, dt=0.05, start_cap=10.0)):
    base = dict(dt=0.001, gamma=0.0, k0=0.05, k=0.1, stride=(40, 40), num_iters=50, max_iters=100,
                stop_v_max=1e-9, dt_max=100, start_cap=0.05, final_cap=10,
                prefer_orig_order=True)
    base.update(kw)
    cfg = mesh.IntegrationConfig(**base)
    vv = lambda: mesh.velocity_verlet(x0, np.zeros_like(x0), prev, cfg, cfg.start_cap)
    env = {'SFM_MESH_PERSISTENT': '0'}
    a = _with_env(env, vv)
    c = _with_env(dict(env, SFM_MESH_TILED='0'), vv)
    for u, m in zip(a[:3], c[:3]):
      np.testing.assert_array_equal(np.array(u), np.array(m))
    assert a[3:] == c[3:]


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 2, 75, 70), (2, 3, 204, 204), (2, 1, 50, 92),
                                   (2, 2, 100, 64), (2, 1, 33, 187), (2, 2, 40, 93)])
@pytest.mark.parametrize('drift', [False, True])
def test_packed_last_tile_column(gpu, shape, drift):
  """Tiled in-plane step with the tile rows of a narrow last tile column packed
  side by side into one workgroup (X % 62 of 8, 18, 30, 2, 1 columns; 31: not
  packed) vs a workgroup per tile (SFM_MESH_PACK=0).  Every node takes the same
  operations; only the grouping of the per-workgroup partial sums changes, so
  without drift removal (FIRE reads the SIGN of the power only) the chunk is bit
  for bit the same, with drift removal it agrees to round-off; both follow the
  oracle."""
  from oracle import mesh_oracle
  from scipy import ndimage
  from sofima_amd import _abi, mesh
  rng = np.random.default_rng(shape[-1] + shape[-2])
  prev = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 2, 2)) * (8 if drift else 40)
  prev = (prev + 0.3 * rng.standard_normal(shape)).astype(np.float32)
  prev[:, :, :2, -3:] = np.nan     # holes in the packed column
  prev[:, :, -1, -1] = np.nan
  x0 = (rng.standard_normal(shape) * 0.3).astype(np.float32)
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.05, k=0.1, stride=(40, 40),
                               num_iters=40, max_iters=120, stop_v_max=1e-9, dt_max=100,
                               start_cap=0.05, final_cap=10, prefer_orig_order=True,
                               remove_drift=drift)
  with _abi.option('SFM_MESH_PERSISTENT', 0):
    a = mesh.relax_mesh(x0, prev, cfg)
    with _abi.option('SFM_MESH_PACK', 0):
      b = mesh.relax_mesh(x0, prev, cfg)
  want = mesh_oracle.relax_mesh(x0, prev, cfg)
  assert a[2] == b[2] == want[2]
  scale = np.abs(want[0]).max()
  if not drift:
    np.testing.assert_array_equal(np.array(a[0]), np.array(b[0]))
    assert list(a[1]) == list(b[1])
  else:
    # (the last-bit difference of the drift means is amplified by the relaxation like
    # any other change of the summation order -- test_banded_c_loop_fused_kernel; on
    # the 204^2 case ONE node of 250 k reaches 5e-4 of the scale after 120 steps)
    diff = np.abs(np.array(a[0]) - np.array(b[0]))
    assert np.nanmax(diff) <= 1e-3 * scale
    assert np.mean(diff > 2e-4 * scale) < 1e-4
  np.testing.assert_allclose(np.array(a[0]), want[0], atol=1e-3 * scale)
  np.testing.assert_allclose(a[1], want[1], rtol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('drift', [False, True])
def test_xcd_tile_order_is_bit_identical(gpu, drift):
  """SFM_MESH_XCD (every XCD takes one contiguous run of tiles of the fused
  in-plane step; default from 2048 tiles on) only changes which workgroup owns
  which tile: the per-tile partial sums are reduced in tile order either way."""
  from sofima_amd import _abi, mesh
  rng = np.random.default_rng(41)
  shape = (2, 3, 150, 333)     # 3 x 10 x 6 = 180 tiles, ragged right / bottom edges
  prev = (rng.standard_normal(shape) * 2).astype(np.float32)
  x0 = np.zeros(shape, np.float32)
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(20., 20.),
                               num_iters=60, max_iters=120, stop_v_max=1e-9, dt_max=1000,
                               start_cap=0.01, final_cap=10, prefer_orig_order=True,
                               remove_drift=drift)
  out = []
  for opt in (0, 1):
    with _abi.option('SFM_MESH_XCD', opt):
      x, e, t = mesh.relax_mesh(x0, prev, cfg)
    out.append((np.array(x), list(e), t))
  np.testing.assert_array_equal(out[0][0], out[1][0])
  assert out[0][1] == out[1][1] and out[0][2] == out[1][2]


_MARCH_SHAPES = [(3, 1, 5, 7, 9), (3, 2, 40, 33, 70), (3, 17, 23, 100), (3, 1, 1, 30, 300),
                 (3, 3, 30, 1, 40), (3, 2, 64, 64, 1), (3, 1, 9, 130, 61)]


@pytest.mark.gpu
@pytest.mark.parametrize('shape', _MARCH_SHAPES)
@pytest.mark.parametrize('prefer', [False, True])
def test_march3d_every_spring_once_is_bit_identical(gpu, shape, prefer):
  """integrate_march3d_kernel (a workgroup marches a column of the volume along z;
  every spring of the 13 default links is evaluated by ONE of its ends and read from
  LDS by the other) against integrate_kernel<3> (every node evaluates its 26 springs).
  A spring seen from its two ends is the same float expression and the terms are
  added in the reference's link order (mesh.py:271-277), so the damped-Verlet chunk --
  positions, velocities AND forces after 7 steps -- is bit-identical for every
  workgroup size and plane run, with NaN targets, with and without the prev pull, on
  meshes with an axis of length 1 and meshes wider than a tile."""
  from sofima_amd import _abi, mesh
  rng = np.random.default_rng(3)
  x0 = (rng.standard_normal(shape) * 2).astype(np.float32)
  v0 = (rng.standard_normal(shape) * 0.1).astype(np.float32)
  for has_prev in (True, False):
    prev = None
    if has_prev:
      prev = (rng.standard_normal(shape) * 4).astype(np.float32)
      prev[..., :1] = np.nan
    cfg = mesh.IntegrationConfig(dt=0.05, gamma=0.5, k0=0.05, k=0.1, stride=(10, 10, 10),
                                 num_iters=7, max_iters=7, stop_v_max=1e-9, dt_max=100,
                                 start_cap=10.0, final_cap=10.0, fire=False,
                                 prefer_orig_order=prefer)
    run = lambda: [np.array(t) for t in mesh.velocity_verlet(
        x0, v0, prev, cfg, cfg.start_cap, mesh_force=mesh.elastic_mesh_3d)]
    with _abi.option('SFM_MESH_MARCH3D', 0):
      want = run()
    for t, zc in ((None, None), (256, 2), (512, 3), (1024, None)):
      with _abi.option('SFM_MESH_MARCH3D', 1), _abi.option('SFM_MESH_MARCH3D_T', t), \
          _abi.option('SFM_MESH_MARCH3D_ZC', zc):
        got = run()
      for name, w, g in zip('xva', want, got):
        np.testing.assert_array_equal(w, g, err_msg=f'{name} T={t} run={zc} prev={has_prev}')


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(3, 2, 40, 33, 70), (3, 1, 9, 130, 61), (3, 17, 23, 100)])
def test_march3d_fire_vs_oracle(gpu, shape):
  """With FIRE the z-march kernel adds the power / drift sums in another order (a
  thread's column, then the workgroup tree): the relaxation agrees with the per-node
  kernel to round-off and with the oracle at the tolerance of the other 3-D tests
  (mesh.py:192-279, 371-521)."""
  from sofima_amd import _abi, mesh
  rng = np.random.default_rng(11)
  x0 = (rng.standard_normal(shape) * 2).astype(np.float32)
  prev = (rng.standard_normal(shape) * 4).astype(np.float32)
  prev[..., :2] = np.nan
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.05, k=0.1, stride=(10, 10, 10),
                               num_iters=30, max_iters=30, stop_v_max=1e-9, dt_max=100,
                               start_cap=1.0, final_cap=10.0, remove_drift=True)
  run = lambda: mesh.relax_mesh(x0.copy(), prev.copy(), cfg, mesh_force=mesh.elastic_mesh_3d)
  with _abi.option('SFM_MESH_MARCH3D', 0):
    a = run()
  with _abi.option('SFM_MESH_MARCH3D', 1):
    b = run()
  wx, we, wt = mesh_oracle.relax_mesh(x0.copy(), prev.copy(), cfg,
                                      mesh_force=mesh_oracle.elastic_mesh_3d)
  scale = np.abs(wx).max()
  assert a[2] == b[2] == wt
  np.testing.assert_allclose(np.array(b[0]), np.array(a[0]), atol=2e-4 * scale)
  np.testing.assert_allclose(np.array(b[0]), wx, atol=1e-3 * scale)
  np.testing.assert_allclose(b[1], we, rtol=1e-3)


@pytest.mark.gpu
def test_march3d_is_the_default_for_large_volumes(gpu):
  """From 1.5 * 10^6 nodes on a default-link volume takes the z-march kernel without any
  switch (the kernel trace under profiles/ names it): same damped-Verlet chunk as
  with SFM_MESH_MARCH3D=0."""
  import torch
  from sofima_amd import _abi, mesh
  shape = (3, 2, 100, 100, 100)
  rng = np.random.default_rng(2)
  dev = torch.device('cuda:0')
  x0 = torch.from_numpy((rng.standard_normal(shape) * 2).astype(np.float32)).to(dev)
  prev = torch.from_numpy((rng.standard_normal(shape) * 4).astype(np.float32)).to(dev)
  cfg = mesh.IntegrationConfig(dt=0.05, gamma=0.5, k0=0.05, k=0.1, stride=(10, 10, 10),
                               num_iters=5, max_iters=5, stop_v_max=1e-9, dt_max=100,
                               start_cap=10.0, final_cap=10.0, fire=False)
  run = lambda: [np.array(t) for t in mesh.velocity_verlet(
      x0, torch.zeros_like(x0), prev, cfg, cfg.start_cap, mesh_force=mesh.elastic_mesh_3d)]
  got = run()
  with _abi.option('SFM_MESH_MARCH3D', 0):
    want = run()
  for w, g in zip(want, got):
    np.testing.assert_array_equal(w, g)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(3, 2, 9, 33, 17), (3, 1, 1, 12, 40), (3, 1, 6, 1, 7)])
def test_forces_3d_with_nan_and_inf_positions_vs_oracle(gpu, shape):
  """elastic_mesh_3d on positions holding NaN and +-inf, also on the faces, edges and
  corners of the volume (mesh.py:192-279: every spring component passes through
  nan_to_num(posinf=0, neginf=0)).  The HIP stencil evaluates a missing neighbour
  against the node itself and adds the resulting +-0 without a select: a non-finite
  position there must still give what the reference's masked 0 gives.  Both kernels
  (per node / z-march through one damped-Verlet step) against the oracle."""
  from sofima_amd import _abi, mesh
  rng = np.random.default_rng(13)
  x = (rng.standard_normal(shape) * 3).astype(np.float32)
  flat = x.reshape(3, -1)
  n = flat.shape[1]
  for k, bad in enumerate((np.nan, np.inf, -np.inf)):
    idx = rng.integers(0, n, max(2, n // 40))
    flat[rng.integers(0, 3, idx.size), idx] = bad
  # corners and face centres of the first volume
  vol = x.reshape((3, -1) + shape[-3:])
  vol[0, 0, 0, 0, 0] = np.nan
  vol[1, 0, -1, -1, -1] = np.inf
  vol[2, 0, 0, -1, 0] = -np.inf
  for poo in (False, True):
    got = np.array(mesh.elastic_mesh_3d(x, 0.1, (40.0, 40.0, 30.0), poo))
    want = mesh_oracle.elastic_mesh_3d(x, 0.1, (40.0, 40.0, 30.0), poo)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got == 0, want == 0)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6)
  # the z-march kernel adds the same floats: one Verlet step from these positions
  cfg = mesh.IntegrationConfig(dt=0.05, gamma=0.5, k0=0.05, k=0.1, stride=(40.0, 40.0, 30.0),
                               num_iters=1, max_iters=1, stop_v_max=1e-9, dt_max=100,
                               start_cap=10.0, final_cap=10.0, fire=False)
  run = lambda: [np.array(t) for t in mesh.velocity_verlet(
      x, np.zeros_like(x), None, cfg, cfg.start_cap, mesh_force=mesh.elastic_mesh_3d)]
  with _abi.option('SFM_MESH_MARCH3D', 0):
    a = run()
  with _abi.option('SFM_MESH_MARCH3D', 1):
    b = run()
  for u, w in zip(a, b):
    np.testing.assert_array_equal(u, w)
