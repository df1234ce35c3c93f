"""CPU oracle for the elastic spring-mesh relaxer.  TEST INFRASTRUCTURE.

A float32 NumPy restatement of the algorithm of the reference's `mesh.py`
(/root/reference/mesh.py), written from its observable behaviour; the checker
for the HIP path and the `cpu_baseline` leg of bench.py.  Never imported by
anything under `sofima_amd/`.

Parity pinning: the reference's known-answer tests (tests/mesh_test.py:25-144,
re-typed in tests/test_reference_kats.py) and golden vectors produced in the
build container by executing the unmodified reference source over a NumPy
stand-in for jax (tests/golden/_refshim; "reference over a stand-in", not XLA).

Conventions (mesh.py:16-27, 400-408): x is [C, ..., z, y, x] float32 of
RELATIVE node offsets, C = 2 (in-plane, components x,y) or 3 (x,y,z); unit
masses, so force == acceleration.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32

# xyz link directions of the 26-neighbourhood, one per opposite pair
# (mesh.py:172-189).
LINKS_3D = (
    (1, 0, 0), (0, 1, 0), (0, 0, 1),
    (1, 1, 0), (-1, 1, 0), (1, 0, 1), (-1, 0, 1), (0, 1, 1), (0, -1, 1),
    (1, 1, 1), (1, 1, -1), (1, -1, 1), (-1, 1, 1),
)
# In-plane families (mesh.py:106-167): -, |, \, /
LINKS_2D = ((1, 0), (0, 1), (1, 1), (-1, 1))


def _shift_views(nsp, direction):
  """Slices selecting the 'far' and 'near' end of every spring of a family.

  direction is in xy[z] order; array axes are [z]yx.  A spring joins node n
  (near) to node n + direction (far) -- for a negative component the roles of
  the two slices along that axis swap (mesh.py:230-245).
  """
  far, near = [], []
  for d in direction[::-1][-nsp:]:
    if d == 1:
      far.append(slice(1, None)); near.append(slice(None, -1))
    elif d == -1:
      far.append(slice(None, -1)); near.append(slice(1, None))
    elif d == 0:
      far.append(slice(None)); near.append(slice(None))
    else:
      raise ValueError('Only |v| <= 1 values supported within links.')
  return tuple(far), tuple(near)


def _hooke(d, length, l0, k_eff, direction, prefer_orig_order):
  with np.errstate(divide='ignore', invalid='ignore'):
    if prefer_orig_order:
      fac = np.ones_like(d)
      for i, s in enumerate(direction):
        if s != 0:
          fac[i] = f32(s) * np.sign(d[i])
      f = f32(-k_eff) * (f32(1.0) - f32(l0) * fac / length) * d
    else:
      f = f32(-k_eff) * (f32(1.0) - f32(l0) / length) * d
  return np.nan_to_num(f, nan=0.0, posinf=0.0, neginf=0.0).astype(f32)


def inplane_force(x, k, stride, prefer_orig_order=False):
  """8-neighbour in-plane spring forces, [2, z, y, x] (mesh.py:42-169)."""
  if len(stride) != 2:
    raise ValueError('stride must be 2D.')
  x = np.asarray(x, f32)
  sx, sy = float(stride[0]), float(stride[1])
  diag = f32(np.linalg.norm(np.array(stride)))
  k2 = f32(k) / np.sqrt(f32(2.0))
  out = np.zeros_like(x)
  for direction, l0, kk in (
      ((1, 0), f32(sx), f32(k)),
      ((0, 1), f32(sy), f32(k)),
      ((1, 1), diag, k2),
      ((-1, 1), diag, k2),
  ):
    rest = (f32(direction[0] * sx), f32(direction[1] * sy))
    far, near = _shift_views(2, direction)
    lead = (slice(None),) * (x.ndim - 2)
    r = np.array(rest, f32).reshape((2,) + (1,) * (x.ndim - 1))
    d = x[lead + far] - x[lead + near] + r
    length = np.sqrt(np.square(d[0]) + np.square(d[1]))
    f = _hooke(d, length, l0, kk, direction, prefer_orig_order)
    out[lead + far] += f
    out[lead + near] -= f
  return out


def elastic_mesh_3d(x, k, stride, prefer_orig_order=False, links=LINKS_3D):
  """26-neighbour (or `links` subset) forces, [3, ..., z, y, x] (mesh.py:192-279)."""
  x = np.asarray(x, f32)
  assert x.shape[0] == 3
  if np.ndim(stride) == 0:
    stride = (stride,) * 3
  stride = np.array(stride, dtype=np.float64)
  out = np.zeros_like(x)
  lead = (slice(None),) * (x.ndim - 3)
  for direction in links:
    rest = np.array(stride * np.array(direction), f32)
    l0 = f32(np.linalg.norm(rest))
    k_eff = k * stride[0] / l0
    far, near = _shift_views(3, direction)
    r = rest.reshape((3,) + (1,) * (x.ndim - 1))
    d = x[lead + far] - x[lead + near] + r
    length = np.sqrt(np.square(d[0]) + np.square(d[1]) + np.square(d[2]))
    f = _hooke(d, length, l0, k_eff, direction, prefer_orig_order)
    out[lead + far] += f
    out[lead + near] -= f
  return out


# ---------------------------------------------------------------------------
# integrator   (mesh.py:371-521)
# ---------------------------------------------------------------------------
def _total_force(x, prev, cap, cfg, mesh_force, prev_fn):
  a = mesh_force(x, cfg.k, cfg.stride, cfg.prefer_orig_order)
  if prev_fn is not None:
    prev = prev_fn(x)
  if prev is not None:
    pull = f32(-cfg.k0) * np.nan_to_num(x - np.asarray(prev, f32))
    a = a + np.clip(pull, -f32(cap), f32(cap))
  return a.astype(f32)


def velocity_verlet(x, v, prev, cfg, force_cap, fire_dt=None, fire_alpha=None,
                    mesh_force=inplane_force, prev_fn=None, snapshots=None,
                    snapshot_every=0, resume=None):
  """cfg.num_iters damped-VV or FIRE steps.

  Returns (x, v, a) or, with FIRE, (x, v, a, dt, alpha, n_pos, cap).
  `snapshots` (a list, FIRE only; test instrumentation, not in the reference):
  receives (step, x, dt, alpha, n_pos, cap) after every `snapshot_every` steps,
  so that a test can follow ONE long chunk without cutting it (a cut would
  restart n_pos, mesh.py:448).  `resume` = (a, n_pos) (test instrumentation):
  continue a chunk from a state taken in the middle of it -- the acceleration
  as it was (computed under the force cap of the step before) and the count of
  downhill steps -- instead of starting one.
  """
  x = np.array(x, f32)
  v = np.array(v, f32)
  cap = f32(force_cap)
  if resume is not None:
    a = np.array(resume[0], f32)
  else:
    a = _total_force(x, prev, cap, cfg, mesh_force, prev_fn)

  def vv(x, v, a, dt, cap):
    x = x + dt * v + f32(0.5) * dt * dt * a
    a_new = _total_force(x, prev, cap, cfg, mesh_force, prev_fn)
    g = f32(cfg.gamma)
    v = (f32(1.0) / (f32(1.0) + f32(0.5) * dt * g)) * (
        v * (f32(1.0) - f32(0.5) * dt * g) + f32(0.5) * dt * (a + a_new))
    return x, v.astype(f32), a_new

  if not cfg.fire:
    for _ in range(cfg.num_iters):
      x, v, a = vv(x, v, a, f32(cfg.dt), cap)
    return x, v, a

  dt = f32(cfg.dt if fire_dt is None else fire_dt)
  alpha = f32(cfg.alpha if fire_alpha is None else fire_alpha)
  n_pos = 0 if resume is None else int(resume[1])
  dt_cap = f32(float(cfg.dt_max) * float(cfg.dt))
  power = 0.0
  for step in range(cfg.num_iters):
    if snapshots is not None and step and step % snapshot_every == 0:
      # (+ the last power and its scale |a||v|: how marginal the last branch was)
      snapshots.append((step, x.copy(), f32(dt), f32(alpha), n_pos, f32(cap),
                        float(power), float(np.linalg.norm(a) * np.linalg.norm(v))))
    x, v, a = vv(x, v, a, dt, cap)
    a_n = np.sqrt(np.sum(np.square(a), axis=0, keepdims=True)) + f32(1e-6)
    v_n = np.sqrt(np.sum(np.square(v), axis=0, keepdims=True))
    power = np.vdot(a, v)
    v = v + alpha * (a / a_n * v_n - v)
    uphill = not (power >= 0)
    n_pos = 0 if uphill else n_pos + 1
    if uphill:
      dt = dt * f32(cfg.f_dec)
      alpha = f32(cfg.alpha)
      v = v * f32(0)
    else:
      if n_pos > cfg.n_min:
        dt = min(dt * f32(cfg.f_inc), dt_cap)
        alpha = alpha * f32(cfg.f_alpha)
      if n_pos > 0 and n_pos % cfg.cap_upscale_every == 0:
        cap = f32(cfg.cap_scale) * cap
    cap = min(cap, f32(cfg.final_cap))
    if cfg.remove_drift:
      # The reference hard-codes axes (1, 2, 3) (mesh.py:496-497).
      x = x - x.mean(axis=(1, 2, 3), keepdims=True, dtype=f32)
      v = v - v.mean(axis=(1, 2, 3), keepdims=True, dtype=f32)
  return x, v, a, f32(dt), f32(alpha), n_pos, f32(cap)


def relax_mesh(x, prev, cfg, mesh_force=inplane_force, prev_fn=None):
  """Chunked relaxation loop (mesh.py:524-608) -> (x, e_kin list, steps)."""
  if cfg.start_cap != cfg.final_cap:
    if not cfg.fire:
      raise NotImplementedError(
          'Adaptive force capping is only supported with FIRE.')
    if cfg.cap_scale <= 1:
      raise ValueError(
          'The scaling factor for the force cap has to be larger '
          'than 1 when the initial and final cap are different.')
  if prev is not None and prev_fn is not None:
    raise ValueError('Only one of: "prev" and "prev_fn" can be specified.')
  x = np.array(x, f32)
  v = np.zeros_like(x)
  t = 0
  dt, alpha, cap = cfg.dt, cfg.alpha, cfg.start_cap
  e_kin = []
  while t < cfg.max_iters:
    st = velocity_verlet(x, v, prev, cfg, cap, dt, alpha, mesh_force, prev_fn)
    t += cfg.num_iters
    x, v = st[0], st[1]
    speed2 = np.sum(np.square(v), axis=0)
    e_kin.append(float(np.sum(speed2)))
    v_max = float(np.sqrt(speed2.max()))
    if cfg.fire:
      dt, alpha, _, cap = st[-4:]
    if v_max < cfg.stop_v_max:
      if cap >= cfg.final_cap:
        break
      cap = min(cap * cfg.cap_scale, cfg.final_cap)
  return x, e_kin, t


def relax_mesh_passes(x, prev, cfg, mask=None, mesh_min_frac=0.5, start_fn=None):
  """The three-pass driver of processor/mesh.py:428-513 (RelaxMesh.relax_mesh):
  relax; on folds relax a fresh mesh towards the solution with k0 / 10; if that
  is regular, relax it against the real targets.  Returns (x, e_kin, steps,
  status) with status 0 regular, 1 prep failed, 2 regularised.  `start_fn(x0,
  prev)` plays maybe_update_init_state (processor/mesh.py:387-398)."""
  import copy
  from oracle import maps_oracle

  def mask_irregular(m, **kw):      # in place, like map_utils.mask_irregular
    masked, bad = maps_oracle.mask_irregular(m[:, 0], cfg.stride, mesh_min_frac, **kw)
    m[:, 0] = masked
    return bad

  x = np.array(x, f32)
  if mask is not None:
    x[:, mask] = np.nan
  x, e_kin, steps = relax_mesh(x, prev, cfg)
  orig = x.copy()
  if not mask_irregular(x, dilation_iters=5).any():
    return x, e_kin, steps, 0
  start = np.zeros_like(x)
  if start_fn is not None:
    start = start_fn(start, prev)
  soft = copy.copy(cfg)
  soft.k0 = cfg.k0 / 10.0
  x, _, prep = relax_mesh(start, x, soft)
  if mask_irregular(x).any():
    return orig, e_kin, steps + prep, 1
  if mask is not None:
    x[:, mask] = np.nan
  x, e_kin2, reg = relax_mesh(x, prev, cfg)
  return x, e_kin2, steps + prep + reg, 2
