"""Elastic spring-mesh relaxation on MI355X.

Drop-in for the reference's `sofima/mesh.py`: the same public names,
signatures, argument meaning, error behaviour and result layout, with the
device work done by hand-written HIP kernels behind the C ABI of
libsofima_amd.so (include/sofima_amd.h).  No JAX, no CPU fallback.

  inplane_force        <-> mesh.py:42-169
  MESH_LINK_DIRECTIONS <-> mesh.py:172-189
  elastic_mesh_3d      <-> mesh.py:192-279
  IntegrationConfig    <-> mesh.py:282-338
  velocity_verlet      <-> mesh.py:371-521
  relax_mesh           <-> mesh.py:524-608

Positions are stored in relative format: the (i, j)-th node of a grid with
stride D holding (dx, dy) sits at (i * D + dx, j * D + dy).  Arrays are
[C, z, y, x] (C = 2, in-plane) or [3, [batch,] z, y, x]; the vector components
are in x, y[, z] order.  Results are returned as `DeviceArray`s that stay in
HBM; `np.array(result)` copies them to the host as the reference's callers do
with jax arrays.
"""
from __future__ import annotations

import collections.abc
import ctypes as C
import dataclasses
import functools
import json
import logging
from typing import Any, Sequence

import numpy as np
import torch

from . import _abi
from . import _dev
from ._dev import DeviceArray


# ---------------------------------------------------------------------------
# force functions
# ---------------------------------------------------------------------------
MESH_LINK_DIRECTIONS = (  # xyz
    # 6 nearest neighbors
    (1, 0, 0),
    (0, 1, 0),
    (0, 0, 1),
    # 12 next-nearest neighbors
    (1, 1, 0),
    (-1, 1, 0),
    (1, 0, 1),
    (-1, 0, 1),
    (0, 1, 1),
    (0, -1, 1),
    # 8 next-next-nearest neighbors
    (1, 1, 1),
    (1, 1, -1),
    (1, -1, 1),
    (-1, 1, 1),
)


def _base_desc(x_t: torch.Tensor, spec, k: float, stride,
               prefer_orig_order: bool) -> _abi.SfmMeshDesc:
  d = _abi.SfmMeshDesc()
  ncomp = spec.ncomp if spec.ncomp is not None else int(x_t.shape[0])
  if ncomp not in (2, 3) or x_t.shape[0] != ncomp:
    raise ValueError(f'x must be [{ncomp}, ...], got {tuple(x_t.shape)}')
  links = spec.links
  d.ncomp = ncomp
  sp = tuple(x_t.shape[1:])
  if spec.kind != _abi.FORCE_SPRINGS:
    # no link stencil across sections: every leading axis is a batch of planes
    if len(sp) < 2:
      raise ValueError('mesh states must be [c, ..., y, x]')
    lead = int(np.prod(sp[:-2])) if len(sp) > 2 else 1
    shape = (1, lead) + sp[-2:]
  elif ncomp == 2:
    if len(sp) != 3:
      raise ValueError('in-plane meshes must be [2, z, y, x]')
    shape = (1,) + sp
  else:
    if len(sp) < 3:
      raise ValueError('3d meshes must be [3, [batch..], z, y, x]')
    batch = int(np.prod(sp[:-3])) if len(sp) > 3 else 1
    shape = (batch,) + sp[-3:]
  d.shape = (C.c_int32 * 4)(*[int(s) for s in shape])
  st = [float(s) for s in stride][:3]
  st += [0.0] * (3 - len(st))
  d.stride = (C.c_double * 3)(*st)
  d.k = float(k)
  d.prefer_orig_order = int(bool(prefer_orig_order))
  if links is not None and tuple(map(tuple, links)) != MESH_LINK_DIRECTIONS:
    links = [tuple(int(v) for v in l) for l in links]
    if len(links) > _abi.MAX_LINKS or len(links) < 1:
      raise ValueError(f'between 1 and {_abi.MAX_LINKS} links are supported')
    for l in links:
      if len(l) != 3 or any(abs(v) > 1 for v in l):
        raise ValueError('Only |v| <= 1 values supported within links.')
    d.n_links = len(links)
    for i, l in enumerate(links):
      for c in range(3):
        d.links[i][c] = l[c]
  d.x = x_t.data_ptr()
  d.stream = _dev.stream_ptr()
  d.force_kind = spec.kind
  if spec.tile is not None:
    spec.tile.bind(d, x_t)
  return d


def _force(x, ncomp, k, stride, prefer_orig_order, links=None) -> DeviceArray:
  dev = _dev.device()
  x_t = _dev.as_device_f32(x, dev, copy=False)
  d = _base_desc(x_t, _SpringSpec(ncomp, links), k, stride, prefer_orig_order)
  out = torch.empty_like(x_t)
  _abi.check(_abi.load().sfm_mesh_force(C.byref(d), out.data_ptr()))
  return DeviceArray(out)


def _SpringSpec(ncomp, links=None):
  return _ForceSpec(_abi.FORCE_SPRINGS, ncomp, links=links)


def inplane_force(x, k: float, stride: Sequence[float],
                  prefer_orig_order: bool = False) -> DeviceArray:
  """In-plane (8-neighbour) spring forces on a [2, z, y, x] mesh.

  Same contract as mesh.inplane_force (mesh.py:42-169); `stride` is (x, y).
  """
  if len(stride) != 2:
    raise ValueError('stride must be 2D.')
  if np.shape(x)[0] != 2:
    raise ValueError('x must be [2, z, y, x]')
  return _force(x, 2, k, stride, prefer_orig_order)


def elastic_mesh_3d(x, k: float, stride: float | Sequence[float],
                    prefer_orig_order: bool = False,
                    links=MESH_LINK_DIRECTIONS) -> DeviceArray:
  """Internal forces of a 3-d spring mesh, [3, [batch..], z, y, x].

  Same contract as mesh.elastic_mesh_3d (mesh.py:192-279); `stride` is a
  scalar or (x, y, z); `links` selects the spring families.
  """
  assert np.shape(x)[0] == 3
  if not isinstance(stride, collections.abc.Sequence):
    stride = (stride,) * 3
  return _force(x, 3, k, stride, prefer_orig_order, links)


# ---------------------------------------------------------------------------
# configuration
# ---------------------------------------------------------------------------
class _JsonMixin:
  """The subset of dataclasses_json.DataClassJsonMixin callers rely on."""

  def to_dict(self) -> dict[str, Any]:
    return dataclasses.asdict(self)

  def to_json(self, **kw) -> str:
    return json.dumps(self.to_dict(), **kw)

  @classmethod
  def from_dict(cls, kvs: dict[str, Any], **_):
    names = {f.name for f in dataclasses.fields(cls)}
    return cls(**{k: v for k, v in kvs.items() if k in names})

  @classmethod
  def from_json(cls, s: str, **_):
    return cls.from_dict(json.loads(s))


@dataclasses.dataclass(frozen=True)
class IntegrationConfig(_JsonMixin):
  """Parameters for numerical integration of the mesh state.

  Field for field the reference's dataclass (mesh.py:282-338).
  """

  dt: float  # time step size
  gamma: float  # damping constant
  k0: float  # spring constant for inter-section springs
  k: float  # spring constant for intra-section springs
  # distance between nearest neighbors of the point grid
  stride: tuple[float, float] | tuple[float, float, float]
  num_iters: int  # number of time steps to execute at once
  max_iters: int  # upper bound for simulation time

  # The simulation terminates when the velocity of all nodes is below this
  # value; with FIRE the force cap must also have reached `final_cap`.
  stop_v_max: float

  fire: bool = True  # use the Fast Inertial Relaxation Engine

  # FIRE parameters.
  f_alpha: float = 0.99
  f_inc: float = 1.1
  f_dec: float = 0.5
  alpha: float = 0.1
  n_min: int = 5  # min. number of steps after which to increase step size
  dt_max: float = 10.0  # max time step size, in units of `dt`

  # Initial and final cap of the inter-section force component magnitude;
  # start_cap != final_cap requires FIRE.
  start_cap: float = 1e6
  final_cap: float = 1e6
  cap_scale: float = 1.1  # upscaling factor for the force cap (> 1)
  # Steps of uninterrupted positive power between force-cap upscalings.
  cap_upscale_every: int = 100

  # Favour the original relative ordering of the nodes (prevents folds).
  prefer_orig_order: bool = False
  # Remove global drift (mean position and mean speed) after every step.
  remove_drift: bool = False

  def __post_init__(self):
    object.__setattr__(self, 'stride', tuple(self.stride))


# ---------------------------------------------------------------------------
# integrator
# ---------------------------------------------------------------------------
class TileMeshForce:
  """Native `mesh_force` of a tile mesh: stitch_rigid.elastic_tile_mesh
  (stitch_rigid.py:330-388) for [2, z, y, x] states, elastic_tile_mesh_3d
  (:391-473) for [3, z, y, x] states, closed over the desired tile offsets
  `cx` / `cy` like the `_mesh_force` closure of optimize_coarse_mesh (:509-510).
  Passing it as `mesh_force=` keeps the whole relaxation in the HIP integrator.
  """

  def __init__(self, cx, cy):
    dev = _dev.device()
    self.cx = _dev.as_device_f32(cx, dev, copy=False)
    self.cy = _dev.as_device_f32(cy, dev, copy=False)
    if self.cx.shape != self.cy.shape or self.cx.ndim != 4 or \
        self.cx.shape[0] not in (2, 3):
      raise ValueError('cx, cy must be [2 or 3, z, y, x] arrays of one shape')
    self.ncomp = int(self.cx.shape[0])

  def bind(self, d: _abi.SfmMeshDesc, x_t: torch.Tensor):
    if tuple(x_t.shape) != tuple(self.cx.shape):
      raise ValueError(f'x {tuple(x_t.shape)} and cx/cy {tuple(self.cx.shape)} '
                       'must have the same shape')
    d.force_kind = _abi.FORCE_TILE_MESH
    d.cx = self.cx.data_ptr()
    d.cy = self.cy.data_ptr()

  def __call__(self, x, k=None, stride=None, prefer_orig_order=False,
               links=None) -> DeviceArray:
    del k, stride, prefer_orig_order, links
    dev = _dev.device()
    x_t = _dev.as_device_f32(x, dev, copy=False)
    d = _base_desc(x_t, _ForceSpec(_abi.FORCE_TILE_MESH, self.ncomp, tile=self),
                   0.0, (1.0,) * self.ncomp, False)
    out = torch.empty_like(x_t)
    _abi.check(_abi.load().sfm_mesh_force(C.byref(d), out.data_ptr()))
    return DeviceArray(out)


@dataclasses.dataclass
class _ForceSpec:
  """What the integrator evaluates as `mesh_force`."""
  kind: int                     # _abi.FORCE_*
  ncomp: int | None             # None: taken from x (external callables)
  links: Any = None             # spring stencils: custom link subset
  tile: TileMeshForce | None = None
  fn: Any = None                # external: the caller's callable


def _resolve_force(mesh_force) -> _ForceSpec:
  """Maps a `mesh_force` callable to the force model of the HIP integrator.

  The package's own force functions run fused inside the integrator kernels.
  Any other callable `f(x, k, stride, prefer_orig_order) -> force` (the
  reference's contract, mesh.py:427-428) is evaluated between the position
  update and the velocity update of every step, on the device: it receives the
  live positions as a `DeviceArray` (`.tensor` is the torch CUDA tensor) and
  may return a DeviceArray, a torch tensor or a NumPy array.
  """
  if mesh_force is inplane_force:
    return _ForceSpec(_abi.FORCE_SPRINGS, 2)
  if mesh_force is elastic_mesh_3d:
    return _ForceSpec(_abi.FORCE_SPRINGS, 3)
  if isinstance(mesh_force, functools.partial) and \
      mesh_force.func is elastic_mesh_3d and not mesh_force.args and \
      set(mesh_force.keywords) <= {'links'}:
    return _ForceSpec(_abi.FORCE_SPRINGS, 3, links=mesh_force.keywords.get('links'))
  if isinstance(mesh_force, TileMeshForce):
    return _ForceSpec(_abi.FORCE_TILE_MESH, mesh_force.ncomp, tile=mesh_force)
  if callable(mesh_force):
    return _ForceSpec(_abi.FORCE_EXTERNAL, None, fn=mesh_force)
  raise TypeError('mesh_force must be callable')


def _resolve_prev_fn(prev_fn):
  """(native TargetMeshFn or None, generic callable or None) for `prev_fn`.

  A `stitch_elastic.TargetMeshFn` (the native form of
  `jax.vmap(compute_target_mesh)`) runs as a kernel inside the integrator.  Any
  other callable `f(x) -> prev` (mesh.py:429-430) is evaluated on the live,
  device-resident positions in front of every force evaluation, like the
  reference's traced closure: it receives a `DeviceArray` and may return a
  DeviceArray, a torch tensor or a NumPy array (NaN = no spring).
  """
  if prev_fn is None:
    return None, None
  from . import stitch_elastic
  if isinstance(prev_fn, stitch_elastic.TargetMeshFn):
    return prev_fn, None
  if callable(prev_fn):
    return None, prev_fn
  raise TypeError('prev_fn must be callable')


def _chunk_desc(x_t, v_t, a_t, prev_t, config: IntegrationConfig, spec,
                ws) -> _abi.SfmMeshDesc:
  d = _base_desc(x_t, spec, config.k, config.stride, config.prefer_orig_order)
  if spec.kind == _abi.FORCE_SPRINGS and spec.ncomp == 2 and \
      len(config.stride) != 2:
    raise ValueError('stride must be 2D.')
  d.k0 = float(config.k0)
  d.dt = float(config.dt)
  d.gamma = float(config.gamma)
  d.num_iters = int(config.num_iters)
  d.fire = int(bool(config.fire))
  d.f_alpha = float(config.f_alpha)
  d.f_inc = float(config.f_inc)
  d.f_dec = float(config.f_dec)
  d.alpha0 = float(config.alpha)
  d.n_min = int(config.n_min)
  d.dt_max = float(config.dt_max)
  d.final_cap = float(config.final_cap)
  d.cap_scale = float(config.cap_scale)
  d.cap_upscale_every = int(config.cap_upscale_every)
  # The reference takes the drift means over axes (1, 2, 3) (mesh.py:496-497):
  # global for [c, z, y, x] states, per x column for 5-D [c, n, z, y, x] states.
  d.remove_drift = (2 if x_t.ndim == 5 else 1) if config.remove_drift else 0
  d.v = v_t.data_ptr()
  d.a = a_t.data_ptr()
  d.prev = prev_t.data_ptr() if prev_t is not None else None
  d.workspace = ws.data_ptr()
  d.workspace_bytes = ws.numel()
  return d


def _external_force(spec, x_t, config):
  """(force buffer, ctypes callback, error list) for a caller-provided
  mesh_force: the callback evaluates it on the live positions and leaves the
  result in the buffer the integrate kernel reads (same stream)."""
  f_t = torch.empty_like(x_t)
  errors = []
  x_view = DeviceArray(x_t)

  def call(_user):
    try:
      f = spec.fn(x_view, config.k, config.stride, config.prefer_orig_order)
      f = _dev.as_device_f32(f, x_t.device, copy=False)
      if tuple(f.shape) != tuple(x_t.shape):
        raise ValueError(f'mesh_force returned shape {tuple(f.shape)}, '
                         f'expected {tuple(x_t.shape)}')
      f_t.copy_(f)
      return 0
    except BaseException as e:  # pylint: disable=broad-except
      errors.append(e)
      return 1

  return f_t, _abi.SfmForceCallback(call), errors


def _external_prev(prev_call, x_t):
  """(prev buffer, ctypes callback, error list) for a generic prev_fn."""
  p_t = torch.empty_like(x_t)
  errors = []
  x_view = DeviceArray(x_t)

  def call(_user):
    try:
      pv = _dev.as_device_f32(prev_call(x_view), x_t.device, copy=False)
      if tuple(pv.shape) != tuple(x_t.shape):
        raise ValueError(f'prev_fn returned shape {tuple(pv.shape)}, '
                         f'expected {tuple(x_t.shape)}')
      p_t.copy_(pv)
      return 0
    except BaseException as e:  # pylint: disable=broad-except
      errors.append(e)
      return 1

  return p_t, _abi.SfmForceCallback(call), errors


def _run_chunk(x_t, v_t, prev_t, config, force_cap, fire_dt, fire_alpha, spec,
               target=None, prev_call=None):
  """One velocity_verlet call on device tensors (x_t, v_t updated in place)."""
  lib = _abi.load()
  a_t = torch.empty_like(x_t)
  ext = None
  if spec.kind == _abi.FORCE_EXTERNAL:
    ext = _external_force(spec, x_t, config)
  extp = None if prev_call is None else _external_prev(prev_call, x_t)
  probe = _base_desc(x_t, spec, config.k, config.stride,
                     config.prefer_orig_order)
  tdesc = None
  if target is not None:
    tdesc = target.bind(x_t)
    probe.target = C.pointer(tdesc)
  ws = _dev.workspace(lib.sfm_mesh_workspace_bytes(C.byref(probe)),
                      x_t.device)
  d = _chunk_desc(x_t, v_t, a_t, prev_t, config, spec, ws)
  if tdesc is not None:
    d.target = C.pointer(tdesc)
  if ext is not None:
    d.ext_force = ext[0].data_ptr()
    d.force_cb = ext[1]
  if extp is not None:
    d.ext_prev = extp[0].data_ptr()
    d.prev_cb = extp[1]
  fire = _abi.SfmFireState()
  fire.dt = np.float32(config.dt if fire_dt is None else fire_dt)
  fire.alpha = np.float32(config.alpha if fire_alpha is None else fire_alpha)
  fire.n_pos = 0
  fire.cap = np.float32(force_cap)
  stats = _abi.SfmChunkStats()
  rc = lib.sfm_mesh_relax_chunk(C.byref(d), C.byref(fire), C.byref(stats))
  if ext is not None and ext[2]:
    raise ext[2][0]  # the caller's mesh_force raised: surface its exception
  if extp is not None and extp[2]:
    raise extp[2][0]
  _abi.check(rc)
  return a_t, fire, stats


def velocity_verlet(x, v, prev, config: IntegrationConfig, force_cap: float,
                    fire_dt: float | None = None,
                    fire_alpha: float | None = None,
                    mesh_force=inplane_force, prev_fn=None):
  """Executes `config.num_iters` (damped) velocity Verlet / FIRE steps.

  Same contract as mesh.velocity_verlet (mesh.py:371-521): returns
  (x, v, a) or, with FIRE, (x, v, a, dt, alpha, n_pos, cap).  The inputs are
  not modified.
  """
  target, prev_call = _resolve_prev_fn(prev_fn)
  if prev_fn is not None and prev is not None:
    raise ValueError('Only one of: "prev" and "prev_fn" can be specified.')
  spec = _resolve_force(mesh_force)
  dev = _dev.device()
  x_t = _dev.as_device_f32(x, dev, copy=True)
  v_t = _dev.as_device_f32(v, dev, copy=True)
  prev_t = None if prev is None else _dev.as_device_f32(prev, dev, copy=False)
  a_t, fire, _ = _run_chunk(x_t, v_t, prev_t, config, force_cap, fire_dt,
                            fire_alpha, spec, target, prev_call)
  out = (DeviceArray(x_t), DeviceArray(v_t), DeviceArray(a_t))
  if config.fire:
    out += (np.float32(fire.dt), np.float32(fire.alpha), int(fire.n_pos),
            np.float32(fire.cap))
  return out


def relax_mesh(x, prev, config: IntegrationConfig, mesh_force=inplane_force,
               prev_fn=None) -> tuple[DeviceArray, list[float], int]:
  """Simulates mesh relaxation (mesh.py:524-608).

  Returns (relaxed positions [device], kinetic-energy history, steps run).
  """
  t = 0
  dt = config.dt
  alpha = config.alpha
  e_kin = []
  cap = config.start_cap

  if config.start_cap != config.final_cap:
    if not config.fire:
      raise NotImplementedError(
          'Adaptive force capping is only supported with FIRE.')
    if config.cap_scale <= 1:
      raise ValueError(
          'The scaling factor for the force cap has to be larger '
          'than 1 when the initial and final cap are different.')

  if prev is not None and prev_fn is not None:
    raise ValueError('Only one of: "prev" and "prev_fn" can be specified.')
  target, prev_call = _resolve_prev_fn(prev_fn)

  spec = _resolve_force(mesh_force)
  dev = _dev.device()
  x_t = _dev.as_device_f32(x, dev, copy=True)
  v_t = torch.zeros_like(x_t)
  prev_t = None if prev is None else _dev.as_device_f32(prev, dev, copy=False)

  while t < config.max_iters:
    _, fire, stats = _run_chunk(x_t, v_t, prev_t, config, cap, dt, alpha,
                                spec, target, prev_call)
    t += config.num_iters
    e_kin.append(float(stats.e_kin))
    v_max = float(stats.v_max)

    if config.fire:
      dt, alpha, n_pos, cap = (np.float32(fire.dt), np.float32(fire.alpha),
                               int(fire.n_pos), np.float32(fire.cap))
      logging.info(
          't=%r: dt=%f, alpha=%f, n_pos=%d, cap=%f, v_max=%f, e_kin=%f',
          t, dt, alpha, n_pos, cap, v_max, e_kin[-1])

    if v_max < config.stop_v_max:
      # float32 comparison like the reference's weakly typed JAX scalars
      # (final_cap = 0.7 must compare equal to the kernel's float32 cap)
      if np.float32(cap) >= np.float32(config.final_cap):
        break
      # Increase cap to ensure progress towards the termination condition.
      cap = min(cap * config.cap_scale, config.final_cap)

  return DeviceArray(x_t), e_kin, t
