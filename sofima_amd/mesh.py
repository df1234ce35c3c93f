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


def _base_desc(x_t: torch.Tensor, ncomp: int, k: float, stride,
               prefer_orig_order: bool, links=None) -> _abi.SfmMeshDesc:
  d = _abi.SfmMeshDesc()
  d.ncomp = ncomp
  sp = tuple(x_t.shape[1:])
  if ncomp == 2:
    if len(sp) != 3:
      raise ValueError('in-plane meshes must be [2, z, y, x]')
    shape = (1,) + sp
  else:
    if len(sp) < 3:
      raise ValueError('3d meshes must be [3, [batch..], z, y, x]')
    batch = int(np.prod(sp[:-3])) if len(sp) > 3 else 1
    shape = (batch,) + sp[-3:]
  d.shape = (C.c_int32 * 4)(*[int(s) for s in shape])
  st = [float(s) for s in stride] + [0.0] * (3 - len(stride))
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
  return d


def _force(x, ncomp, k, stride, prefer_orig_order, links=None) -> DeviceArray:
  dev = _dev.device()
  x_t = _dev.as_device_f32(x, dev, copy=False)
  d = _base_desc(x_t, ncomp, k, stride, prefer_orig_order, links)
  out = torch.empty_like(x_t)
  _abi.check(_abi.load().sfm_mesh_force(C.byref(d), out.data_ptr()))
  return DeviceArray(out)


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
def _resolve_force(mesh_force):
  """Maps a `mesh_force` callable to (ncomp, links) of a native kernel."""
  if mesh_force is inplane_force:
    return 2, None
  if mesh_force is elastic_mesh_3d:
    return 3, None
  if isinstance(mesh_force, functools.partial) and \
      mesh_force.func is elastic_mesh_3d and not mesh_force.args and \
      set(mesh_force.keywords) <= {'links'}:
    return 3, mesh_force.keywords.get('links')
  raise NotImplementedError(
      'mesh_force must be sofima_amd.mesh.inplane_force, elastic_mesh_3d or '
      'functools.partial(elastic_mesh_3d, links=...); arbitrary Python '
      'callables cannot be fused into the HIP integrator (see DESIGN.md).')


def _native_prev_fn(prev_fn):
  """Returns the TargetMeshFn behind `prev_fn`, or raises for JAX closures."""
  from . import stitch_elastic
  if isinstance(prev_fn, stitch_elastic.TargetMeshFn):
    return prev_fn
  raise NotImplementedError(
      'prev_fn must be a sofima_amd.stitch_elastic.TargetMeshFn; arbitrary '
      'Python callables cannot be fused into the HIP integrator (DESIGN.md)')


def _chunk_desc(x_t, v_t, a_t, prev_t, config: IntegrationConfig, ncomp, links,
                ws) -> _abi.SfmMeshDesc:
  d = _base_desc(x_t, ncomp, config.k, config.stride, config.prefer_orig_order,
                 links)
  if ncomp == 2 and len(config.stride) != 2:
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


def _run_chunk(x_t, v_t, prev_t, config, force_cap, fire_dt, fire_alpha, ncomp,
               links, target=None):
  """One velocity_verlet call on device tensors (x_t, v_t updated in place)."""
  lib = _abi.load()
  a_t = torch.empty_like(x_t)
  probe = _base_desc(x_t, ncomp, config.k, config.stride,
                     config.prefer_orig_order, links)
  tdesc = None
  if target is not None:
    tdesc = target.bind(x_t)
    probe.target = C.pointer(tdesc)
  ws = _dev.workspace(lib.sfm_mesh_workspace_bytes(C.byref(probe)),
                      x_t.device)
  d = _chunk_desc(x_t, v_t, a_t, prev_t, config, ncomp, links, ws)
  if tdesc is not None:
    d.target = C.pointer(tdesc)
  fire = _abi.SfmFireState()
  fire.dt = np.float32(config.dt if fire_dt is None else fire_dt)
  fire.alpha = np.float32(config.alpha if fire_alpha is None else fire_alpha)
  fire.n_pos = 0
  fire.cap = np.float32(force_cap)
  stats = _abi.SfmChunkStats()
  _abi.check(lib.sfm_mesh_relax_chunk(C.byref(d), C.byref(fire),
                                      C.byref(stats)))
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
  target = None if prev_fn is None else _native_prev_fn(prev_fn)
  if target is not None and prev is not None:
    raise ValueError('Only one of: "prev" and "prev_fn" can be specified.')
  ncomp, links = _resolve_force(mesh_force)
  dev = _dev.device()
  x_t = _dev.as_device_f32(x, dev, copy=True)
  v_t = _dev.as_device_f32(v, dev, copy=True)
  prev_t = None if prev is None else _dev.as_device_f32(prev, dev, copy=False)
  a_t, fire, _ = _run_chunk(x_t, v_t, prev_t, config, force_cap, fire_dt,
                            fire_alpha, ncomp, links, target)
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
  target = None if prev_fn is None else _native_prev_fn(prev_fn)

  ncomp, links = _resolve_force(mesh_force)
  dev = _dev.device()
  x_t = _dev.as_device_f32(x, dev, copy=True)
  v_t = torch.zeros_like(x_t)
  prev_t = None if prev is None else _dev.as_device_f32(prev, dev, copy=False)

  while t < config.max_iters:
    _, fire, stats = _run_chunk(x_t, v_t, prev_t, config, cap, dt, alpha,
                                ncomp, links, target)
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
