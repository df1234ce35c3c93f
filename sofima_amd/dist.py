"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed.

The path shards by INDEPENDENT units and needs no data-path collective
(SURVEY.md 8e): the patches of one flow field are split by whole batches
(batch membership must not change, because the reference's second-peak
suppression and masked-NCC tolerances are batch-coupled), tile pairs and
section pairs are split as units, and every rank relaxes its own meshes.  The
only communication is the gather of the (small) results.

Backends: "nccl" (= RCCL over xGMI) on GPUs; "gloo" for the CPU tests, which
replace the device call by a host function (`batch_fn`) to check that sharded
and single-process results are identical.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np
import torch.distributed as dist


# True: a process group of ONE rank still goes through its collectives (RCCL
# accepts a world of one): the nccl-only branches -- `gather_boundaries` on
# device tensors, the library communicator bootstrapped from a broadcast id --
# then execute on a single-GPU box (`bench.py --force-multi-gpu-legs`, tests).
import os as _os
FORCE_COLLECTIVES = None   # None: the environment decides; True / False: set by the caller


def force_collectives() -> bool:
  """The switch, read where it is used: `FORCE_COLLECTIVES` if a caller assigned it
  (bench.py --force-multi-gpu-legs), else SFM_FORCE_COLLECTIVES=1 in the environment."""
  if FORCE_COLLECTIVES is not None:
    return bool(FORCE_COLLECTIVES)
  return _os.environ.get('SFM_FORCE_COLLECTIVES') == '1'


def _alone(ws: int) -> bool:
  """No collective needed: one rank, unless collectives are forced."""
  return ws == 1 and not (force_collectives() and dist.is_available() and
                          dist.is_initialized())


def world(group=None) -> tuple[int, int]:
  if dist.is_available() and dist.is_initialized():
    return dist.get_rank(group), dist.get_world_size(group)
  return 0, 1


def shard_units(n_units: int, rank: int, world_size: int) -> list[int]:
  """Round-robin assignment of unit indices (batches, tile pairs, sections)."""
  return list(range(rank, n_units, world_size))


def gather_objects(obj, group=None) -> list:
  rank, ws = world(group)
  if _alone(ws):
    return [obj]
  out = [None] * ws
  dist.all_gather_object(out, obj, group=group)
  return out


def sharded_flow_field(calc, pre_image, post_image, patch_size, step, *,
                       pre_mask=None, post_mask=None,
                       mask_only_for_patch_selection=False,
                       selection_mask=None, max_masked=0.75, batch_size=4096,
                       post_patch_size=None, pre_targeting_field=None,
                       pre_targeting_step=None, post_targeting_field=None,
                       post_targeting_step=None, group=None,
                       batch_fn: Callable | None = None) -> np.ndarray:
  """`calc.flow_field(...)` with the batches of patches spread over the ranks.

  Every rank must hold the same images (or at least the rows its batches
  touch) and passes the same arguments; every rank returns the full field,
  bit-identical to the single-process result.

  batch_fn(pre_starts, post_starts) -> peaks [len, dim + 2] replaces the GPU
  call (CPU tests).
  """
  nd = pre_image.ndim
  seq = lambda v: tuple(int(a) for a in v) if np.ndim(v) else (int(v),) * nd
  patch_size = seq(patch_size)
  post_patch_size = patch_size if post_patch_size is None else seq(
      post_patch_size)
  step = seq(step)
  if pre_targeting_step is not None:
    pre_targeting_step = seq(pre_targeting_step)
  if post_targeting_step is not None:
    post_targeting_step = seq(post_targeting_step)
  plan = calc.plan(tuple(pre_image.shape), tuple(post_image.shape), patch_size,
                   step, pre_mask, post_mask, selection_mask, max_masked,
                   batch_size, post_patch_size, pre_targeting_field,
                   pre_targeting_step, post_targeting_field,
                   post_targeting_step)
  rank, ws = world(group)
  mine = shard_units(plan['n_batches'], rank, ws)
  if mask_only_for_patch_selection:
    pre_mask = post_mask = None
  if batch_fn is None:
    local = calc.compute_batches(pre_image, post_image, pre_mask, post_mask,
                                 patch_size, post_patch_size, plan, batch_size,
                                 mine)
  else:
    parts = [
        batch_fn(plan['pre_starts'][b * batch_size:(b + 1) * batch_size],
                 plan['post_starts'][b * batch_size:(b + 1) * batch_size])
        for b in mine
    ]
    local = (np.concatenate(parts) if parts else
             np.zeros((0, nd + 2), np.float32))
  peaks = np.zeros((plan['n_batches'] * batch_size, nd + 2), np.float32)
  for r, part in enumerate(gather_objects(np.asarray(local), group)):
    for k, b in enumerate(shard_units(plan['n_batches'], r, ws)):
      peaks[b * batch_size:(b + 1) * batch_size] = part[k * batch_size:
                                                        (k + 1) * batch_size]
  return calc.assemble(plan, nd, peaks)


def map_units(units: Sequence, fn: Callable, group=None) -> list:
  """Applies `fn` to this rank's share of independent `units` (section pairs,
  tile pairs, meshes) and returns the results of ALL units, in order, on
  every rank."""
  rank, ws = world(group)
  mine = shard_units(len(units), rank, ws)
  local = [fn(units[i]) for i in mine]
  out = [None] * len(units)
  for r, part in enumerate(gather_objects(local, group)):
    for k, i in enumerate(shard_units(len(units), r, ws)):
      out[i] = part[k]
  return out


# ---------------------------------------------------------------------------
# One mesh across GPUs: bands of rows with a 1-node halo.
# ---------------------------------------------------------------------------
def band_bounds(n_rows: int, n_bands: int) -> list[tuple[int, int]]:
  """Even split of the mesh rows into `n_bands` contiguous bands."""
  if n_bands < 1 or n_rows < n_bands:
    raise ValueError(f'cannot split {n_rows} rows into {n_bands} bands')
  edges = [n_rows * i // n_bands for i in range(n_bands + 1)]
  return list(zip(edges[:-1], edges[1:]))


class HipBand:
  """One band of a mesh on the current GPU, stepped through the C ABI
  (sfm_mesh_shard_*).  Local arrays are [C, ..., rows, X]: the owned rows plus
  one halo row towards every neighbouring band."""

  def __init__(self, x, prev, config, spec, own, global_nodes, n_bands):
    import ctypes as C
    import torch
    from . import _abi, _dev, mesh
    self._C, self._abi, self._mesh = C, _abi, mesh
    dev = _dev.device()
    self.config = config
    self.spec = spec
    self.own = own
    self.x = _dev.as_device_f32(x, dev, copy=True)
    self.v = torch.zeros_like(self.x)
    self.a = torch.empty_like(self.x)
    self.prev = None if prev is None else _dev.as_device_f32(prev, dev, copy=True)
    self.sums = torch.zeros((n_bands, 8), dtype=torch.float32, device=dev)
    self.my_sums = torch.zeros((8,), dtype=torch.float32, device=dev)
    lib = _abi.load()
    probe = mesh._base_desc(self.x, spec, config.k, config.stride,
                            config.prefer_orig_order)
    self.ws = _dev.workspace(lib.sfm_mesh_workspace_bytes(C.byref(probe)), dev)
    self.desc = mesh._chunk_desc(self.x, self.v, self.a, self.prev, config, spec,
                                 self.ws)
    sh = _abi.SfmMeshShard()
    sh.own_y0, sh.own_y1 = own
    sh.global_nodes = int(global_nodes)
    sh.n_ranks = n_bands
    sh.sums = self.sums.data_ptr()
    sh.my_sums = self.my_sums.data_ptr()
    self.shard = sh
    self.lib = lib

  def _refresh(self):
    from . import _dev
    self.desc.stream = _dev.stream_ptr()

  def begin(self, dt, alpha, cap):
    C, _abi = self._C, self._abi
    self._refresh()
    fire = _abi.SfmFireState()
    fire.dt, fire.alpha, fire.n_pos, fire.cap = (
        np.float32(dt), np.float32(alpha), 0, np.float32(cap))
    _abi.check(self.lib.sfm_mesh_shard_begin(C.byref(self.desc),
                                             C.byref(self.shard), C.byref(fire)))

  def advance(self):
    self._abi.check(self.lib.sfm_mesh_shard_advance(
        self._C.byref(self.desc), self._C.byref(self.shard)))

  def integrate(self):
    self._abi.check(self.lib.sfm_mesh_shard_integrate(
        self._C.byref(self.desc), self._C.byref(self.shard)))

  def finish(self):
    C, _abi = self._C, self._abi
    fire = _abi.SfmFireState()
    stats = _abi.SfmChunkStats()
    _abi.check(self.lib.sfm_mesh_shard_finish(C.byref(self.desc), C.byref(self.shard),
                                              C.byref(fire), C.byref(stats)))
    return (np.float32(fire.dt), np.float32(fire.alpha), int(fire.n_pos),
            np.float32(fire.cap), np.float32(stats.e_kin), np.float32(stats.v_max))

  # -- halo rows -------------------------------------------------------------
  def boundary(self, side):
    """(x, v, a) of the owned edge row at `side`, packed [3, C, ..., X]."""
    import torch
    row = self.own[0] if side == 'lo' else self.own[1] - 1
    return torch.stack([t[..., row, :] for t in (self.x, self.v, self.a)]).contiguous()

  def set_halo(self, side, packed):
    row = self.own[0] - 1 if side == 'lo' else self.own[1]
    for t, src in zip((self.x, self.v, self.a), packed):
      t[..., row, :] = src

  def owned_x(self) -> np.ndarray:
    return self.x[..., self.own[0]:self.own[1], :].cpu().numpy()


class BandTransport:
  """Moves halo rows and partial sums between the bands of one mesh.

  Bands that live in the same process exchange by direct copies; the first /
  last band of a process talks to the neighbouring rank through
  torch.distributed (gloo on CPU, nccl = RCCL on GPUs) or, with `comm`, through
  the library's own RCCL entry points (sfm_comm_halo_exchange /
  sfm_comm_allgather, one band per rank).
  """

  def __init__(self, group=None, comm=None):
    self.group = group
    self.rank, self.world = world(group)
    self.comm = comm
    # gloo moves host tensors only: device rows are staged through the host
    # (several ranks sharing one GPU in the tests; RCCL needs a GPU per rank)
    self.stage = (comm is None and self.world > 1 and
                  dist.get_backend(group) == 'gloo')

  def exchange(self, bands):
    for lo_band, hi_band in zip(bands[:-1], bands[1:]):
      up, down = lo_band.boundary('hi'), hi_band.boundary('lo')
      lo_band.set_halo('hi', down)
      hi_band.set_halo('lo', up)
    if self.world == 1:
      return
    first, last = bands[0], bands[-1]
    has_lo, has_hi = self.rank > 0, self.rank < self.world - 1
    send_lo = first.boundary('lo') if has_lo else None
    send_hi = last.boundary('hi') if has_hi else None
    import torch
    dev_of = (send_lo if has_lo else send_hi)
    dev_of = None if dev_of is None else dev_of.device
    if self.stage:
      send_lo = send_lo.cpu() if has_lo else None
      send_hi = send_hi.cpu() if has_hi else None
    recv_lo = torch.empty_like(send_lo) if has_lo else None
    recv_hi = torch.empty_like(send_hi) if has_hi else None
    if self.comm is not None:
      self.comm.halo_exchange(self.rank - 1 if has_lo else -1, send_lo, recv_lo,
                              self.rank + 1 if has_hi else -1, send_hi, recv_hi)
    else:
      ops = []
      if has_lo:
        ops += [dist.P2POp(dist.isend, send_lo, self.rank - 1, self.group),
                dist.P2POp(dist.irecv, recv_lo, self.rank - 1, self.group)]
      if has_hi:
        ops += [dist.P2POp(dist.isend, send_hi, self.rank + 1, self.group),
                dist.P2POp(dist.irecv, recv_hi, self.rank + 1, self.group)]
      for req in dist.batch_isend_irecv(ops):
        req.wait()
    if self.stage:
      recv_lo = recv_lo.to(dev_of) if has_lo else None
      recv_hi = recv_hi.to(dev_of) if has_hi else None
    if has_lo:
      first.set_halo('lo', recv_lo)
    if has_hi:
      last.set_halo('hi', recv_hi)

  def gather_sums(self, bands):
    """Every band's `sums` = the my_sums rows of ALL bands in global order."""
    import torch
    local = torch.stack([b.my_sums for b in bands]).contiguous()
    if self.world == 1:
      full = local
    elif self.comm is not None:
      full = self.comm.allgather(local)
    else:
      src = local.cpu() if self.stage else local
      parts = [torch.empty_like(src) for _ in range(self.world)]
      dist.all_gather(parts, src, group=self.group)
      full = torch.cat(parts).to(local.device)
    for b in bands:
      b.sums.copy_(full)

  def gather_stats(self, local: list) -> list:
    """Chunk statistics of all bands, in global band order (host objects)."""
    return [s for part in gather_objects(local, self.group) for s in part]


class RcclComm:
  """The library's RCCL communicator (sfm_comm_*), bootstrapped over the
  torch.distributed store: rank 0 creates the id, everyone joins."""

  def __init__(self, group=None):
    import ctypes as C
    from . import _abi
    self._C, self._abi = C, _abi
    self.lib = _abi.load()
    self.rank, self.world = world(group)
    ident = [None]
    if self.rank == 0:
      buf = C.create_string_buffer(_abi.COMM_ID_BYTES)
      _abi.check(self.lib.sfm_comm_unique_id(buf))
      ident = [bytes(buf.raw)]
    if not _alone(self.world):
      dist.broadcast_object_list(ident, src=0, group=group)
    handle = C.c_void_p()
    _abi.check(self.lib.sfm_comm_init(C.byref(handle), ident[0], self.rank,
                                      self.world))
    self.handle = handle

  def close(self):
    if self.handle:
      self._abi.check(self.lib.sfm_comm_destroy(self.handle))
      self.handle = None

  @staticmethod
  def _ptr(t):
    return None if t is None else t.data_ptr()

  def halo_exchange(self, peer_lo, send_lo, recv_lo, peer_hi, send_hi, recv_hi):
    from . import _dev
    ref = send_lo if send_lo is not None else send_hi
    count = 0 if ref is None else ref.numel()
    self._abi.check(self.lib.sfm_comm_halo_exchange(
        self.handle, peer_lo, self._ptr(send_lo), self._ptr(recv_lo), peer_hi,
        self._ptr(send_hi), self._ptr(recv_hi), count, _dev.stream_ptr()))

  def allgather(self, local):
    import torch
    from . import _dev
    out = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype,
                      device=local.device)
    self._abi.check(self.lib.sfm_comm_allgather(
        self.handle, local.data_ptr(), out.data_ptr(), local.numel(),
        _dev.stream_ptr()))
    return out.reshape((-1,) + tuple(local.shape[1:]))

  def allreduce(self, t, op='sum'):
    from . import _dev
    self._abi.check(self.lib.sfm_comm_allreduce_scalars(
        self.handle, t.data_ptr(), t.numel(),
        self._abi.REDUCE_SUM if op == 'sum' else self._abi.REDUCE_MAX,
        _dev.stream_ptr()))
    return t


class HostStagedTransport:
  """The host_halo / host_allgather pair of `SfmBandedDesc` over a
  torch.distributed group that moves HOST tensors (gloo): the library stages
  the packed edge rows and the bands' partial sums through host memory and
  calls back here, so the inter-rank branch of `sfm_mesh_relax_banded` runs
  where RCCL cannot connect the ranks (two processes sharing one GPU).

  A callback that raises records the exception in `.error` and makes the C loop
  return an error on THIS rank only; the neighbours stay in their recv /
  all-gather until the process group's timeout fires, so create the group with
  a finite `timeout=` (torch.distributed.new_group / init_process_group)."""

  def __init__(self, group=None):
    from . import _abi
    self.group = group
    self.rank, self.world = world(group)
    self.error = None
    self.calls = {'halo': 0, 'allgather': 0}
    self.halo = _abi.HOST_HALO_FN(self._halo)
    self.allgather = _abi.HOST_ALLGATHER_FN(self._allgather)

  @staticmethod
  def _tensor(ptr, n):
    import torch
    return torch.from_numpy(np.ctypeslib.as_array(ptr, shape=(int(n),)))

  def _halo(self, user, peer_lo, send_lo, recv_lo, peer_hi, send_hi, recv_hi, count):
    try:
      ops = []
      for peer, snd, rcv in ((peer_lo, send_lo, recv_lo), (peer_hi, send_hi, recv_hi)):
        if peer < 0:
          continue
        # the library hands over GROUP ranks; P2POp takes global ranks
        gpeer = peer if self.group is None else dist.get_global_rank(self.group, peer)
        ops += [dist.P2POp(dist.isend, self._tensor(snd, count), gpeer, self.group),
                dist.P2POp(dist.irecv, self._tensor(rcv, count), gpeer, self.group)]
      for req in dist.batch_isend_irecv(ops):
        req.wait()
      self.calls['halo'] += 1
      return 0
    except BaseException as e:      # never unwind through the C frames
      self.error = e
      return 1

  def _allgather(self, user, send, recv, count):
    try:
      src = self._tensor(send, count).clone()     # `send` lies inside `recv`
      parts = list(self._tensor(recv, count * self.world).split(int(count)))
      dist.all_gather(parts, src, group=self.group)
      self.calls['allgather'] += 1
      return 0
    except BaseException as e:
      self.error = e
      return 1


def relax_mesh_sharded(x, prev, config, mesh_force=None, group=None,
                       bands_per_rank: int = 1, band_factory=None,
                       transport: BandTransport | None = None):
  """`mesh.relax_mesh(x, prev, config)` for ONE mesh spread over the ranks.

  The rows (y) of every section are split into world_size * bands_per_rank
  bands; a rank relaxes its bands and exchanges, per step, the boundary rows
  with the neighbouring bands plus 7 floats of partial sums per band (FIRE
  power, drift means), which every band reduces in the same order -- all ranks
  take the same FIRE branches.  Chunk logic (force-cap schedule, stopping) is
  the reference's (mesh.py:570-606) on the global statistics.

  Every rank passes the full arrays and gets the full relaxed mesh back:
  (x [np.ndarray], e_kin history, steps).  `band_factory(x, prev, config, spec,
  own, global_nodes, n_bands)` replaces the HIP band (CPU tests).
  """
  rank, ws = world(group)
  x = np.asarray(x, dtype=np.float32)
  prev = None if prev is None else np.asarray(prev, dtype=np.float32)
  if config.start_cap != config.final_cap:
    if not config.fire:
      raise NotImplementedError(
          'Adaptive force capping is only supported with FIRE.')
    if config.cap_scale <= 1:
      raise ValueError(
          'The scaling factor for the force cap has to be larger '
          'than 1 when the initial and final cap are different.')
  if config.remove_drift and x.ndim == 5:
    raise NotImplementedError('per-column drift removal is not sharded')
  spec = None
  if band_factory is None:
    from . import mesh
    spec = mesh._resolve_force(mesh.inplane_force if mesh_force is None else mesh_force)
    band_factory = HipBand
  n_bands = ws * bands_per_rank
  bounds = band_bounds(x.shape[-2], n_bands)
  global_nodes = int(np.prod(x.shape[1:]))
  bands = []
  for i in range(bands_per_rank):
    g = rank * bands_per_rank + i
    y0, y1 = bounds[g]
    lo = y0 - (1 if g > 0 else 0)
    hi = y1 + (1 if g < n_bands - 1 else 0)
    bands.append(band_factory(
        x[..., lo:hi, :], None if prev is None else prev[..., lo:hi, :], config,
        spec, (y0 - lo, y1 - lo), global_nodes, n_bands))
  transport = transport or BandTransport(group)

  t = 0
  dt, alpha, cap = config.dt, config.alpha, config.start_cap
  e_kin = []
  while t < config.max_iters:
    for b in bands:
      b.begin(dt, alpha, cap)
    for k in range(config.num_iters):
      transport.exchange(bands)
      if config.fire and k > 0:
        transport.gather_sums(bands)
      for b in bands:
        b.advance()
      for b in bands:
        b.integrate()
    if config.fire and config.num_iters > 0:
      transport.gather_sums(bands)
    stats = transport.gather_stats([b.finish() for b in bands])
    t += config.num_iters
    ek = np.float32(0)
    for s in stats:                      # fixed (global band) order
      ek = np.float32(ek + s[4])
    e_kin.append(float(ek))
    v_max = float(max(s[5] for s in stats))
    if config.fire:
      dt, alpha, _, cap = stats[0][:4]
    if v_max < config.stop_v_max:
      if np.float32(cap) >= np.float32(config.final_cap):
        break
      cap = min(cap * config.cap_scale, config.final_cap)

  owned = transport.gather_stats([b.owned_x() for b in bands])
  return np.concatenate(owned, axis=-2), e_kin, t


_COMM_STREAMS = {}


def _comm_stream(dev):
  """One side stream per device for the exchange leg of the banded step."""
  import torch
  key = torch.device(dev).index
  if key not in _COMM_STREAMS:
    _COMM_STREAMS[key] = torch.cuda.Stream(device=dev)
  return _COMM_STREAMS[key]


def relax_mesh_banded(x, prev, config, mesh_force=None, group=None,
                      bands_per_rank: int = 1, comm: 'RcclComm | None' = None,
                      loopback: bool = False, overlap: bool = True, timing=None,
                      transport: str = 'auto'):
  """`mesh.relax_mesh(x, prev, config)` for ONE mesh spread over the ranks, with
  the step loop inside the library (sfm_mesh_relax_banded).

  Same split as `relax_mesh_sharded` -- world_size * bands_per_rank bands of
  rows, one halo row per neighbour -- but a chunk of `config.num_iters` steps is
  ONE C call: edge rows move between local bands by device copies and between
  ranks through the library's RCCL communicator (`comm`, created over the
  process group when there is more than one rank), the bands' partial sums are
  all-gathered once per step, and the exchange of a step's edge rows overlaps
  the integration of its interior rows on a second stream (`overlap`).
  `loopback=True` sends the edges between LOCAL bands through RCCL self
  send / recv as well (exercises the transport on one GPU).  `transport`:
  'rccl' (one GPU per rank), 'host' (the library stages rows and sums through
  host memory and `HostStagedTransport` moves them over the process group --
  ranks that share a GPU, gloo groups) or 'auto' = 'rccl' on an nccl group,
  'host' otherwise.

  Scope: meshes with a fixed `prev`.  A montage (`prev_fn`: the target of a
  tile depends on its neighbour TILES, stitch_elastic.py:624-676) does not
  split into bands of rows; montages and volumetric 5-D states with per-column
  drift removal scale as replicas.

  Every rank passes the full arrays and gets the full relaxed mesh back:
  (x [np.ndarray], e_kin history, steps).  `timing`, a dict, receives
  `banded_chunk_s`: seconds inside the C calls.
  """
  import ctypes as C
  import time
  import torch
  from . import _abi, _dev, mesh
  rank, ws = world(group)
  x = np.asarray(x, dtype=np.float32)
  prev = None if prev is None else np.asarray(prev, dtype=np.float32)
  if config.start_cap != config.final_cap:
    if not config.fire:
      raise NotImplementedError(
          'Adaptive force capping is only supported with FIRE.')
    if config.cap_scale <= 1:
      raise ValueError(
          'The scaling factor for the force cap has to be larger '
          'than 1 when the initial and final cap are different.')
  if config.remove_drift and x.ndim == 5:
    raise NotImplementedError('per-column drift removal is not sharded')
  spec = mesh._resolve_force(mesh.inplane_force if mesh_force is None else mesh_force)
  if spec.kind == _abi.FORCE_EXTERNAL:
    raise NotImplementedError('banded meshes need a native mesh_force')
  dev = _dev.device()
  lib = _abi.load()
  own_comm = None
  host = None
  if transport not in ('auto', 'rccl', 'host'):
    raise ValueError(f'unknown transport {transport!r}')
  if transport == 'auto':
    transport = 'rccl' if (comm is not None or ws == 1 or
                           dist.get_backend(group) == 'nccl') else 'host'
  if transport == 'host':
    if comm is not None or loopback:
      raise ValueError('the host-staged transport takes no RCCL communicator')
    if ws > 1:
      host = HostStagedTransport(group)
  elif comm is None and (ws > 1 or loopback):
    # (a forced world of one gathers through collectives, but only `loopback` asks for
    # RCCL self send / recv between its local bands: no communicator otherwise)
    comm = own_comm = RcclComm(group)
  n_local = int(bands_per_rank)
  n_bands = ws * n_local
  bounds = band_bounds(x.shape[-2], n_bands)
  global_nodes = int(np.prod(x.shape[1:]))

  descs = (_abi.SfmMeshDesc * n_local)()
  shards = (_abi.SfmMeshShard * n_local)()
  keep = []
  owns = []
  for i in range(n_local):
    g = rank * n_local + i
    y0, y1 = bounds[g]
    lo = y0 - (1 if g > 0 else 0)
    hi = y1 + (1 if g < n_bands - 1 else 0)
    x_t = _dev.as_device_f32(x[..., lo:hi, :], dev, copy=True)
    v_t = torch.zeros_like(x_t)
    a_t = torch.empty_like(x_t)
    p_t = None if prev is None else _dev.as_device_f32(prev[..., lo:hi, :], dev, copy=True)
    probe = mesh._base_desc(x_t, spec, config.k, config.stride, config.prefer_orig_order)
    wsp = _dev.workspace(lib.sfm_mesh_workspace_bytes(C.byref(probe)), dev)
    descs[i] = mesh._chunk_desc(x_t, v_t, a_t, p_t, config, spec, wsp)
    shards[i].own_y0, shards[i].own_y1 = y0 - lo, y1 - lo
    shards[i].global_nodes = global_nodes
    keep.append((x_t, v_t, a_t, p_t, wsp))
    owns.append((y0 - lo, y1 - lo))

  bd = _abi.SfmBandedDesc()
  bd.n_local = n_local
  bd.bands = descs
  bd.shards = shards
  bd.comm = comm.handle if comm is not None else None
  bd.rank, bd.n_ranks = rank, ws
  bd.flags = (_abi.BANDED_LOOPBACK if loopback else 0) | (
      0 if overlap else _abi.BANDED_NO_OVERLAP)
  side = _comm_stream(dev) if overlap else None
  bd.comm_stream = side.cuda_stream if side is not None else None
  if host is not None:
    bd.host_halo, bd.host_allgather = host.halo, host.allgather
  scratch = _dev.workspace(lib.sfm_mesh_banded_scratch_bytes(C.byref(bd)), dev)
  bd.scratch = scratch.data_ptr()
  bd.scratch_bytes = scratch.numel()

  t = 0
  dt, alpha, cap = config.dt, config.alpha, config.start_cap
  e_kin = []
  spent = 0.0
  try:
    while t < config.max_iters:
      fire = _abi.SfmFireState()
      fire.dt, fire.alpha, fire.n_pos, fire.cap = (
          np.float32(dt), np.float32(alpha), 0, np.float32(cap))
      stats = _abi.SfmChunkStats()
      for i in range(n_local):
        descs[i].stream = _dev.stream_ptr()
      t0 = time.perf_counter()
      rc = lib.sfm_mesh_relax_banded(C.byref(bd), C.byref(fire), C.byref(stats))
      if host is not None and host.error is not None:
        raise host.error
      _abi.check(rc)
      spent += time.perf_counter() - t0
      t += config.num_iters
      e_kin.append(float(stats.e_kin))
      v_max = float(stats.v_max)
      if config.fire:
        dt, alpha, cap = (np.float32(fire.dt), np.float32(fire.alpha),
                          np.float32(fire.cap))
      if v_max < config.stop_v_max:
        if np.float32(cap) >= np.float32(config.final_cap):
          break
        cap = min(cap * config.cap_scale, config.final_cap)
  finally:
    if own_comm is not None:
      torch.cuda.synchronize(dev)
      own_comm.close()
  if timing is not None:
    timing['banded_chunk_s'] = spent
    if host is not None:
      timing['host_calls'] = dict(host.calls)
  local = [k[0][..., o[0]:o[1], :].cpu().numpy() for k, o in zip(keep, owns)]
  owned = [s for part in gather_objects(local, group) for s in part]
  return np.concatenate(owned, axis=-2), e_kin, t


# ---------------------------------------------------------------------------
# Section alignment in blocks (BASELINE configs[3]; em_alignment notebook,
# cells 25 and 38-48): sections depend on the previous solved section, so the
# z axis is split into blocks that are solved independently, one (or more) per
# rank; the LAST solved mesh of every block is the block's boundary and is sent
# to all ranks, where it forms the "cross-block" flow field that a second, much
# smaller relaxation (one virtual section per block) aligns.
# ---------------------------------------------------------------------------
def block_ranges(n_sections: int, n_blocks: int) -> list[tuple[int, int]]:
  """[start, stop) section ranges of the blocks (consecutive blocks share the
  boundary section like the notebook's [0..50], [50..100], ...)."""
  if n_blocks < 1 or n_sections < n_blocks:
    raise ValueError(f'cannot split {n_sections} sections into {n_blocks} blocks')
  edges = [n_sections * i // n_blocks for i in range(n_blocks + 1)]
  return list(zip(edges[:-1], edges[1:]))


def solve_section_block(flow, config, stride, relax_fn=None, compose_fn=None,
                        with_last=False):
  """Sequential section-by-section relaxation of one block (notebook cell 25):
  prev = compose_maps_fast(flow[z], solved[-1]); x = relax_mesh(0, prev).

  flow: [2, n, y, x] cleaned flow of the block's sections.  Returns the solved
  meshes [2, n + 1, y, x] (entry 0 is the zero mesh of the block's first
  section).  The state stays on the device between the two ops.  `with_last`:
  also returns the last solved mesh as `relax_fn` produced it (a DeviceArray on
  the HIP path), for a hand-off that never leaves the device.
  """
  if relax_fn is None or compose_fn is None:
    from . import map_utils, mesh
    relax_fn = relax_fn or mesh.relax_mesh
    compose_fn = compose_fn or map_utils.compose_maps_fast
  origin = (0.0, 0.0)
  flow = np.asarray(flow, dtype=np.float32)
  zero = np.zeros_like(flow[:, 0:1])
  solved = [zero]
  for z in range(flow.shape[1]):
    # solved[-1] is whatever relax_fn returned (a DeviceArray on the HIP path):
    # no host round trip between relaxation and composition
    prev = compose_fn(flow[:, z:z + 1], origin, stride, solved[-1], origin, stride)
    x, _, _ = relax_fn(zero, prev, config)
    solved.append(x)
  out = np.concatenate([np.asarray(s, dtype=np.float32) for s in solved], axis=1)
  return (out, solved[-1]) if with_last else out


def gather_boundaries(last_local: list, n_blocks: int, group=None,
                      mesh_shape=None) -> np.ndarray:
  """The mesh-boundary hand-off of the block chain as ONE tensor all-gather.

  last_local: the last solved mesh [2, 1, y, x] of each of THIS rank's blocks
  (round-robin deal), DeviceArrays / device tensors on the HIP path.  On an nccl
  (= RCCL) group the meshes travel GPU to GPU over xGMI without touching the
  host; on a gloo group as host tensors.  Returns [2, n_blocks, y, x] (host),
  identical on every rank.  A rank without a block (n_blocks < world size)
  takes part with a zero-filled buffer: it needs `mesh_shape` = the shape
  [2, 1, y, x] of one boundary mesh; without it EVERY rank raises (decided from
  (n_blocks, world size) alone, so nobody is left alone in the collective).
  """
  import torch
  rank, ws = world(group)
  tensors = []
  for obj in last_local:
    t = getattr(obj, 'tensor', obj)
    if not isinstance(t, torch.Tensor):
      t = torch.from_numpy(np.ascontiguousarray(np.asarray(t, dtype=np.float32)))
    tensors.append(t.to(torch.float32))
  if _alone(ws):
    return np.concatenate([t.cpu().numpy() for t in tensors], axis=1)
  if n_blocks < ws and mesh_shape is None:
    raise ValueError(f'{n_blocks} blocks for {ws} ranks: ranks without a block need '
                     '`mesh_shape` to take part in the hand-off')
  on_gpu = dist.get_backend(group) == 'nccl'
  per = -(-n_blocks // ws)                      # blocks per rank, padded
  if tensors:
    shape, device = tuple(tensors[0].shape), tensors[0].device
  else:
    shape, device = tuple(int(v) for v in mesh_shape), torch.device('cpu')
  if on_gpu and device.type != 'cuda':
    device = torch.device('cuda', torch.cuda.current_device())
  send = torch.zeros((per,) + shape, dtype=torch.float32,
                     device=device if on_gpu else 'cpu')
  for k, t in enumerate(tensors):
    send[k].copy_(t)                            # device to device on the HIP path
  parts = [torch.empty_like(send) for _ in range(ws)]
  dist.all_gather(parts, send, group=group)
  last = [None] * n_blocks
  for r in range(ws):
    host = parts[r].cpu().numpy()
    for k, b in enumerate(shard_units(n_blocks, r, ws)):
      last[b] = host[k]
  return np.concatenate(last, axis=1)


def align_sections_blocked(flow, config, stride, n_blocks=None, group=None,
                           xblk_config=None, relax_fn=None, compose_fn=None,
                           timing=None):
  """Block-parallel section alignment.

  flow: [2, n_sections, y, x] (every rank passes the same array, or at least
  its own blocks' sections).  Blocks are dealt round-robin to the ranks and
  solved independently; the last-section mesh of every block ([2, 1, y, x],
  336 KB for an 8192^2 section at stride 40) is the only data that crosses
  ranks (`gather_boundaries`: one all-gather, device resident on RCCL).
  Returns (blocks: {block index: [2, n_b + 1, y, x]} of THIS rank,
  last: [2, n_blocks, y, x] boundary meshes of all blocks, xblk: the solved
  cross-block mesh [2, n_blocks, y, x], identical on every rank).  `timing`, a
  dict, receives the seconds of the three phases (`solve_s`, `handoff_s`,
  `xblk_s`; the hand-off includes waiting for the slowest rank).
  """
  import time
  rank, ws = world(group)
  flow = np.asarray(flow, dtype=np.float32)
  n_blocks = ws if n_blocks is None else n_blocks
  ranges = block_ranges(flow.shape[1], n_blocks)
  mine = shard_units(n_blocks, rank, ws)

  def sync():
    try:
      import torch
      if torch.cuda.is_available():
        torch.cuda.synchronize()
    except ImportError:
      pass

  t0 = time.perf_counter()
  blocks, last_local = {}, []
  for b in mine:
    blocks[b], last_obj = solve_section_block(
        flow[:, ranges[b][0]:ranges[b][1]], config, stride, relax_fn, compose_fn,
        with_last=True)
    last_local.append(last_obj)
  if timing is not None:
    sync()
  t1 = time.perf_counter()
  # mesh-boundary exchange: last solved section of every block, to every rank
  last = gather_boundaries(last_local, n_blocks, group,
                           mesh_shape=(flow.shape[0], 1) + tuple(flow.shape[2:]))
  t2 = time.perf_counter()
  # cross-block mesh: every block is a virtual section (notebook cell 47); it
  # is tiny, so every rank solves it redundantly instead of broadcasting it
  xcfg = xblk_config or config
  if relax_fn is None or compose_fn is None:
    from . import map_utils, mesh
    relax_fn = relax_fn or mesh.relax_mesh
    compose_fn = compose_fn or map_utils.compose_maps_fast
  xblk = []
  origin = (0.0, 0.0)
  for z in range(n_blocks):
    if z == 0:
      prev = last[:, z:z + 1]
    else:
      prev = compose_fn(last[:, z:z + 1], origin, stride, xblk[-1], origin, stride)
    x, _, _ = relax_fn(np.zeros_like(last[:, 0:1]), prev, xcfg)
    xblk.append(np.array(x, dtype=np.float32))
  if timing is not None:
    sync()
    timing.update(solve_s=t1 - t0, handoff_s=t2 - t1, xblk_s=time.perf_counter() - t2)
  return blocks, last, np.concatenate(xblk, axis=1)
