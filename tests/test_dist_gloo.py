"""N > 1 path on CPU: world_size 2, gloo.  The device call is replaced by the
CPU oracle so the test checks the sharding / gather logic: the sharded result
must be identical to the single-process one (batch membership preserved)."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import flow_oracle as fo


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _pair():
  from scipy import ndimage
  rng = np.random.default_rng(21)
  base = ndimage.gaussian_filter(rng.standard_normal((220, 200)), 2.0)
  base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
  return base[10:202, 10:170].copy(), base[13:205, 5:165].copy()


def _worker(rank, world_size, port, out_dir):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world_size)
  from sofima_amd import dist as sdist, flow_field
  pre, post = _pair()
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()

  def batch_fn(pre_starts, post_starts):
    return fo.batched_xcorr_peaks(pre, post, None, None, (48, 48), pre_starts,
                                  None, 2, 0.5, 5, (48, 48), post_starts)

  field = sdist.sharded_flow_field(calc, pre, post, 48, 24, batch_size=8,
                                   batch_fn=batch_fn)
  units = list(range(7))
  mapped = sdist.map_units(units, lambda u: u * u + rank * 0)
  np.save(os.path.join(out_dir, f'field_{rank}.npy'), field)
  np.save(os.path.join(out_dir, f'mapped_{rank}.npy'), np.array(mapped))
  dist.barrier()
  dist.destroy_process_group()


def test_sharded_flow_field_matches_single_process(tmp_path):
  port = _free_port()
  mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  pre, post = _pair()
  want = fo.flow_field(pre, post, 48, 24, batch_size=8)
  for r in range(2):
    got = np.load(tmp_path / f'field_{r}.npy')
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(got[~np.isnan(got)], want[~np.isnan(want)])
    np.testing.assert_array_equal(np.load(tmp_path / f'mapped_{r}.npy'),
                                  np.arange(7)**2)


def test_shard_units_cover_everything_once():
  from sofima_amd import dist as sdist
  for n in (0, 1, 5, 40, 41):
    for ws in (1, 2, 3, 8):
      seen = sorted(i for r in range(ws) for i in sdist.shard_units(n, r, ws))
      assert seen == list(range(n))


# ---------------------------------------------------------------------------
# One mesh across ranks (bands + halo rows) and the block chain of configs[3].
# The HIP band is replaced by a NumPy band that follows the same split step
# (advance / integrate around the two exchange points), so the test covers the
# band bookkeeping, the halo exchange and the rank-ordered reduction of
# sofima_amd.dist; the result must follow the single-process oracle.
# ---------------------------------------------------------------------------
class NumpyBand:
  """CPU stand-in for sofima_amd.dist.HipBand (test infrastructure)."""

  def __init__(self, x, prev, config, spec, own, global_nodes, n_bands):
    import torch
    from oracle import mesh_oracle as mo
    self.mo, self.torch = mo, torch
    self.cfg = config
    self.own = own
    self.x = np.array(x, np.float32)
    self.v = np.zeros_like(self.x)
    self.a = np.zeros_like(self.x)
    self.prev = None if prev is None else np.array(prev, np.float32)
    self.n_global = np.float32(global_nodes)
    self.sums = torch.zeros((n_bands, 8), dtype=torch.float32)
    self.my_sums = torch.zeros((8,), dtype=torch.float32)

  def _force(self, cap):
    f32 = np.float32
    a = self.mo.inplane_force(self.x, self.cfg.k, self.cfg.stride,
                              self.cfg.prefer_orig_order)
    if self.prev is not None:
      pull = f32(-self.cfg.k0) * np.nan_to_num(self.x - self.prev)
      a = a + np.clip(pull, -f32(cap), f32(cap))
    return a.astype(f32)

  def begin(self, dt, alpha, cap):
    f32 = np.float32
    self.dt, self.alpha, self.cap, self.n_pos = f32(dt), f32(alpha), f32(cap), 0
    self.phase = 0
    self.a = self._force(self.cap)

  def _pending(self):
    """FIRE bookkeeping of the previous step from the gathered sums."""
    f32 = np.float32
    cfg = self.cfg
    tot = np.zeros(8, f32)
    for row in self.sums.numpy():          # rank order
      tot = (tot + row).astype(f32)
    downhill = tot[0] >= 0
    self.n_pos = self.n_pos + 1 if downhill else 0
    if downhill:
      if self.n_pos > cfg.n_min:
        self.dt = min(self.dt * f32(cfg.f_inc), f32(float(cfg.dt_max) * float(cfg.dt)))
        self.alpha = self.alpha * f32(cfg.f_alpha)
      if self.n_pos > 0 and self.n_pos % cfg.cap_upscale_every == 0:
        self.cap = f32(cfg.cap_scale) * self.cap
    else:
      self.dt = self.dt * f32(cfg.f_dec)
      self.alpha = f32(cfg.alpha)
      self.v = self.v * f32(0)
    self.cap = min(self.cap, f32(cfg.final_cap))
    if cfg.remove_drift:
      gate = f32(1 if downhill else 0)
      shape = (-1,) + (1,) * (self.x.ndim - 1)
      self.x = self.x - (tot[1:1 + self.x.shape[0]] / self.n_global).reshape(shape)
      self.v = self.v - ((tot[4:4 + self.x.shape[0]] / self.n_global) * gate).reshape(shape)

  def advance(self):
    f32 = np.float32
    if self.cfg.fire and self.phase > 0:
      self._pending()
    dt = self.dt if self.cfg.fire else f32(self.cfg.dt)
    self.x = (self.x + (dt * self.v + f32(0.5) * (dt * dt) * self.a)).astype(f32)
    self.phase += 1

  def integrate(self):
    f32 = np.float32
    cfg = self.cfg
    dt = self.dt if cfg.fire else f32(cfg.dt)
    f = self._force(self.cap)
    g = f32(cfg.gamma)
    hdtg = f32(0.5) * dt * g
    vn = (f32(1) / (f32(1) + hdtg)) * (self.v * (f32(1) - hdtg) +
                                       f32(0.5) * dt * (self.a + f))
    self.a = f
    if cfg.fire:
      sl = (slice(None),) * (self.x.ndim - 2) + (slice(*self.own),)
      power = np.sum((f * vn)[sl], dtype=f32)
      a_n = np.sqrt(np.sum(np.square(f), axis=0, keepdims=True)) + f32(1e-6)
      v_n = np.sqrt(np.sum(np.square(vn), axis=0, keepdims=True))
      vn = vn + self.alpha * (f / a_n * v_n - vn)
      row = np.zeros(8, f32)
      row[0] = power
      c = self.x.shape[0]
      axes = tuple(range(1, self.x.ndim))
      row[1:1 + c] = np.sum(self.x[sl], axis=axes, dtype=f32)
      row[4:4 + c] = np.sum(vn[sl], axis=axes, dtype=f32)
      self.my_sums.copy_(self.torch.from_numpy(row))
    self.v = vn.astype(f32)

  def finish(self):
    f32 = np.float32
    if self.cfg.fire and self.phase > 0:
      self._pending()
    sl = (slice(None),) * (self.x.ndim - 2) + (slice(*self.own),)
    speed2 = np.sum(np.square(self.v[sl]), axis=0)
    return (f32(self.dt), f32(self.alpha), int(self.n_pos), f32(self.cap),
            f32(np.sum(speed2)), f32(np.sqrt(speed2.max())))

  def boundary(self, side):
    row = self.own[0] if side == 'lo' else self.own[1] - 1
    return self.torch.from_numpy(np.stack(
        [t[..., row, :] for t in (self.x, self.v, self.a)]).copy())

  def set_halo(self, side, packed):
    row = self.own[0] - 1 if side == 'lo' else self.own[1]
    p = packed.numpy()
    self.x[..., row, :], self.v[..., row, :], self.a[..., row, :] = p[0], p[1], p[2]

  def owned_x(self):
    return self.x[..., self.own[0]:self.own[1], :].copy()


def _mesh_case(remove_drift):
  import types
  from scipy import ndimage
  rng = np.random.default_rng(4)
  shape = (2, 2, 23, 17)
  prev = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 2, 2)) * 25
  prev = prev.astype(np.float32)
  prev[:, 0, :2] = np.nan
  prev[:, 1, 10:12, 5:9] = np.nan
  cfg = types.SimpleNamespace(
      dt=0.001, gamma=0.0, k0=0.05, k=0.1, stride=(10.0, 10.0), num_iters=40,
      max_iters=120, stop_v_max=1e-9, fire=True, f_alpha=0.99, f_inc=1.1, f_dec=0.5,
      alpha=0.1, n_min=5, dt_max=100.0, start_cap=0.1, final_cap=10.0, cap_scale=1.1,
      cap_upscale_every=10, prefer_orig_order=True, remove_drift=remove_drift)
  x0 = (rng.standard_normal(shape) * 0.3).astype(np.float32)
  return x0, prev, cfg


def _band_worker(rank, world_size, port, out_dir):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world_size)
  from oracle import maps_oracle, mesh_oracle
  from sofima_amd import dist as sdist
  for drift in (False, True):
    x0, prev, cfg = _mesh_case(drift)
    for per_rank in (1, 2):
      x, ek, t = sdist.relax_mesh_sharded(x0, prev, cfg, bands_per_rank=per_rank,
                                          band_factory=NumpyBand)
      np.savez(os.path.join(out_dir, f'mesh_{int(drift)}_{per_rank}_{rank}.npz'),
               x=x, ek=np.array(ek), t=t)
  # block chain (configs[3]): 6 sections in 3 blocks over 2 ranks
  flow, cfg = _chain_case()
  blocks, last, xblk = sdist.align_sections_blocked(
      flow, cfg, 10.0, n_blocks=3, relax_fn=mesh_oracle.relax_mesh,
      compose_fn=maps_oracle.compose_maps_fast)
  np.savez(os.path.join(out_dir, f'chain_{rank}.npz'), last=last, xblk=xblk,
           **{f'block{b}': v for b, v in blocks.items()})
  # fewer blocks than ranks: rank 1 holds no block and must still take part in
  # the hand-off (a lopsided raise would leave rank 0 in all_gather)
  blocks1, last1, xblk1 = sdist.align_sections_blocked(
      flow, cfg, 10.0, n_blocks=1, relax_fn=mesh_oracle.relax_mesh,
      compose_fn=maps_oracle.compose_maps_fast)
  np.savez(os.path.join(out_dir, f'chain1_{rank}.npz'), last=last1, xblk=xblk1,
           n_mine=len(blocks1))
  # and without a mesh shape EVERY rank raises before the collective
  try:
    sdist.gather_boundaries([last1[:, :1]] if rank == 0 else [], 1)
    raised = False
  except ValueError:
    raised = True
  np.save(os.path.join(out_dir, f'raised_{rank}.npy'), np.array(raised))
  dist.barrier()
  dist.destroy_process_group()


def _chain_case():
  import types
  from scipy import ndimage
  rng = np.random.default_rng(6)
  flow = ndimage.gaussian_filter(rng.standard_normal((2, 6, 14, 15)), (0, 0, 2, 2)) * 12
  flow = flow.astype(np.float32)
  flow[:, 2, :2, :3] = np.nan
  cfg = types.SimpleNamespace(
      dt=0.001, gamma=0.0, k0=0.05, k=0.1, stride=(10.0, 10.0), num_iters=50,
      max_iters=150, stop_v_max=0.01, fire=True, f_alpha=0.99, f_inc=1.1, f_dec=0.5,
      alpha=0.1, n_min=5, dt_max=100.0, start_cap=1.0, final_cap=10.0, cap_scale=1.1,
      cap_upscale_every=10, prefer_orig_order=True, remove_drift=False)
  return flow, cfg


def test_band_sharded_mesh_follows_single_process(tmp_path):
  from oracle import mesh_oracle
  port = _free_port()
  mp.spawn(_band_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  for drift in (False, True):
    x0, prev, cfg = _mesh_case(drift)
    wx, we, wt = mesh_oracle.relax_mesh(x0, prev, cfg)
    for per_rank in (1, 2):
      got = [np.load(tmp_path / f'mesh_{int(drift)}_{per_rank}_{r}.npz')
             for r in range(2)]
      # both ranks hold the same full result
      np.testing.assert_array_equal(got[0]['x'], got[1]['x'])
      np.testing.assert_array_equal(got[0]['ek'], got[1]['ek'])
      assert int(got[0]['t']) == wt
      # and it follows the single-process trajectory (same FIRE branches: the
      # energy trace agrees chunk by chunk)
      np.testing.assert_allclose(got[0]['x'], wx, atol=2e-4 * np.abs(wx).max())
      np.testing.assert_allclose(got[0]['ek'], we, rtol=2e-3)


def test_block_chain_matches_single_process(tmp_path):
  from oracle import maps_oracle, mesh_oracle
  from sofima_amd import dist as sdist
  if not (tmp_path / 'chain_0.npz').exists():
    port = _free_port()
    mp.spawn(_band_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  flow, cfg = _chain_case()
  blocks, last, xblk = sdist.align_sections_blocked(
      flow, cfg, 10.0, n_blocks=3, relax_fn=mesh_oracle.relax_mesh,
      compose_fn=maps_oracle.compose_maps_fast)
  assert sorted(blocks) == [0, 1, 2] and last.shape == (2, 3, 14, 15)
  seen = {}
  for r in range(2):
    g = np.load(tmp_path / f'chain_{r}.npz')
    np.testing.assert_array_equal(g['last'], last)
    np.testing.assert_array_equal(g['xblk'], xblk)
    for k in g.files:
      if k.startswith('block'):
        seen[int(k[5:])] = g[k]
  assert sorted(seen) == [0, 1, 2]          # every block solved by exactly one rank
  for b in range(3):
    np.testing.assert_array_equal(seen[b], blocks[b])
    np.testing.assert_array_equal(seen[b][:, -1], last[:, b])


def test_block_chain_with_fewer_blocks_than_ranks(tmp_path):
  from oracle import maps_oracle, mesh_oracle
  from sofima_amd import dist as sdist
  if not (tmp_path / 'chain1_0.npz').exists():
    port = _free_port()
    mp.spawn(_band_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  flow, cfg = _chain_case()
  _, last, xblk = sdist.align_sections_blocked(
      flow, cfg, 10.0, n_blocks=1, relax_fn=mesh_oracle.relax_mesh,
      compose_fn=maps_oracle.compose_maps_fast)
  g = [np.load(tmp_path / f'chain1_{r}.npz') for r in range(2)]
  assert [int(v['n_mine']) for v in g] == [1, 0]
  for v in g:
    np.testing.assert_array_equal(v['last'], last)
    np.testing.assert_array_equal(v['xblk'], xblk)
  assert all(bool(np.load(tmp_path / f'raised_{r}.npy')) for r in range(2))


def test_band_bounds_and_block_ranges():
  from sofima_amd import dist as sdist
  assert sdist.band_bounds(10, 3) == [(0, 3), (3, 6), (6, 10)]
  assert sdist.block_ranges(64, 8)[0] == (0, 8) and sdist.block_ranges(64, 8)[-1] == (56, 64)
  import pytest
  with pytest.raises(ValueError):
    sdist.band_bounds(2, 3)
