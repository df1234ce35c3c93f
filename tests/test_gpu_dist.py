"""Multi-GPU building blocks on ONE GPU (-m gpu): the band-sharded mesh step
(sfm_mesh_shard_*) with several bands in one process, and the library's RCCL
entry points at world size 1 (self send / recv, all-gather, all-reduce)."""
import numpy as np
import pytest

from oracle import mesh_oracle

pytestmark = pytest.mark.gpu


def _case(shape, drift, fire=True, amp=30.0):
  from scipy import ndimage
  from sofima_amd import mesh
  rng = np.random.default_rng(7)
  prev = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 3, 3)) * amp
  prev = prev.astype(np.float32)
  prev[:, 0, :2] = np.nan
  prev[:, -1, 20:24, 5:9] = np.nan
  kw = dict(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40), num_iters=50,
            max_iters=150, stop_v_max=1e-9, dt_max=1000, start_cap=0.01, final_cap=10,
            prefer_orig_order=True, remove_drift=drift)
  if not fire:
    kw.update(fire=False, gamma=0.5, dt=0.05, start_cap=10.0, final_cap=10.0)
  x0 = (rng.standard_normal(shape) * 0.4).astype(np.float32)
  return x0, prev, mesh.IntegrationConfig(**kw)


@pytest.mark.parametrize('drift,fire', [(False, True), (True, True), (False, False)])
@pytest.mark.parametrize('n_bands', [1, 2, 5])
def test_banded_mesh_equals_whole_mesh(gpu, n_bands, drift, fire):
  """`n_bands` bands of one mesh (halo rows copied between them, partial sums
  reduced in band order) follow the un-split relaxation and the oracle."""
  from sofima_amd import dist as sdist, mesh
  x0, prev, cfg = _case((2, 2, 61, 47), drift, fire)
  gx, ge, gt = sdist.relax_mesh_sharded(x0, prev, cfg, bands_per_rank=n_bands)
  wx, we, wt = mesh_oracle.relax_mesh(x0, prev, cfg)
  sx, se, st = mesh.relax_mesh(x0, prev, cfg)
  assert gt == wt == st
  scale = np.abs(wx).max()
  np.testing.assert_allclose(gx, wx, atol=1e-3 * scale)
  np.testing.assert_allclose(ge, we, rtol=1e-3)
  np.testing.assert_allclose(gx, np.array(sx), atol=2e-4 * scale)


def test_banded_tile_mesh_force(gpu):
  """Bands also carry the tile-mesh force model (no z coupling, 4 neighbours)."""
  from sofima_amd import dist as sdist, mesh
  rng = np.random.default_rng(3)
  cx = (rng.standard_normal((2, 1, 9, 7)) * 30).astype(np.float32)
  cy = (rng.standard_normal((2, 1, 9, 7)) * 30).astype(np.float32)
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.0, k=0.1, stride=(1, 1),
                               num_iters=100, max_iters=300, stop_v_max=1e-9, dt_max=100)
  x0 = np.zeros_like(cx)

  class Bands:
    def __init__(self, y0, y1):
      self.force = mesh.TileMeshForce(cx[..., y0:y1, :], cy[..., y0:y1, :])

  # every band needs its own rows of cx / cy: build the bands by hand
  bounds = sdist.band_bounds(9, 3)
  specs = []
  for g, (y0, y1) in enumerate(bounds):
    lo, hi = y0 - (g > 0), y1 + (g < 2)
    specs.append(mesh._resolve_force(mesh.TileMeshForce(cx[..., lo:hi, :],
                                                        cy[..., lo:hi, :])))
  it = iter(specs)
  factory = lambda x, prev, config, spec, own, n, nb: sdist.HipBand(
      x, prev, config, next(it), own, n, nb)
  gx, ge, gt = sdist.relax_mesh_sharded(x0, None, cfg, bands_per_rank=3,
                                        band_factory=factory)
  sx, se, st = mesh.relax_mesh(x0, None, cfg, mesh_force=mesh.TileMeshForce(cx, cy))
  assert gt == st
  np.testing.assert_allclose(gx, np.array(sx), atol=2e-3)


def test_rccl_entry_points_world_size_one(gpu):
  """sfm_comm_*: communicator of one rank; all-gather / all-reduce are copies,
  the halo exchange is exercised as a send / recv pair to the rank itself."""
  import torch
  from sofima_amd import dist as sdist
  comm = sdist.RcclComm()
  try:
    assert comm.world == 1
    a = torch.arange(24, dtype=torch.float32, device=gpu).reshape(3, 8)
    out = comm.allgather(a)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), a.cpu().numpy())
    s = torch.tensor([1.5, -2.0, 7.0], device=gpu)
    comm.allreduce(s, 'sum')
    comm.allreduce(s, 'max')
    torch.cuda.synchronize()
    np.testing.assert_array_equal(s.cpu().numpy(), [1.5, -2.0, 7.0])
    lo = torch.rand(1000, device=gpu)
    hi = torch.rand(1000, device=gpu)
    rlo, rhi = torch.zeros_like(lo), torch.zeros_like(hi)
    comm.halo_exchange(0, lo, rlo, 0, hi, rhi)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rlo.cpu().numpy(), lo.cpu().numpy())
    np.testing.assert_array_equal(rhi.cpu().numpy(), hi.cpu().numpy())
    # no neighbours: nothing to do, no error
    comm.halo_exchange(-1, None, None, -1, None, None)
    # the band transport over this communicator (one band, one rank)
    x0, prev, cfg = _case((2, 1, 40, 33), True)
    tr = sdist.BandTransport(comm=comm)
    gx, ge, gt = sdist.relax_mesh_sharded(x0, prev, cfg, transport=tr)
    wx, we, wt = mesh_oracle.relax_mesh(x0, prev, cfg)
    assert gt == wt
    np.testing.assert_allclose(gx, wx, atol=1e-3 * np.abs(wx).max())
  finally:
    comm.close()


@pytest.mark.parametrize('shape,n_blocks,max_iters', [((2, 6, 30, 34), 3, 1000),
                                                      ((2, 8, 201, 201), 4, 300)])
def test_block_chain_on_device_vs_oracle(gpu, shape, n_blocks, max_iters):
  """configs[3] block chain with the HIP relax / compose ops vs the oracle: a
  small chain, and eight sections of the configs[1] field size (201 x 201
  vectors) in four blocks."""
  from oracle import maps_oracle
  from scipy import ndimage
  from sofima_amd import dist as sdist, mesh
  rng = np.random.default_rng(6)
  big = shape[-1] > 100
  sigma = 12 if big else 3
  flow = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, sigma, sigma))
  flow = (flow / np.abs(flow).max() * 4.0).astype(np.float32)
  flow[:, 2, :2, :3] = np.nan
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.05, k=0.1, stride=(40, 40),
                               num_iters=100, max_iters=max_iters, stop_v_max=0.005,
                               dt_max=1000, start_cap=0.1, final_cap=10,
                               prefer_orig_order=True)
  blocks, last, xblk = sdist.align_sections_blocked(flow, cfg, 40.0, n_blocks=n_blocks)
  wb, wl, wx = sdist.align_sections_blocked(
      flow, cfg, 40.0, n_blocks=n_blocks, relax_fn=mesh_oracle.relax_mesh,
      compose_fn=maps_oracle.compose_maps_fast)
  np.testing.assert_allclose(last, wl, atol=2e-3)
  np.testing.assert_allclose(xblk, wx, atol=2e-3)
  for b in range(n_blocks):
    np.testing.assert_array_equal(np.isnan(blocks[b]), np.isnan(wb[b]))
    np.testing.assert_allclose(np.nan_to_num(blocks[b]), np.nan_to_num(wb[b]), atol=2e-3)


# ---------------------------------------------------------------------------
# The C-side step loop (sfm_mesh_relax_banded)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('drift,fire', [(False, True), (True, True), (False, False)])
@pytest.mark.parametrize('n_bands,mode', [(1, 'plain'), (2, 'plain'), (4, 'plain'),
                                          (3, 'serial'), (2, 'loopback'), (4, 'loopback')])
def test_banded_c_loop_fused_kernel(gpu, n_bands, mode, drift, fire):
  """One mesh wide enough for the fused tiled kernel (X >= 40), split into
  bands stepped by ONE C call per chunk: local copies / no-overlap single
  stream / RCCL self send-recv between the bands all follow the un-split
  relaxation (2e-4 of the displacement scale) and the oracle."""
  from sofima_amd import dist as sdist, mesh
  # (drift removal: a calmer target field, see the tolerance note below)
  x0, prev, cfg = _case((2, 3, 75, 70), drift, fire, amp=6.0 if drift else 30.0)
  gx, ge, gt = sdist.relax_mesh_banded(
      x0, prev, cfg, bands_per_rank=n_bands, loopback=(mode == 'loopback'),
      overlap=(mode != 'serial'))
  wx, we, wt = mesh_oracle.relax_mesh(x0, prev, cfg)
  # the un-split relaxation on the same (tiled, fused) kernel
  from sofima_amd import _abi
  with _abi.option('SFM_MESH_PERSISTENT', 0):
    sx, se, st = mesh.relax_mesh(x0, prev, cfg)
    # (the un-split step packs the tile rows of the last, 8-column wide tile column
    # into one workgroup; bands keep a workgroup per tile: other partial sums)
    with _abi.option('SFM_MESH_PACK', 0):
      ux, ue, ut = mesh.relax_mesh(x0, prev, cfg)
  assert gt == wt == st == ut
  scale = np.abs(wx).max()
  if n_bands == 1 or not drift:
    # One band: the same tiles, the same sums in the same order.  Several bands
    # without drift removal: only the SIGN of the power sum enters the step, so
    # the split changes nothing either -- bit for bit the un-split trajectory.
    np.testing.assert_array_equal(gx, np.array(ux))
    if not drift:
      np.testing.assert_array_equal(gx, np.array(sx))
    else:
      np.testing.assert_allclose(gx, np.array(sx), atol=2e-4 * scale)
    np.testing.assert_allclose(ge, ue, rtol=1e-6)
    np.testing.assert_allclose(ge, se, rtol=1e-6 if not drift else 1e-3)
    np.testing.assert_allclose(gx, wx, atol=1e-3 * scale)
  else:
    # Drift removal subtracts mean(x), mean(v): the band-ordered sums differ from
    # the tile-ordered ones in the last bit (2e-9 after one step), and the
    # relaxation amplifies that: 1.4e-7 of the scale after 50 steps, <= 1.3e-4
    # after 150 on this case (1.2e-3 with the 5x larger target field of the
    # other cases: e_kin 350 at step 50) -- tools/measure/banded_c_loop.py.
    np.testing.assert_allclose(gx, np.array(sx), atol=2e-4 * scale)
    np.testing.assert_allclose(ge, se, rtol=1e-3)
    np.testing.assert_allclose(gx, wx, atol=1e-3 * scale)
  np.testing.assert_allclose(ge, we, rtol=1e-3)


def test_banded_c_loop_modes_are_bit_identical(gpu):
  """Overlapped (edge tiles first, exchange on the second stream), serial and
  loop-back runs of the same split compute the same numbers."""
  from sofima_amd import dist as sdist
  x0, prev, cfg = _case((2, 2, 150, 64), True)
  a = sdist.relax_mesh_banded(x0, prev, cfg, bands_per_rank=3)
  b = sdist.relax_mesh_banded(x0, prev, cfg, bands_per_rank=3, overlap=False)
  c = sdist.relax_mesh_banded(x0, prev, cfg, bands_per_rank=3, loopback=True)
  np.testing.assert_array_equal(a[0], b[0])
  np.testing.assert_array_equal(a[0], c[0])
  assert a[1] == b[1] == c[1] and a[2] == b[2] == c[2]


@pytest.mark.parametrize('n_bands', [1, 3])
def test_banded_c_loop_two_launch_fallback(gpu, n_bands):
  """Meshes the fused kernel does not take (narrow in-plane mesh, volumetric
  mesh) run the advance / integrate pair inside the same C loop."""
  from sofima_amd import dist as sdist, mesh
  x0, prev, cfg = _case((2, 2, 61, 33), True)
  gx, ge, gt = sdist.relax_mesh_banded(x0, prev, cfg, bands_per_rank=n_bands)
  sx, se, st = mesh.relax_mesh(x0, prev, cfg)
  assert gt == st
  scale = np.abs(np.array(sx)).max()
  np.testing.assert_allclose(gx, np.array(sx), atol=2e-4 * scale)
  rng = np.random.default_rng(5)
  x3 = (rng.standard_normal((3, 6, 31, 12)) * 0.5).astype(np.float32)
  p3 = (rng.standard_normal((3, 6, 31, 12)) * 4).astype(np.float32)
  cfg3 = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(20, 20, 20),
                                num_iters=40, max_iters=80, stop_v_max=1e-9, dt_max=1000,
                                start_cap=0.01, final_cap=10)
  gx, ge, gt = sdist.relax_mesh_banded(x3, p3, cfg3, mesh_force=mesh.elastic_mesh_3d,
                                       bands_per_rank=n_bands)
  sx, se, st = mesh.relax_mesh(x3, p3, cfg3, mesh_force=mesh.elastic_mesh_3d)
  assert gt == st
  np.testing.assert_allclose(gx, np.array(sx), atol=2e-4 * np.abs(np.array(sx)).max())


def test_banded_c_loop_step_cost(gpu):
  """[2, 64, 204, 204] (the montage size): the banded step costs at most 1.3x
  (two bands) / 1.45x (four bands; measured 1.29x) the un-split step on the same
  GPU (the Python-driven split step was 5.7x)."""
  import time
  import torch
  from sofima_amd import dist as sdist, mesh
  rng = np.random.default_rng(11)
  shape = (2, 64, 204, 204)
  prev = (rng.standard_normal(shape) * 2).astype(np.float32)
  x0 = np.zeros(shape, np.float32)
  iters = 300
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(20., 20.),
                               num_iters=iters, max_iters=iters, stop_v_max=1e-9,
                               dt_max=1000, start_cap=0.01, final_cap=10,
                               prefer_orig_order=True)
  x_d = torch.from_numpy(x0).cuda()
  p_d = torch.from_numpy(prev).cuda()
  mesh.relax_mesh(x_d, p_d, cfg)
  torch.cuda.synchronize()
  whole = float('inf')
  for _ in range(3):   # best of three: the bound is on the step cost, not on a noisy box
    t0 = time.perf_counter()
    sx, _, _ = mesh.relax_mesh(x_d, p_d, cfg)
    torch.cuda.synchronize()
    whole = min(whole, (time.perf_counter() - t0) / iters)
  res = {}
  for nb in (2, 4):
    sdist.relax_mesh_banded(x0, prev, cfg, bands_per_rank=nb)
    res[nb] = float('inf')
    for _ in range(3):
      tm = {}
      gx, _, _ = sdist.relax_mesh_banded(x0, prev, cfg, bands_per_rank=nb, timing=tm)
      res[nb] = min(res[nb], tm['banded_chunk_s'] / iters)
    np.testing.assert_allclose(gx, np.array(sx), atol=2e-4 * np.abs(gx).max())
  print('un-split %.1f us/step; banded %s' % (
      whole * 1e6, {k: round(v * 1e6, 1) for k, v in res.items()}))
  # measured 1.11x (2 bands) and 1.29x (4 bands): the second bound leaves room for
  # box-to-box variation
  assert res[2] <= 1.3 * whole and res[4] <= 1.45 * whole, (whole, res)


def test_bench_runs_as_two_ranks(gpu):
  """`bench.py --gpus 2` end to end: spawn_ranks -> torch.distributed.run -> two
  ranks (gloo, sharing this GPU) -> barrier-bracketed timing, max over ranks,
  one JSON line from rank 0 with the whole-job aggregate."""
  import json
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, SFM_BENCH_BACKEND='gloo', SFM_BENCH_ONE_DEVICE='1')
  cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--size', '1024',
         '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--mesh-iters', '100',
         '--sustain', '0.5']
  out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout[-2000:]
  line = json.loads(lines[0])
  assert line['n_gpus'] == 2 and line['steps'] == 2 and line['warmup'] == 1
  assert line['scaling'] == 'weak' and line['unit'] == 'Mpix/s'
  # whole-job aggregate: 2 ranks x 1024^2 pixels x steps / max-over-ranks time
  want = 2 * 1024 * 1024 * 2 / (line['flow_ms_per_step'] * 2 * 1e-3) / 1e6
  assert abs(line['value'] - want) <= 1e-6 * want
  assert line['roofline'] and line['mesh']['value'] > 0
  assert line['sustained']['steps'] >= 1
  # the communicating legs are part of the default N > 1 line
  mg = line['multi_gpu']
  assert mg['ranks_seen'] == 2 and mg['backend'] == 'gloo'
  chain = mg['section_chain']
  assert chain['sections'] == 16 and chain['blocks'] == 2
  assert chain['sections_per_s'] > 0 and chain['handoff_only_us'] > 0
  assert chain['finite_fraction'] > 0.9
  assert chain['ms_flow'] > 0 and chain['mpix_s'] > 0     # flow_field + clean_flow per section
  vol = mg['volumetric_chunks']
  assert vol['tile_pairs'] == 8 and vol['tile_pairs_per_s'] > 0 and vol['finite_fraction'] > 0.9
  band = mg['mesh_sharded']
  assert band['ranks'] == 2 and band['banded_us_per_step'] > 0
  assert 'host-staged' in band['transport']


def test_bench_multi_gpu_legs_over_rccl_at_world_size_one(gpu):
  """First contact with RCCL before an 8-GPU box: `bench.py --gpus 1
  --force-multi-gpu-legs` runs every communicating leg over an nccl process
  group of ONE rank -- the device all-gather of the boundary meshes
  (`dist.gather_boundaries` on an nccl group), the library communicator
  bootstrapped from a broadcast unique id, RCCL send / recv + all-gather inside
  sfm_mesh_relax_banded (two local bands, loop-back), the all-reduces and the
  watchdog -- and prints the `multi_gpu` object."""
  import json
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ)
  env.pop('SFM_BENCH_BACKEND', None)
  cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--size', '1024',
         '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--mesh-iters', '100',
         '--sustain', '0', '--no-legs', '--force-multi-gpu-legs']
  out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout[-2000:]
  line = json.loads(lines[0])
  mg = line['multi_gpu']
  assert mg['backend'] == 'rccl' and mg['rccl_ranks'] == 1, mg
  for leg in ('section_chain', 'volumetric_chunks', 'mesh_sharded'):
    assert 'error' not in mg[leg], (leg, mg[leg])
  chain = mg['section_chain']
  assert chain['sections'] == 8 and chain['blocks'] == 1
  assert 'RCCL' in chain['handoff'] and chain['handoff_only_us'] > 0
  assert chain['finite_fraction'] > 0.9
  assert mg['volumetric_chunks']['tile_pairs'] == 4
  band = mg['mesh_sharded']
  assert band['bands_per_rank'] == 2 and 'RCCL' in band['transport']
  assert band['banded_us_per_step'] > 0


def _two_rank_band_worker(rank, world_size, port, out_dir):
  import os
  import torch
  import torch.distributed as dist
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world_size)
  torch.cuda.set_device(0)
  from sofima_amd import dist as sdist
  x0, prev, cfg = _case((2, 2, 61, 47), True)
  gx, ge, gt = sdist.relax_mesh_sharded(x0, prev, cfg, bands_per_rank=2)
  np.save(os.path.join(out_dir, f'x_{rank}.npy'), gx)
  np.save(os.path.join(out_dir, f'e_{rank}.npy'), np.array(ge + [gt]))
  dist.barrier()
  dist.destroy_process_group()


def test_two_ranks_with_hip_bands_share_one_gpu(gpu, tmp_path):
  """World size 2 with REAL HIP bands: two processes (gloo, rows staged through
  the host) hold two bands each of one mesh on this GPU; both return the mesh
  the single process with four bands returns, bit for bit."""
  import socket
  import torch.multiprocessing as mp
  from sofima_amd import dist as sdist
  sock = socket.socket()
  sock.bind(('127.0.0.1', 0))
  port = sock.getsockname()[1]
  sock.close()
  mp.spawn(_two_rank_band_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  x0, prev, cfg = _case((2, 2, 61, 47), True)
  wx, we, wt = sdist.relax_mesh_sharded(x0, prev, cfg, bands_per_rank=4)
  for r in range(2):
    np.testing.assert_array_equal(np.load(tmp_path / f'x_{r}.npy'), wx)
    np.testing.assert_array_equal(np.load(tmp_path / f'e_{r}.npy'), np.array(we + [wt]))


def _two_rank_c_loop_worker(rank, world_size, port, out_dir):
  import os
  import torch
  import torch.distributed as dist
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world_size)
  torch.cuda.set_device(0)
  from sofima_amd import dist as sdist, mesh
  rng = np.random.default_rng(5)
  x3 = (rng.standard_normal((3, 6, 31, 12)) * 0.5).astype(np.float32)
  p3 = (rng.standard_normal((3, 6, 31, 12)) * 4).astype(np.float32)
  cfg3 = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(20, 20, 20),
                                num_iters=40, max_iters=80, stop_v_max=1e-9, dt_max=1000,
                                start_cap=0.01, final_cap=10)
  cases = {
      # fused tiled kernel in band mode, edge-first split on the second stream
      'fused': (_case((2, 3, 75, 70), False), {}),
      'fused_drift': (_case((2, 3, 75, 70), True, amp=6.0), {}),
      'fused_serial': (_case((2, 2, 150, 64), True), {'overlap': False}),
      'verlet': (_case((2, 2, 61, 47), False, fire=False), {}),
      # advance / integrate pair inside the loop (narrow mesh, volumetric mesh)
      'narrow': (_case((2, 2, 61, 33), True), {}),
      'vol': ((x3, p3, cfg3), {'mesh_force': mesh.elastic_mesh_3d}),
  }
  for name, ((x0, prev, cfg), kw) in cases.items():
    timing = {}
    gx, ge, gt = sdist.relax_mesh_banded(x0, prev, cfg, bands_per_rank=2, transport='host',
                                         timing=timing, **kw)
    np.save(os.path.join(out_dir, f'{name}_x_{rank}.npy'), gx)
    np.save(os.path.join(out_dir, f'{name}_e_{rank}.npy'), np.array(ge + [gt]))
    calls = timing['host_calls']
    # one exchange at the head of every chunk + one per step; one all-gather of
    # the sums per step (FIRE only) + one of the chunk statistics
    chunks = gt // cfg.num_iters
    assert calls['halo'] == chunks * (cfg.num_iters + 1), calls
    assert calls['allgather'] == chunks * ((cfg.num_iters if cfg.fire else 0) + 1), calls
  dist.barrier()
  dist.destroy_process_group()


def test_two_ranks_drive_the_c_banded_loop_on_one_gpu(gpu, tmp_path):
  """The INTER-RANK branch of sfm_mesh_relax_banded (peer indexing, packed edge
  rows, in-place all-gather with two bands per rank, edge-first split on the
  second stream) with a transport that works between two processes sharing
  this GPU: the library stages rows and sums through the host, gloo moves them
  (SfmBandedDesc.host_halo / host_allgather).  2 ranks x 2 bands == 1 process x
  4 bands through the same C loop, bit for bit, on both ranks."""
  import socket
  import torch.multiprocessing as mp
  from sofima_amd import dist as sdist, mesh
  sock = socket.socket()
  sock.bind(('127.0.0.1', 0))
  port = sock.getsockname()[1]
  sock.close()
  mp.spawn(_two_rank_c_loop_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  rng = np.random.default_rng(5)
  x3 = (rng.standard_normal((3, 6, 31, 12)) * 0.5).astype(np.float32)
  p3 = (rng.standard_normal((3, 6, 31, 12)) * 4).astype(np.float32)
  cfg3 = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(20, 20, 20),
                                num_iters=40, max_iters=80, stop_v_max=1e-9, dt_max=1000,
                                start_cap=0.01, final_cap=10)
  cases = {
      'fused': (_case((2, 3, 75, 70), False), {}),
      'fused_drift': (_case((2, 3, 75, 70), True, amp=6.0), {}),
      'fused_serial': (_case((2, 2, 150, 64), True), {'overlap': False}),
      'verlet': (_case((2, 2, 61, 47), False, fire=False), {}),
      'narrow': (_case((2, 2, 61, 33), True), {}),
      'vol': ((x3, p3, cfg3), {'mesh_force': mesh.elastic_mesh_3d}),
  }
  for name, ((x0, prev, cfg), kw) in cases.items():
    wx, we, wt = sdist.relax_mesh_banded(x0, prev, cfg, bands_per_rank=4, **kw)
    for r in range(2):
      np.testing.assert_array_equal(np.load(tmp_path / f'{name}_x_{r}.npy'), wx, err_msg=name)
      np.testing.assert_array_equal(np.load(tmp_path / f'{name}_e_{r}.npy'),
                                    np.array(we + [wt]), err_msg=name)


def test_banded_host_transport_failure_is_an_error_not_a_hang(gpu):
  """A host-staged transport callback that fails aborts the chunk with an error
  code and message (like force_cb / prev_cb), with the exchange stream joined
  back: the next call on the same streams works."""
  import ctypes as C
  import torch
  from sofima_amd import _abi, _dev, dist as sdist, mesh
  x0, prev, cfg = _case((2, 2, 61, 47), False)
  lib = _abi.load()
  dev = _dev.device()
  # rank 0 of 2, one band: rows [0, 30) + one halo row
  bounds = sdist.band_bounds(x0.shape[-2], 2)
  y0, y1 = bounds[0]
  x_t = _dev.as_device_f32(x0[..., :y1 + 1, :], dev)
  v_t = torch.zeros_like(x_t)
  a_t = torch.empty_like(x_t)
  p_t = _dev.as_device_f32(prev[..., :y1 + 1, :], dev)
  spec = mesh._resolve_force(mesh.inplane_force)
  probe = mesh._base_desc(x_t, spec, cfg.k, cfg.stride, cfg.prefer_orig_order)
  wsp = _dev.workspace(lib.sfm_mesh_workspace_bytes(C.byref(probe)), dev)
  descs = (_abi.SfmMeshDesc * 1)()
  descs[0] = mesh._chunk_desc(x_t, v_t, a_t, p_t, cfg, spec, wsp)
  descs[0].stream = _dev.stream_ptr()
  shards = (_abi.SfmMeshShard * 1)()
  shards[0].own_y0, shards[0].own_y1 = 0, y1
  shards[0].global_nodes = int(np.prod(x0.shape[1:]))
  calls = {'n': 0}

  def halo(user, plo, slo, rlo, phi, shi, rhi, count):
    calls['n'] += 1
    assert plo == -1 and phi == 1          # rank 0 only has an upper neighbour
    if calls['n'] >= 3:
      return 1
    np.ctypeslib.as_array(rhi, shape=(count,))[:] = np.ctypeslib.as_array(shi, shape=(count,))
    return 0

  def gather(user, send, recv, count):
    src = np.ctypeslib.as_array(send, shape=(count,)).copy()
    out = np.ctypeslib.as_array(recv, shape=(2 * count,))
    out[:count] = src
    out[count:] = src
    return 0

  bd = _abi.SfmBandedDesc()
  bd.n_local, bd.bands, bd.shards = 1, descs, shards
  bd.rank, bd.n_ranks = 0, 2
  side = torch.cuda.Stream(device=dev)
  bd.comm_stream = side.cuda_stream
  halo_c, gather_c = _abi.HOST_HALO_FN(halo), _abi.HOST_ALLGATHER_FN(gather)
  bd.host_halo, bd.host_allgather = halo_c, gather_c
  scratch = _dev.workspace(lib.sfm_mesh_banded_scratch_bytes(C.byref(bd)), dev)
  bd.scratch, bd.scratch_bytes = scratch.data_ptr(), scratch.numel()
  fire = _abi.SfmFireState()
  fire.dt, fire.alpha, fire.n_pos, fire.cap = cfg.dt, cfg.alpha, 0, cfg.start_cap
  stats = _abi.SfmChunkStats()
  rc = lib.sfm_mesh_relax_banded(C.byref(bd), C.byref(fire), C.byref(stats))
  assert rc == -1 and b'host_halo callback failed' in lib.sfm_last_error()
  assert calls['n'] == 3
  torch.cuda.synchronize(dev)               # nothing left dangling on either stream
  # without a transport at all the same descriptor is refused up front
  bd.host_halo = _abi.HOST_HALO_FN()
  rc = lib.sfm_mesh_relax_banded(C.byref(bd), C.byref(fire), C.byref(stats))
  assert rc == -1 and b'communicator or the host_halo' in lib.sfm_last_error()
  # and the streams still work
  gx, _, gt = mesh.relax_mesh(x0, prev, cfg)
  assert gt == cfg.max_iters and np.isfinite(np.array(gx)).all()
