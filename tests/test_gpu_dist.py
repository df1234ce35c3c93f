"""Multi-GPU building blocks on ONE GPU (-m gpu): the band-sharded mesh step
(sfm_mesh_shard_*) with several bands in one process, and the library's RCCL
entry points at world size 1 (self send / recv, all-gather, all-reduce)."""
import numpy as np
import pytest

from oracle import mesh_oracle

pytestmark = pytest.mark.gpu


def _case(shape, drift, fire=True):
  from scipy import ndimage
  from sofima_amd import mesh
  rng = np.random.default_rng(7)
  prev = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 3, 3)) * 30
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
  np.testing.assert_allclose(ge, we, rtol=1e-2)
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


def test_block_chain_on_device_vs_oracle(gpu):
  """configs[3] block chain with the HIP relax / compose ops vs the oracle."""
  from oracle import maps_oracle
  from scipy import ndimage
  from sofima_amd import dist as sdist, mesh
  rng = np.random.default_rng(6)
  flow = ndimage.gaussian_filter(rng.standard_normal((2, 6, 30, 34)), (0, 0, 3, 3)) * 12
  flow = flow.astype(np.float32)
  flow[:, 2, :2, :3] = np.nan
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.05, k=0.1, stride=(40, 40),
                               num_iters=100, max_iters=1000, stop_v_max=0.005,
                               dt_max=1000, start_cap=0.1, final_cap=10,
                               prefer_orig_order=True)
  blocks, last, xblk = sdist.align_sections_blocked(flow, cfg, 40.0, n_blocks=3)
  wb, wl, wx = sdist.align_sections_blocked(
      flow, cfg, 40.0, n_blocks=3, relax_fn=mesh_oracle.relax_mesh,
      compose_fn=maps_oracle.compose_maps_fast)
  np.testing.assert_allclose(last, wl, atol=2e-2)
  np.testing.assert_allclose(xblk, wx, atol=2e-2)
  for b in range(3):
    np.testing.assert_array_equal(np.isnan(blocks[b]), np.isnan(wb[b]))
    np.testing.assert_allclose(np.nan_to_num(blocks[b]), np.nan_to_num(wb[b]), atol=2e-2)
