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
