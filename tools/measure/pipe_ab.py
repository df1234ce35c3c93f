"""Cross-patch pipeline of the correlation kernel (SFM_MFMA_PIPE=1) against the
two-workgroup kernel: bit-identity on the stress images of the pruning tests and on
the bench pair, then the same-box A/B of the production launch (HIP-event time of
the correlation kernel through the library's profile hooks, and flow wall time).

  python tools/measure/pipe_ab.py [check|time|all] [reps]
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
import torch

from bench import synth_pair, WARP
from sofima_amd import _abi, flow_field as ff


def check_small():
  from tests.test_gpu_flow import _prune_images
  bad = 0
  rng = np.random.default_rng(5)
  for kind in ('em', 'far', 'periodic', 'noise', 'edges', 'smooth', 'fine'):
    h, w = 460, 500
    pre, post = _prune_images(kind, 31, h, w)
    for (py, px), radius, md, thr in [((160, 160), 5, 2, 0.5), ((160, 160), 30, 2, 0.9),
                                      ((160, 160), 5, 2, 0.2), ((128, 160), 5, 2, 0.5)]:
      for b in (24, 57, 1):
        starts = np.stack([rng.integers(-10, h - py + 10, b),
                           rng.integers(-10, w - px + 10, b)], axis=1)
        kw = dict(min_distance=md, threshold_rel=thr, peak_radius=radius,
                  post_patch_size=(py, px), post_starts=starts)
        for mean in (None, 100.0):
          args = (pre, post, None, None, (py, px), starts, mean)
          ref = ff.batched_xcorr_peaks(*args, method=2, **kw)
          for grid in (None, 2, 6):
            with _abi.option('SFM_MFMA_PIPE', 1):
              if grid:
                with _abi.option('SFM_MFMA_GRID', grid):
                  got = ff.batched_xcorr_peaks(*args, method=2, **kw)
              else:
                got = ff.batched_xcorr_peaks(*args, method=2, **kw)
            if not np.array_equal(ref, got, equal_nan=True):
              bad += 1
              print('MISMATCH', kind, (py, px), radius, thr, b, mean, grid,
                    int((~np.isclose(ref, got, equal_nan=True)).sum()), flush=True)
    print('checked', kind, 'bad so far', bad, flush=True)
  return bad


def timed(pre_d, post_d, reps, pipe):
  lib = _abi.load()
  calc = ff.JAXMaskedXCorrWithStatsCalculator()
  with _abi.option('SFM_MFMA_PIPE', 1 if pipe else 0):
    f = calc.flow_field(pre_d, post_d, 160, 40, batch_size=1024)
    torch.cuda.synchronize()
    prof = _abi.SfmProfile()
    lib.sfm_profile_read(C.byref(prof))
    lib.sfm_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(reps):
      f = calc.flow_field(pre_d, post_d, 160, 40, batch_size=1024)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    lib.sfm_profile_enable(0)
    lib.sfm_profile_read(C.byref(prof))
  return f, dt, prof


def main():
  what = sys.argv[1] if len(sys.argv) > 1 else 'all'
  reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
  if what in ('check', 'all'):
    bad = check_small()
    print('small cases: mismatches', bad, flush=True)
  if what in ('time', 'all'):
    pre, post = synth_pair(8192, 1002, warp=WARP)
    dev = torch.device('cuda:0')
    pre_d, post_d = torch.from_numpy(pre).to(dev), torch.from_numpy(post).to(dev)
    res = {}
    for rnd in range(3):
      for pipe in (0, 1):
        f, dt, prof = timed(pre_d, post_d, reps, pipe)
        n = max(int(prof.launches[0]), 1)
        print(f'round {rnd} pipe={pipe}: flow {dt * 1e3:.3f} ms per pair; correlation kernel '
              f'{prof.kernel_ms[0] / n:.3f} ms per launch ({n} launches), clock '
              f'{prof.clock_mhz[0]:.0f} MHz, mfma issued {prof.mfma_issued[0] / n / 1e6:.1f} M, '
              f'tiles drawn {prof.tiles_drawn[0] / n:.0f} skipped {prof.tiles_skipped[0] / n:.0f} '
              f'abandoned {prof.tiles_abandoned[0] / n:.0f}', flush=True)
        res[pipe] = np.asarray(f)
    print('bench pair identical:', np.array_equal(res[0], res[1], equal_nan=True), flush=True)


if __name__ == '__main__':
  main()
