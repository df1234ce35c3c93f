"""integrate_march3d_kernel (every spring of a default-link volume evaluated once)
against integrate_kernel<3>: bit-identity of the damped-Verlet trajectories on a set
of shapes, closeness of the FIRE ones, and the per-step time on the aux leg's
[3, 4, 100, 100, 100] state.

  python tools/measure/march3d_ab.py [check|time|all]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
import torch

from sofima_amd import _abi, mesh

SHAPES = [(3, 1, 5, 7, 9), (3, 2, 40, 33, 70), (3, 17, 23, 100), (3, 1, 1, 30, 300),
          (3, 3, 30, 1, 40), (3, 2, 64, 64, 1), (3, 1, 12, 12, 12), (3, 1, 9, 130, 61)]


def run(x0, prev, cfg, on, extra=()):
  opts = [('SFM_MESH_MARCH3D', 1 if on else 0)] + list(extra)
  ctx = [_abi.option(k, v) for k, v in opts]
  for c in ctx:
    c.__enter__()
  try:
    return mesh.relax_mesh(x0.copy(), None if prev is None else prev.copy(), cfg,
                           mesh_force=mesh.elastic_mesh_3d)
  finally:
    for c in reversed(ctx):
      c.__exit__(None, None, None)


def check():
  rng = np.random.default_rng(3)
  bad = 0
  for shape in SHAPES:
    for prefer in (False, True):
      for has_prev in (True, False):
        x0 = (rng.standard_normal(shape) * 2).astype(np.float32)
        prev = (rng.standard_normal(shape) * 4).astype(np.float32) if has_prev else None
        if has_prev:
          prev[:, ..., :1] = np.nan
        kw = dict(dt=0.05, gamma=0.5, k0=0.05, k=0.1, stride=(10, 10, 10), num_iters=7,
                  max_iters=7, stop_v_max=1e-9, dt_max=100, start_cap=10.0, final_cap=10.0,
                  fire=False, prefer_orig_order=prefer)
        cfg = mesh.IntegrationConfig(**kw)
        ref = run(x0, prev, cfg, False)
        for t, zc in ((0, 0), (256, 2), (512, 3), (1024, 0)):
          extra = ([('SFM_MESH_MARCH3D_T', t)] if t else []) + \
                  ([('SFM_MESH_MARCH3D_ZC', zc)] if zc else [])
          got = run(x0, prev, cfg, True, extra)
          same = np.array_equal(np.asarray(ref[0]), np.asarray(got[0]), equal_nan=True)
          if not same:
            bad += 1
            d = np.abs(np.asarray(ref[0]) - np.asarray(got[0]))
            print('MISMATCH verlet', shape, prefer, has_prev, t, zc, float(np.nanmax(d)),
                  int((d > 0).sum()), flush=True)
        # FIRE: the sums are added in another order
        kw.update(fire=True, dt=0.001, gamma=0.0, num_iters=30, max_iters=30, start_cap=1.0,
                  remove_drift=True)
        cfg = mesh.IntegrationConfig(**kw)
        ref = run(x0, prev, cfg, False)
        got = run(x0, prev, cfg, True)
        scale = float(np.nanmax(np.abs(np.asarray(ref[0]))))
        err = float(np.nanmax(np.abs(np.asarray(ref[0]) - np.asarray(got[0]))))
        if not err <= 2e-4 * scale or ref[2] != got[2]:
          bad += 1
          print('FIRE far', shape, prefer, has_prev, err, scale, flush=True)
    print('checked', shape, 'bad so far', bad, flush=True)
  return bad


def timed(shape, on, iters=200, extra=()):
  dev = torch.device('cuda:0')
  rng = np.random.default_rng(0)
  prev = torch.from_numpy((rng.standard_normal(shape) * 3).astype(np.float32)).to(dev)
  x0 = torch.zeros_like(prev)
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40, 40),
                               num_iters=iters, max_iters=iters, stop_v_max=1e-9,
                               dt_max=1000, start_cap=0.1, final_cap=10)
  opts = [('SFM_MESH_MARCH3D', 1 if on else 0)] + list(extra)
  ctx = [_abi.option(k, v) for k, v in opts]
  for c in ctx:
    c.__enter__()
  try:
    best = 1e9
    for _ in range(3):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      r = mesh.relax_mesh(x0, prev, cfg, mesh_force=mesh.elastic_mesh_3d)
      torch.cuda.synchronize()
      best = min(best, time.perf_counter() - t0)
  finally:
    for c in reversed(ctx):
      c.__exit__(None, None, None)
  return best / iters * 1e6, r


def main():
  what = sys.argv[1] if len(sys.argv) > 1 else 'all'
  if what in ('check', 'all'):
    print('mismatches', check(), flush=True)
  if what in ('time', 'all'):
    shapes = [(3, 4, 100, 100, 100), (3, 1, 160, 160, 160), (3, 64, 12, 12, 12),
              (3, 8, 48, 48, 48), (3, 1, 64, 64, 64)]
    if os.environ.get('MARCH_SHAPES'):
      shapes = [shapes[int(i)] for i in os.environ['MARCH_SHAPES'].split(',')]
    ts = [int(t) for t in os.environ.get('MARCH_T', '0,256,512,1024').split(',')]
    for shape in shapes:
      us0, r0 = timed(shape, False)
      print(shape, 'integrate_kernel<3>: %.1f us per step' % us0, flush=True)
      for t in ts:
        for zc in (0,):
          extra = ([('SFM_MESH_MARCH3D_T', t)] if t else []) + \
                  ([('SFM_MESH_MARCH3D_ZC', zc)] if zc else [])
          us1, r1 = timed(shape, True, extra=extra)
          err = float(np.nanmax(np.abs(np.asarray(r0[0].cpu() if hasattr(r0[0], 'cpu') else r0[0]) -
                                       np.asarray(r1[0].cpu() if hasattr(r1[0], 'cpu') else r1[0]))))
          print(shape, 'march T=%d zc=%d: %.1f us per step (%.2fx), max |dx| %.3g' %
                (t, zc, us1, us0 / us1, err), flush=True)


if __name__ == '__main__':
  main()
