"""Time-boxed random fuzz of integrate_march3d_kernel against integrate_kernel<3>:
random volume shapes (axes of 1 ... 70, batches), workgroup sizes, plane runs,
prefer_orig_order, NaN targets, NaN / inf positions; damped-Verlet chunks of 1-5 steps
must agree bit for bit (x, v, a).

  python tools/measure/march3d_fuzz.py [seconds]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np

from sofima_amd import _abi, mesh


def main():
  budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
  rng = np.random.default_rng(int(time.time()))
  t0 = time.time()
  cases = bad = 0
  while time.time() - t0 < budget:
    dims = [int(rng.choice([1, 2, 3, 5, 9, 17, 33, 40, 70])) for _ in range(3)]
    if dims[0] * dims[1] * dims[2] > 120000:
      continue
    b = int(rng.choice([1, 1, 2, 5]))
    shape = (3, b, *dims) if rng.random() < 0.7 else (3, *dims)
    x0 = (rng.standard_normal(shape) * rng.choice([0.5, 3.0, 30.0])).astype(np.float32)
    v0 = (rng.standard_normal(shape) * 0.2).astype(np.float32)
    prev = None
    if rng.random() < 0.6:
      prev = (rng.standard_normal(shape) * 4).astype(np.float32)
      prev[rng.random(shape) < 0.1] = np.nan
    if rng.random() < 0.3:
      for badv in (np.nan, np.inf, -np.inf):
        x0[rng.random(shape) < 0.002] = badv
    cfg = mesh.IntegrationConfig(
        dt=0.05, gamma=0.5, k0=0.05, k=0.1, stride=tuple(float(s) for s in rng.choice([10, 25, 40], 3)),
        num_iters=int(rng.integers(1, 6)), max_iters=5, stop_v_max=1e-9, dt_max=100,
        start_cap=10.0, final_cap=10.0, fire=False, prefer_orig_order=bool(rng.random() < 0.5))
    run = lambda: [np.array(t) for t in mesh.velocity_verlet(
        x0, v0, prev, cfg, cfg.start_cap, mesh_force=mesh.elastic_mesh_3d)]
    with _abi.option('SFM_MESH_MARCH3D', 0):
      want = run()
    t = int(rng.choice([0, 256, 512, 1024]))
    zc = int(rng.choice([0, 1, 2, 3, 7]))
    with _abi.option('SFM_MESH_MARCH3D', 1), _abi.option('SFM_MESH_MARCH3D_T', t or None), \
        _abi.option('SFM_MESH_MARCH3D_ZC', zc or None):
      got = run()
    cases += 1
    if not all(np.array_equal(w, g, equal_nan=True) for w, g in zip(want, got)):
      bad += 1
      print('MISMATCH', shape, t, zc, cfg.prefer_orig_order, prev is not None, flush=True)
  print('z-march against per-node kernel: %d random cases, %d mismatches' % (cases, bad), flush=True)
  return bad


if __name__ == '__main__':
  sys.exit(1 if main() else 0)
