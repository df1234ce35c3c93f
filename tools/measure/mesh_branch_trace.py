"""Where do the FIRE branch sequences of the HIP kernel and the oracle part on the
headline mesh leg (1000 steps, [2,1,205,205])?  Oracle: per-step snapshots incl.
the power; HIP: runs of k steps.  argv: first step, last step."""
import dataclasses, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
import bench
from oracle import mesh_oracle
from sofima_amd import mesh
from tests.util import cfg_from

lo, hi = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1002)
n = (8192 - (bench.PATCH - bench.STEP)) // bench.STEP
yy, xx = np.mgrid[:n, :n].astype(np.float32) * bench.STEP + bench.PATCH / 2
d = bench.WARP[0] * np.sin(2 * np.pi * xx / bench.WARP[1]) * np.cos(2 * np.pi * yy / bench.WARP[1])
flow = np.stack([np.rint(-5 - d), np.rint(3 + d)]).astype(np.float32)
flow[:, rng.random((n, n)) < 0.003] = np.nan
prev = bench.mesh_inputs(flow, bench.PATCH // 2 // bench.STEP)
cfg = mesh.IntegrationConfig(
    dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(bench.STEP, bench.STEP),
    num_iters=hi, max_iters=hi, stop_v_max=0.005,
    dt_max=1000, start_cap=0.01, final_cap=10, prefer_orig_order=True)
ocfg = cfg_from(dataclasses.asdict(cfg))
x0 = np.zeros_like(prev)
snaps = []
mesh_oracle.velocity_verlet(x0, x0.copy(), prev, ocfg, cfg.start_cap, snapshots=snaps,
                            snapshot_every=1)
by_step = {s[0]: s for s in snaps}
scale = float(np.abs(snaps[-1][1]).max())
for k in range(lo, hi):
  go = mesh.velocity_verlet(x0, x0.copy(), prev, dataclasses.replace(cfg, num_iters=k), cfg.start_cap)
  _, wx, wdt, walpha, wnpos, wcap = by_step[k][:6]
  extra = by_step[k][6:] if len(by_step[k]) > 6 else ()
  print(k, 'hip dt %.6g a %.6g n %d cap %.4g | ora dt %.6g a %.6g n %d cap %.4g | dx/scale %.2e' % (
      go[3], go[4], go[5], go[6], wdt, walpha, wnpos, wcap,
      float(np.abs(np.array(go[0]) - wx).max()) / scale), *extra)
