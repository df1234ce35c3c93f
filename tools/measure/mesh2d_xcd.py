"""Large in-plane meshes: step time with / without the XCD-contiguous tile mapping (SFM_MESH_XCD)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import mesh, _abi
rng = np.random.default_rng(0)
iters = int(os.environ.get('ITERS', '200'))
shapes = ((2, 4, 2048, 2048), (2, 64, 204, 204), (2, 1, 1000, 1000))
if os.environ.get('ONLY'): shapes = shapes[:1]
for shape in shapes:
  prev = torch.from_numpy((rng.standard_normal(shape) * 3).astype(np.float32)).cuda()
  x0 = torch.zeros_like(prev)
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40), num_iters=iters,
                               max_iters=iters, stop_v_max=1e-9, dt_max=1000, start_cap=0.1, final_cap=10)
  nodes = int(np.prod(shape[1:]))
  ref = None
  for opt in (0, 1, 0, 1):   # explicit off / on (the default switches at 2048 tiles)
    with _abi.option('SFM_MESH_XCD', opt):
      r = mesh.relax_mesh(x0, prev, cfg); torch.cuda.synchronize()
      t = time.perf_counter(); r = mesh.relax_mesh(x0, prev, cfg); torch.cuda.synchronize()
      dt = time.perf_counter() - t
    xr = r[0].cpu().numpy() if hasattr(r[0], 'cpu') else np.asarray(r[0])
    if ref is None: ref = xr
    print(shape, 'xcd=%d' % opt, '%.1f us/step %.2f TB/s alg' % (dt / iters * 1e6, nodes * iters * 56 / dt / 1e12),
          'same' if np.array_equal(ref, xr) else 'DIFF %g' % np.abs(ref - xr).max(), flush=True)
