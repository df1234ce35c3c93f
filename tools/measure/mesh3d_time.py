"""Step rate of volumetric meshes: brick kernel vs multi-launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import mesh
for shape in ((3, 4, 100, 100, 100), (3, 1, 64, 256, 256), (3, 16, 12, 12, 12), (3, 64, 12, 12, 12)):
  rng = np.random.default_rng(0)
  prev = torch.from_numpy((rng.standard_normal(shape) * 3).astype(np.float32)).cuda()
  x = torch.zeros_like(prev)
  iters = 200
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40, 40), num_iters=iters,
                               max_iters=iters, stop_v_max=1e-9, dt_max=1000, start_cap=0.1, final_cap=10)
  nodes = int(np.prod(shape[1:]))
  for env in ({'SFM_MESH_BRICKS': '1'}, {}):
    os.environ.pop('SFM_MESH_BRICKS', None); os.environ.update(env)
    mesh.relax_mesh(x, prev, cfg, mesh_force=mesh.elastic_mesh_3d); torch.cuda.synchronize()
    t = time.perf_counter(); mesh.relax_mesh(x, prev, cfg, mesh_force=mesh.elastic_mesh_3d); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(shape, 'bricks' if env else 'multi-launch', '%.1f us/step  %.2f G node-updates/s  %.2f TB/s algorithmic' % (
        dt / iters * 1e6, nodes * iters / dt / 1e9, nodes * iters * 84 / dt / 1e12))
