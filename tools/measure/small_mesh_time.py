"""Launch-bound meshes: step rate with and without the hipGraph replay."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import mesh
for shape, force in (((3, 16, 12, 12, 12), mesh.elastic_mesh_3d), ((2, 4, 17, 17), mesh.inplane_force), ((2, 1, 205, 205), mesh.inplane_force)):
  rng = np.random.default_rng(0)
  prev = (rng.standard_normal(shape) * 3).astype(np.float32)
  nd = shape[0]
  cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(20,) * nd, num_iters=400, max_iters=400,
                               stop_v_max=1e-9, dt_max=1000, start_cap=0.01, final_cap=10, remove_drift=True)
  x = torch.zeros(shape, device='cuda'); pv = torch.from_numpy(prev).cuda()
  for env in ({'SFM_MESH_PERSISTENT': '0', 'SFM_MESH_GRAPH': '0'}, {'SFM_MESH_PERSISTENT': '0', 'SFM_MESH_GRAPH': '1'}):
    os.environ.update(env)
    mesh.relax_mesh(x, pv, cfg, mesh_force=force); torch.cuda.synchronize()
    t = time.perf_counter(); r = mesh.relax_mesh(x, pv, cfg, mesh_force=force); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(shape, env['SFM_MESH_GRAPH'], '%.2f us/step' % (dt / 400 * 1e6), 'e_kin %.6g' % r[1][-1])
