"""Step time of a volumetric montage [3,64,12,12,12]: the multi-launch step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import mesh, stitch_elastic, _abi
from tests.util import synth_montage
rng = np.random.default_rng(1006)
nb, fx, fy, x0 = synth_montage(rng, 8, 8, (12, 12, 12), 3, amp=5.0)
stride = (40.0, 40.0, 40.0)
drift = os.environ.get('DRIFT', '1') == '1'
cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=stride, num_iters=500, max_iters=500,
                             stop_v_max=1e-9, dt_max=100, start_cap=0.1, final_cap=10.0, remove_drift=drift)
fn = stitch_elastic.TargetMeshFn(nb, fx, fy, stride)
x = torch.from_numpy(x0).cuda()
mesh.relax_mesh(x, None, cfg, mesh_force=mesh.elastic_mesh_3d, prev_fn=fn); torch.cuda.synchronize()
t = time.perf_counter(); _, ek, it = mesh.relax_mesh(x, None, cfg, mesh_force=mesh.elastic_mesh_3d, prev_fn=fn); torch.cuda.synchronize()
dt = time.perf_counter() - t
print('volumetric montage [3,64,12,12,12] drift=%d: %.1f us/step' % (drift, dt / it * 1e6), flush=True)
