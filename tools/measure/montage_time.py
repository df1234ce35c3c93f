"""Step rate of a cfg-3 sized montage relaxation (8 x 8 tiles, 204^2 nodes each,
native target-mesh prev_fn).  Neighbour table and flows are synthetic (smooth
random flows, consistent NeighborInfo rows): timing only."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scipy import ndimage
from sofima_amd import mesh, stitch_elastic
NI = stitch_elastic.NeighborInfo
rng = np.random.default_rng(3)
gx, gy, m, ov = 8, 8, 204, 20
n = gx * gy
def smooth(shape, amp):
  a = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 3, 3))
  return (a / np.abs(a).max() * amp).astype(np.float32)
fx = smooth((2, n, m, ov), 4.0)
fy = smooth((2, n, ov, m), 4.0)
nb = -np.ones((n, 4, 8), np.int32)
def entry(nbor, flow_idx, dim, flow):
  e = -np.ones(8, np.int32)
  e[NI.nbor_idx], e[NI.flow_idx], e[NI.dim] = nbor, flow_idx, dim
  fyy, fxx = flow.shape[-2:]
  e[NI.flow_size_overlap] = fxx if dim == 0 else fyy
  e[NI.flow_size_ortho] = fyy if dim == 0 else fxx
  e[NI.coarse_offset_ortho] = rng.integers(-2, 3)
  e[NI.fine_off_x], e[NI.fine_off_y] = rng.integers(-2, 3, 2)
  return e
for t in range(n):
  tx, ty = t % gx, t // gx
  k = 0
  if tx > 0: nb[t, k] = entry(t - 1, t - 1, 0, fx); k += 1
  if tx < gx - 1: nb[t, k] = entry(t + 1, t, 0, fx); k += 1
  if ty > 0: nb[t, k] = entry(t - gx, t - gx, 1, fy); k += 1
  if ty < gy - 1: nb[t, k] = entry(t + gx, t, 1, fy); k += 1
fn = stitch_elastic.TargetMeshFn(nb, fx, fy, (20.0, 20.0))
x = torch.zeros((2, n, m, m), device='cuda')
cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(20, 20), num_iters=200, max_iters=200,
                             stop_v_max=1e-9, dt_max=1000, start_cap=0.1, final_cap=10, prefer_orig_order=True,
                             remove_drift=True)
for env in ({}, {'SFM_MESH_FUSE_TARGET': '0'}, {'SFM_MESH_TILED': '0'}):
  os.environ.pop('SFM_MESH_TILED', None); os.environ.pop('SFM_MESH_FUSE_TARGET', None); os.environ.update(env)
  mesh.relax_mesh(x, None, cfg, prev_fn=fn); torch.cuda.synchronize()
  t = time.perf_counter(); _, ek, it = mesh.relax_mesh(x, None, cfg, prev_fn=fn); torch.cuda.synchronize(); dt = time.perf_counter() - t
  nodes = n * m * m
  print('montage 8x8x204^2 %s: %.1f us/step, %.1f G node-updates/s' % (env or 'tiled', dt / it * 1e6, nodes * it / dt / 1e9))
