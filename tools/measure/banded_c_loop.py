import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, dataclasses, torch
from tests.test_gpu_dist import _case
from sofima_amd import dist as sdist, mesh, _abi
for amp in (30.0, 6.0):
  x0, prev, cfg = _case((2, 3, 75, 70), True, True, amp)
  for n in (50, 150):
   for chunks in (1, 3):
    c = dataclasses.replace(cfg, num_iters=n // chunks, max_iters=n)
    with _abi.option('SFM_MESH_PERSISTENT', 0):
      sx, se, st = mesh.relax_mesh(x0, prev, c)
    sx = np.array(sx)
    for nb in (2, 3, 4):
      gx, ge, gt = sdist.relax_mesh_banded(x0, prev, c, bands_per_rank=nb)
      print('amp', amp, 'steps', n, 'chunks', chunks, 'bands', nb, 'maxdiff/scale %.3e' % (np.abs(gx - sx).max() / np.abs(sx).max()), 'ekin', ge[-1], se[-1], flush=True)
rng = np.random.default_rng(11)
shape = (2, 64, 204, 204)
prev = (rng.standard_normal(shape) * 2).astype(np.float32)
x0 = np.zeros(shape, np.float32)
iters = 300
cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(20., 20.), num_iters=iters, max_iters=iters, stop_v_max=1e-9, dt_max=1000, start_cap=0.01, final_cap=10, prefer_orig_order=True)
x_d = torch.from_numpy(x0).cuda(); p_d = torch.from_numpy(prev).cuda()
for tag, env in (('persist-or-tiled', None),):
  mesh.relax_mesh(x_d, p_d, cfg); torch.cuda.synchronize()
  t0 = time.perf_counter(); mesh.relax_mesh(x_d, p_d, cfg); torch.cuda.synchronize()
  print('un-split us/step', (time.perf_counter() - t0) / iters * 1e6)
for nb in (1, 2, 4, 8):
  for kw in ({}, {'loopback': True}, {'loopback': True, 'overlap': False}):
    if nb == 1 and kw: continue
    sdist.relax_mesh_banded(x0, prev, cfg, bands_per_rank=nb, **kw)
    tm = {}
    sdist.relax_mesh_banded(x0, prev, cfg, bands_per_rank=nb, timing=tm, **kw)
    print('bands', nb, kw, 'us/step %.1f' % (tm['banded_chunk_s'] / iters * 1e6), flush=True)
