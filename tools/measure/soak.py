"""Soak: repeated flow / mesh runs must be bit-identical (races in the dynamic
patch queue, the LDS-direct prefetch, the granule exchange would show up here)."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import flow_field, mesh
from bench import synth_pair
pre, post = synth_pair(4096, 7)
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
ref = calc.flow_field(a, b, 160, 40, batch_size=1024)
bad = 0
t0 = time.time()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for i in range(n):
  f = calc.flow_field(a, b, 160, 40, batch_size=1024)
  if not np.array_equal(f, ref, equal_nan=True): bad += 1
print('flow: %d runs, %d mismatches, %.1f s' % (n, bad, time.time() - t0))
rng = np.random.default_rng(0)
prev = (rng.standard_normal((2, 1, 205, 205)) * 5).astype(np.float32)
cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40), num_iters=500, max_iters=500,
                             stop_v_max=1e-9, dt_max=1000, start_cap=0.01, final_cap=10, prefer_orig_order=True)
pv = torch.from_numpy(prev).cuda()
r0 = np.array(mesh.relax_mesh(torch.zeros_like(pv), pv, cfg)[0])
bad = 0
for i in range(n // 3):
  r = np.array(mesh.relax_mesh(torch.zeros_like(pv), pv, cfg)[0])
  if not np.array_equal(r, r0): bad += 1
print('mesh (persistent): %d runs, %d mismatches' % (n // 3, bad))
big = (rng.standard_normal((2, 8, 204, 204)) * 5).astype(np.float32)
pb = torch.from_numpy(big).cuda()
cfg2 = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40), num_iters=100, max_iters=100,
                              stop_v_max=1e-9, dt_max=1000, start_cap=0.01, final_cap=10, remove_drift=True)
r0 = np.array(mesh.relax_mesh(torch.zeros_like(pb), pb, cfg2)[0])
bad = 0
for i in range(n // 6):
  r = np.array(mesh.relax_mesh(torch.zeros_like(pb), pb, cfg2)[0])
  if not np.array_equal(r, r0): bad += 1
print('mesh (tiled): %d runs, %d mismatches' % (n // 6, bad))
# masked correlation (class lists, integral-image tables, two-pass assembly, atomics)
pm = np.zeros(pre.shape, bool); qm = np.zeros(pre.shape, bool)
for k in range(40):
  y, x = rng.integers(100, 3900, 2)
  (pm if k % 2 else qm)[y - 80:y + 80, x - 80:x + 80] = True
pmt = torch.from_numpy(pm).cuda(); qmt = torch.from_numpy(qm).cuda()
run = lambda: calc.flow_field(a, b, 160, 40, pre_mask=pmt, post_mask=qmt, batch_size=1024,
                              mask_only_for_patch_selection=False)
ref = run()
bad = 0
for i in range(n // 6):
  if not np.array_equal(run(), ref, equal_nan=True): bad += 1
print('flow (masked): %d runs, %d mismatches' % (n // 6, bad))
