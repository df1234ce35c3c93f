"""Copies what tools/measure/profile_round6.sh left under gpurun_out/r6p/ into the
tracked profiles/r06_* files (run here, after the gpurun call has merged its output).

  python tools/measure/collect_round6.py
"""
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O = os.path.join(ROOT, 'gpurun_out', 'r6p')
P = os.path.join(ROOT, 'profiles')
sha = open(os.path.join(O, 'build_sha.txt')).read().strip()
rd = lambda f: open(os.path.join(O, f)).read().strip()

with open(os.path.join(P, 'r06_trace_summary.md'), 'w') as f:
  f.write('# r06: kernel trace of bench.py --no-cpu-baseline --no-legs --sustain 0 --steps 3 '
          '--warmup 1 (production launches only); build %s\n\n' % sha)
  f.write(rd('trace_summary.md') + '\n\n')
  f.write('# the same bench WITH its roofline legs (--steps 1 --warmup 1): xcorr_mfma_kernel<10, 11, 5> '
          '+ mfma_prep_same_kernel<false> are the un-pruned leg roofline.frac is computed from\n\n')
  f.write(rd('trace_aux_summary.md') + '\n')
with open(os.path.join(P, 'r06_trace_masked_summary.md'), 'w') as f:
  f.write('# r06: kernel trace of tools/measure/masked_time.py, case "blobs r=130" '
          '(configs[1] masked flow); build %s\n\n' % sha)
  f.write(rd('trace_masked_summary.md') + '\n')
parts = [('mesh_time', 'mesh_time.log'), ('montage_time', 'montage_time.log'),
         ('montage3d_time', 'montage3d_time.log'), ('masked_time', 'masked_time.log'),
         ('pipe_ab', 'pipe_ab.log'), ('patch_size_rates', 'patch_size_rates.txt'),
         ('search_window_rates', 'search_window_rates.txt'),
         ('march3d (integrate_kernel<3> against integrate_march3d_kernel, auto plan)',
          'march3d_trace_summary.txt'),
         ('march3d_sizes (per-node kernel against z-march over volume sizes)', 'march3d_sizes.txt'),
         ('pytest -m gpu', 'pytest.log')]
with open(os.path.join(P, 'r06_other_times.txt'), 'w') as f:
  f.write('# r06 other timings, build %s\n' % sha)
  for name, src in parts:
    f.write('## %s\n%s\n' % (name, rd(src)))
for src, dst in [('pmc_sq_full.txt', 'r06_pmc_sq_full.txt'), ('pmc_sq_pruned.txt', 'r06_pmc_sq_pruned.txt'),
                 ('soak_fuzz.log', 'r06_soak_fuzz.txt'), ('bench.json', 'r06_bench.json'),
                 ('pmc_traffic.json', 'r06_pmc_traffic.json'),
                 ('pmc_traffic_aux.json', 'r06_pmc_traffic_aux.json')]:
  shutil.copy(os.path.join(O, src), os.path.join(P, dst))
print('profiles/r06_* <- gpurun_out/r6p (build %s)' % sha)
