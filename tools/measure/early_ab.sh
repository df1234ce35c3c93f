# A/B of the in-loop cold test of the correlation kernel (SFM_MFMA_EARLY = row groups
# between two tests): the parity tests that pin it, then the bench line per setting.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/early
timeout 900 python -m pytest tests/test_gpu_flow.py -x -q -m gpu -k "pruned or abandoned" > gpurun_out/early/tests.log 2>&1
tail -3 gpurun_out/early/tests.log
for e in ${EARLY_SET:-0 2 4 6 8}; do
  SFM_MFMA_EARLY=$e timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/early/bench_$e.json 2> gpurun_out/early/bench_$e.err
  python - <<PY
import json
j = json.loads(open('gpurun_out/early/bench_$e.json').read().strip().splitlines()[-1])
r = j['roofline']['pruned']
print('EARLY=$e value', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 3), 'kernel ms', r['avg_launch_ms'],
      'issued/alg', r['issued_over_algorithmic'], 'abandoned', r.get('row_tiles_abandoned_frac'),
      'skipped', r['row_tiles_skipped_frac'], 'MHz', r['sustained_clock_mhz'])
PY
done
