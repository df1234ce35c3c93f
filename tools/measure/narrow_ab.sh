# A/B of the in-flight column narrowing of the correlation kernel (SFM_MFMA_NARROW)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/early
echo skip tests

for e in ${NARROW_SET:-6 5 64 0}; do
  SFM_MFMA_NARROW=$e timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/early/bench_n$e.json 2> gpurun_out/early/bench_n$e.err
  python - <<PY
import json
j = json.loads(open('gpurun_out/early/bench_n$e.json').read().strip().splitlines()[-1])
r = j['roofline']['pruned']
print('NARROW=$e value', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 3), 'kernel ms', r['avg_launch_ms'],
      'issued/alg', r['issued_over_algorithmic'], 'abandoned', r.get('row_tiles_abandoned_frac'),
      'skipped', r['row_tiles_skipped_frac'], 'MHz', r['sustained_clock_mhz'], 'other pair kernel', j['roofline']['other_pair']['pruned']['avg_launch_ms'])
PY
done
