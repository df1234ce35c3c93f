cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/early
echo skip

for e in ${SET:-1 2 3}; do
  SFM_MFMA_EARLY=$e timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/early/bench_s$e.json 2> gpurun_out/early/bench_s$e.err
  python - <<PY
import json
j = json.loads(open('gpurun_out/early/bench_s$e.json').read().strip().splitlines()[-1])
r = j['roofline']['pruned']
print('EARLY=$e value', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 3), 'kernel ms', r['avg_launch_ms'],
      'issued/alg', r['issued_over_algorithmic'], 'abandoned', r.get('row_tiles_abandoned_frac'),
      'MHz', r['sustained_clock_mhz'], 'other pair kernel', j['roofline']['other_pair']['pruned']['avg_launch_ms'])
PY
done

