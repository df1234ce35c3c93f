# quick A/B of one option of the correlation kernel: OPT=name SET="v1 v2" (bench pruned leg)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/early
if [ -n "$TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -2; fi
for e in $SET; do
  env $OPT=$e timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/early/bench_q$e.json 2> gpurun_out/early/bench_q$e.err
  python - <<PY
import json
j = json.loads(open('gpurun_out/early/bench_q$e.json').read().strip().splitlines()[-1])
r = j['roofline']['pruned']
print('$OPT=$e value', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 3), 'kernel ms', r['avg_launch_ms'],
      'issued/alg', r['issued_over_algorithmic'], 'abandoned', r.get('row_tiles_abandoned_frac'),
      'MHz', r['sustained_clock_mhz'], 'other pair kernel', j['roofline']['other_pair']['pruned']['avg_launch_ms'])
PY
done
