# one workgroup per CU against two: how much of the matrix pipe does a lone wave per SIMD fill?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/early
for w in 1 2; do
  SFM_MFMA_MAX_WG_PER_CU=$w timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/early/bench_wg$w.json 2> gpurun_out/early/bench_wg$w.err
  python - <<PY
import json
j = json.loads(open('gpurun_out/early/bench_wg$w.json').read().strip().splitlines()[-1])
r = j['roofline']
print('WG/CU=$w value', j['value'], 'ms', j['ms_per_step'], 'pruned', r['pruned']['avg_launch_ms'], r['pruned']['issued_over_algorithmic'], r['pruned']['sustained_clock_mhz'], 'unpruned', r['unpruned']['avg_launch_ms'], r['unpruned']['sustained_clock_mhz'])
PY
done
