# A/B of the correction table built in the epilogue (SFM_MFMA_LAZYG): tests, then bench + trace either way
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; mkdir -p $O
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest tests/test_gpu_flow.py tests/test_gpu_prune_hardening.py -x -q -m gpu $TESTARGS 2>&1 | tail -15; fi
for e in ${SET:-1 0}; do
  env SFM_MFMA_LAZYG=$e timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-legs --sustain 0 > $O/bench_g$e.json 2> $O/bench_g$e.err
  python - <<PY
import json
j = json.loads(open('$O/bench_g$e.json').read().strip().splitlines()[-1])
r = j['roofline']['pruned']
print('LAZYG=$e value', round(j['value'], 1), 'flow ms', j.get('flow_ms_per_step'), 'ms/step', round(j['ms_per_step'], 3), 'kernel ms', r['avg_launch_ms'],
      'issued/alg', r['issued_over_algorithmic'], 'MHz', r['sustained_clock_mhz'])
PY
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in ${SET:-1 0}; do
env SFM_MFMA_LAZYG=$e timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace_g$e -o t -- python $R/bench.py --no-cpu-baseline --no-legs --sustain 0 --steps 3 --warmup 1 > $R/$O/trace_g$e.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/$O/trace_g$e -name '*.db' | head -1) | head -6
done
find $R/$O -name '*.db' -delete
