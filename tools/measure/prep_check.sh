# flow tests + bench line + kernel trace of the production launches (prep kernel changes)
cd $GRAFT_REPO_ROOT; O=gpurun_out/prep; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_flow.py tests/test_gpu_prune_hardening.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|rror" | tail -8
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-legs --sustain 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
r = j['roofline']['pruned']
print('value', round(j['value'], 1), 'flow ms', j.get('flow_ms_per_step'), 'mesh ms', j.get('mesh_ms_per_step'), 'ms/step', round(j['ms_per_step'], 3), 'kernel ms', r['avg_launch_ms'])
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o t -- python $R/bench.py --no-cpu-baseline --no-legs --sustain 0 --steps 3 --warmup 1 > $R/$O/trace.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/$O/trace -name '*.db' | head -1) | head -8 | cut -c1-160
find $R/$O -name '*.db' -delete
