cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_configs.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline --no-legs --sustain 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['flow_ms_per_step'], j['roofline']['traffic'])"; done
