# flow time vs patches per C call (workspace size): gpurun -- bash tools/measure/launch_ab.sh
cd $GRAFT_REPO_ROOT
for r in 1 2; do for L in 45056 20480 14336 10240; do
  SFM_LAUNCH_PATCHES=$L timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs --sustain 0 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LAUNCH_PATCHES $L', 'Mpix/s %.0f flow %.3f ms' % (b['value'], b['flow_ms_per_step']))"
done; done
