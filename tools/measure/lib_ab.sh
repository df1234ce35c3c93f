# same-box A/B of two builds of the library on the bench step (flow + mesh):
#   gpurun -- bash tools/measure/lib_ab.sh libsofima_amd_prev.so libsofima_amd.so [rounds]
cd $GRAFT_REPO_ROOT; A=$1; B=$2; N=${3:-3}
for r in $(seq $N); do for L in $A $B; do
  SOFIMA_AMD_LIB=$GRAFT_REPO_ROOT/sofima_amd/lib/$L timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs --sustain 3 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', 'Mpix/s %.0f flow %.3f ms mesh %.3f ms kernel %.3f ms sustained %.0f' % (b['value'], b['flow_ms_per_step'], b['mesh_ms_per_step'], (b['roofline'].get('pruned') or b['roofline']).get('avg_launch_ms', -1), b['sustained']['mpix_s']))"
done; done
