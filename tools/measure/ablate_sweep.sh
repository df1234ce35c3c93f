cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_nosweep -o t -- python $R/bench.py --no-cpu-baseline --no-legs --sustain 0 --steps 3 --warmup 1 > $O/trace_nosweep.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/trace_nosweep -name '*.db' | head -1) | head -8
find $O -name '*.db' -delete
