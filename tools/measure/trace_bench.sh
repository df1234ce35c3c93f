#!/bin/bash
# kernel trace of the bench (3 steps) -> gpurun_out/tb/summary.md
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tb; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --mesh-iters 10 > $O/trace.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/trace -name '*.db' | head -1) > $O/summary.md 2>&1
rm -rf $O/trace
head -8 $O/summary.md
