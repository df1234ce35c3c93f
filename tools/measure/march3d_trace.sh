#!/bin/bash
# kernel trace of one volume: integrate_kernel<3> against integrate_march3d_kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/march3d_prof; rm -rf $O; mkdir -p $O
MARCH_SHAPES=${MARCH_SHAPES:-0} MARCH_T=${MARCH_T:-0} rocprofv3 --kernel-trace --stats -d $O -o m -- python $R/tools/measure/march3d_ab.py time > $R/gpurun_out/march3d_trace.txt 2>&1
grep "us per step" $R/gpurun_out/march3d_trace.txt > $R/gpurun_out/march3d_trace_summary.txt
python $R/tools/rocpd_summary.py $(find $O -name '*.db' | head -1) 2>&1 | head -12 | cut -c1-200 >> $R/gpurun_out/march3d_trace_summary.txt
find $O -name '*.db' -delete
cat $R/gpurun_out/march3d_trace_summary.txt
