#!/bin/bash
# kernel trace of the masked flow, blob case r = 130 -> gpurun_out/tm/summary.md
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tm; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MASK_CASE="${MASK_CASE:-blobs r=130}" timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/tools/measure/masked_time.py > $O/trace.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/trace -name '*.db' | head -1) > $O/summary.md 2>&1
rm -rf $O/trace
head -16 $O/summary.md
