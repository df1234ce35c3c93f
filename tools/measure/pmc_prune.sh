#!/bin/bash
# SQ counters of the correlation kernel with and without pruning -> gpurun_out/pp/*.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pp; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
for mode in pruned full; do
  if [ $mode = full ]; then export SFM_MFMA_PRUNE=0; else unset SFM_MFMA_PRUNE; fi
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $O/$mode -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --mesh-iters 10 > $O/$mode.log 2>&1
  python $R/tools/measure/pmc_kernel_summary.py xcorr_mfma $(find $O/$mode -name '*counter_collection.csv' | head -1) > $O/$mode.txt 2>&1
  rm -rf $O/$mode
done
cat $O/pruned.txt $O/full.txt
