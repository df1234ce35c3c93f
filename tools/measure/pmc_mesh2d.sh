#!/bin/bash
# Counters of integrate_shared2d_kernel on a [2,4,2048,2048] mesh (profiles/r03_pmc_mesh2d.md).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export ONLY=1 ITERS=10
M="python $R/tools/measure/mesh2d_xcd.py"
rm -rf /tmp/pm2d
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm2d/p1 -o a -- $M > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pm2d/p2 -o a -- $M > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pm2d/p3 -o a -- $M > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pm2d/p4 -o a -- $M > /dev/null 2>&1
python $R/tools/measure/pmc_agg.py $(find /tmp/pm2d -name '*counter_collection.csv')
