R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/montprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/t -o m -- python $R/tools/measure/montage_time.py > $O/log.txt 2>&1
python $R/tools/rocpd_summary.py $(find $O/t -name '*.db' | head -1) 2>&1 | head -10 | cut -c1-150
find $O -name '*.db' -delete
