R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/meshprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 0 1 2; do
rocprofv3 --kernel-trace --stats -d $O/t$i -o m -- python $R/tools/measure/mesh_big.py $i > $O/log$i.txt 2>&1
python $R/tools/rocpd_summary.py $(find $O/t$i -name '*.db' | head -1) 2>&1 | head -8 | cut -c1-160
done
find $O -name '*.db' -delete
