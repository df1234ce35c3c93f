# kernel trace + host profile of one configs[2] strip pair (4096 x 400, patch 120, step 20)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/measure/strip_prof.py 2>&1 | grep -v amdgpu | head -30 > $O/host.txt
rocprofv3 --kernel-trace --stats -d $O/t -o s -- python $R/tools/measure/strip_prof.py > $O/log.txt 2>&1
python $R/tools/rocpd_summary.py $(find $O/t -name '*.db' | head -1) 2>&1 | head -14 | cut -c1-170 > $O/summary.md
find $O -name '*.db' -delete
cat $O/host.txt | head -24; cat $O/summary.md
