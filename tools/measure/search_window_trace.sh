# kernel trace of the search-window form: gpurun -- bash tools/measure/search_window_trace.sh [P]
R=$GRAFT_REPO_ROOT; P=${1:-240}; O=$R/gpurun_out/swtrace_$P; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o sw -- python $R/tools/measure/search_window_rates.py $P > $O/trace.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/trace -name '*.db' | head -1) > $O/trace_summary.md 2>&1
find $O -name '*.db' -size +20M -delete; find $O -name '*.csv' -size +5M -delete
grep "method" $O/trace.log; head -16 $O/trace_summary.md
