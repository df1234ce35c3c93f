# phase ticks of the correlation kernel, both forms, on the bench pair (timing library built by
# tools/measure/build_timing_lib.sh): gpurun -- bash tools/measure/pipe_ticks.sh [tag]
cd $GRAFT_REPO_ROOT; T=${1:-a}; O=gpurun_out/ticks_$T; mkdir -p $O
export SOFIMA_AMD_LIB=$GRAFT_REPO_ROOT/sofima_amd/lib/libsofima_amd_timing.so
for P in 0 1; do
  SFM_MFMA_PIPE=$P timeout 300 python tools/measure/pipe_ticks.py > $O/raw_pipe$P.txt 2>&1
  grep "^WG" $O/raw_pipe$P.txt | awk '{n++; p+=$10; w+=$12; c+=$14} END {print "WGs", n, "patches/WG", p/n, "wall ticks", w/n, "cycles", c/n}' > $O/ticks_pipe$P.txt
  grep "^wave\|^pipewave" $O/raw_pipe$P.txt >> $O/ticks_pipe$P.txt
  cat $O/ticks_pipe$P.txt
done
