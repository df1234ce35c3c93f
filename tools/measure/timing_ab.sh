# phase ticks of the correlation kernel (timing build), with and without the in-loop cold test
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/early
for e in 0 8; do
  SFM_MFMA_EARLY=$e bash tools/measure/timing_run.sh > gpurun_out/early/ticks_$e.txt 2>&1
  echo "EARLY=$e"; grep "^wave" gpurun_out/early/ticks_$e.txt | tail -8
  grep "^WG" gpurun_out/early/ticks_$e.txt | awk '{w+=$12; c+=$14; m+=$18; e+=$20; n++} END {print "WGs", n, "avg wall", w/n, "cycles", c/n, "mfma/patch", m/n, "epi/patch", e/n}'
done
