cd $GRAFT_REPO_ROOT; O=gpurun_out/pipe_c; mkdir -p $O
for A in 2 3 4 5; do
  SFM_MFMA_PIPE_ADMIT=$A timeout 200 python tools/measure/pipe_ab.py time 10 > $O/time_admit$A.log 2>&1; echo admit $A rc=$?; grep "round [12]" $O/time_admit$A.log
done
export SFM_MFMA_PIPE_ADMIT=4
bash tools/measure/pipe_ticks.sh c4 > $O/ticks.log 2>&1; grep -v "^wave" $O/ticks.log
