# correctness + A/B of the pipeline kernel: gpurun -- bash tools/measure/pipe_round.sh tag [admit values...]
cd $GRAFT_REPO_ROOT; T=${1:-a}; shift; O=gpurun_out/pipe_$T; mkdir -p $O
timeout 120 python tools/measure/pipe_smoke.py > $O/smoke.log 2>&1; echo smoke rc=$?; grep -c "identical True" $O/smoke.log; grep "identical False" $O/smoke.log | head -3
timeout 500 python tools/measure/pipe_ab.py check > $O/check.log 2>&1; echo check rc=$?; tail -2 $O/check.log
timeout 200 python tools/measure/pipe_ab.py time 10 > $O/time.log 2>&1; echo time rc=$?; tail -7 $O/time.log
for A in "$@"; do
  SFM_MFMA_PIPE_ADMIT=$A timeout 200 python tools/measure/pipe_ab.py time 10 > $O/time_admit$A.log 2>&1; echo admit $A rc=$?; grep "pipe=1" $O/time_admit$A.log
done
bash tools/measure/pipe_ticks.sh $T > $O/ticks.log 2>&1; grep -v "^wave" $O/ticks.log
