#!/bin/bash
# Round-5 checkpoint: GPU tests, bench line, kernel trace of the production launches.
# Run on the GPU box:  gpurun -- bash tools/measure/check_round5.sh [tag]
R=$GRAFT_REPO_ROOT; T=${1:-a}; O=$R/gpurun_out/r5$T; mkdir -p $O
cd $R
SHA=$(python -c "from sofima_amd import _build; print(_build.source_hash())")
echo $SHA > $O/build_sha.txt
timeout 2700 python -m pytest tests -m gpu -q -rf --tb=line -x 2>&1 | grep -E "^FAILED|passed|failed|error|Error" | tail -12 > $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 2 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-legs --sustain 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o r5 -- $B --steps 3 --warmup 1 > $O/trace.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/trace -name '*.db' | head -1) > $O/trace_summary.md 2>&1
find $O -name '*.db' -size +20M -delete; find $O -name '*.csv' -size +5M -delete
cat $O/pytest.log; head -c 1200 $O/bench.json; tail -3 $O/bench.err; head -30 $O/trace_summary.md
