#!/bin/bash
# End-of-round measurement (round 6): tests, bench, kernel trace, PMC passes (headline
# kernels + the aux legs), soak / fuzz logs, masked and mesh timings.
# Run on the GPU box:  gpurun -- bash tools/measure/profile_round6.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6p; rm -rf $O; mkdir -p $O
cd $R
SHA=$(python -c "from sofima_amd import _build; print(_build.source_hash())")
echo $SHA > $O/build_sha.txt
timeout 2400 python -m pytest tests -m gpu -q -rf --tb=line 2>&1 | grep -E "^FAILED|passed|failed|error" | tail -12 > $O/pytest.log
cd /tmp && export TMPDIR=/tmp
# production launches only in the traced / counted runs: no un-pruned legs, no sustained loop
B="python $R/bench.py --no-cpu-baseline --no-legs --sustain 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o r6 -- $B --steps 3 --warmup 1 > $O/trace.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/trace -name '*.db' | head -1) > $O/trace_summary.md 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_aux -o r6 -- python $R/bench.py --no-cpu-baseline --sustain 0 --steps 1 --warmup 1 > $O/trace_aux.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/trace_aux -name '*.db' | head -1) > $O/trace_aux_summary.md 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- $B --steps 1 --warmup 1 > $O/pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- $B --steps 1 --warmup 1 > $O/pmc_w.log 2>&1
python $R/tools/pmc_summary.py $(find $O/pmc_f -name '*counter_collection.csv' | head -1) $(find $O/pmc_w -name '*counter_collection.csv' | head -1) $O/pmc_traffic.json $SHA 40401 > $O/pmc_summary.log 2>&1
python - <<PY
import json
p = '$O/pmc_traffic.json'
d = json.load(open(p)); d['_meta']['pair'] = 'warped'; json.dump(d, open(p, 'w'), indent=1)
PY
for MODE in pruned full; do
  if [ $MODE = full ]; then export SFM_MFMA_PRUNE=0; fi
  timeout 600 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_$MODE -o sq -- $B --steps 1 --warmup 1 --mesh-iters 10 > $O/pmc_sq_$MODE.log 2>&1
  python - <<PY > $O/pmc_sq_$MODE.txt 2>&1
import csv, collections, glob
f = glob.glob('$O/pmc_sq_$MODE/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
  a = agg[r['Kernel_Name'][:60]][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, v in agg.items():
  print(k, {c: (n, s / n) for c, (n, s) in v.items()})
PY
done
unset SFM_MFMA_PRUNE
# HBM traffic of the aux legs (one counted process per leg)
bash $R/tools/measure/pmc_aux.sh > $O/pmc_aux.log 2>&1
cp $R/gpurun_out/pmc_aux/traffic_aux.json $O/pmc_traffic_aux.json
# the bench line LAST, with both traffic files of THIS build in place
cp $O/pmc_traffic.json $R/profiles/r06_pmc_traffic.json
cp $O/pmc_traffic_aux.json $R/profiles/r06_pmc_traffic_aux.json
cd $R
timeout 900 python bench.py --steps 20 --warmup 2 > $O/bench.json 2> $O/bench.err
python $R/tools/measure/mesh_time.py 2>&1 | grep -v amdgpu > $O/mesh_time.log
python $R/tools/measure/montage_time.py > $O/montage_time.log 2>&1
python $R/tools/measure/masked_time.py 2>&1 | grep -v amdgpu > $O/masked_time.log
python $R/tools/measure/patch_size_rates.py > $O/patch_size_rates.txt 2>&1
python $R/tools/measure/search_window_rates.py 176 192 224 240 256 320 2>&1 | grep -v amdgpu > $O/search_window_rates.txt
DRIFT=1 python $R/tools/measure/montage3d_time.py 2>&1 | grep -v amdgpu > $O/montage3d_time.log
timeout 300 python $R/tools/measure/pipe_ab.py time 10 2>&1 | grep -v amdgpu > $O/pipe_ab.log
MARCH_SHAPES=0,1 MARCH_T=0 bash $R/tools/measure/march3d_trace.sh > /dev/null 2>&1; cp $R/gpurun_out/march3d_trace_summary.txt $O/march3d_trace_summary.txt
python $R/tools/measure/march3d_sizes.py 2>&1 | grep -v amdgpu > $O/march3d_sizes.txt
cd /tmp
(timeout 400 python $R/tools/measure/soak.py 200; timeout 200 python $R/tools/measure/prune_fuzz.py 120; timeout 200 python $R/tools/measure/march3d_fuzz.py 60) 2>&1 | grep -v amdgpu > $O/soak_fuzz.log
cd /tmp; MASK_CASE="blobs r=130" timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_masked -o t -- python $R/tools/measure/masked_time.py > $O/trace_masked.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/trace_masked -name '*.db' | head -1) > $O/trace_masked_summary.md 2>&1
find $O -name '*.db' -delete; find $O -name '*.csv' -size +5M -delete; rm -rf $O/trace_masked $R/gpurun_out/pmc_aux
cat $O/pytest.log; head -c 400 $O/bench.json; tail -3 $O/bench.err; cat $O/soak_fuzz.log
