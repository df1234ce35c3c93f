#!/bin/bash
# HBM traffic of the aux_rooflines legs of bench.py: one counted process per leg,
# FETCH_SIZE and WRITE_SIZE in separate passes -> gpurun_out/pmc_aux/traffic_aux.json
# (copy to profiles/r06_pmc_traffic_aux.json).   gpurun -- bash tools/measure/pmc_aux.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_aux; rm -rf $O; mkdir -p $O
SHA=$(cd $R && python -c "from sofima_amd import _build; print(_build.source_hash())")
cd /tmp && export TMPDIR=/tmp
for LEG in xcorr_fft_2d xcorr_fft_3d mesh_3d mesh_2d_large mesh_2d_montage_size; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $O/$LEG.$C -o c -- python $R/bench.py --aux-leg $LEG > $O/$LEG.$C.log 2>&1
  done
  CALLS=$(grep -o '"calls_in_counted_process": [0-9]*' $O/$LEG.FETCH_SIZE.log | grep -o '[0-9]*$' | head -1)
  python $R/tools/pmc_leg_summary.py $LEG ${CALLS:-1} $(find $O/$LEG.FETCH_SIZE -name '*counter_collection.csv' | head -1) \
      $(find $O/$LEG.WRITE_SIZE -name '*counter_collection.csv' | head -1) $O/traffic_aux.json $SHA
done
find $O -name '*.csv' -size +5M -delete; find $O -name '*.db' -delete
