#!/bin/bash
# Measurement build of the library beside the production one:
#   sofima_amd/lib/libsofima_amd_timing.so   (select it with SOFIMA_AMD_LIB=...)
# = the correlation unit with clock64 phase ticks + device printf (-DSFM_MFMA_TIMING) and the
# measurement-only switches (-DSFM_MEASUREMENT_SWITCHES: SFM_MFMA_PROBE / TOUCH_ALL / EXACT /
# QUEUE / PRIO / MAX_WG_PER_CU, which the production library ignores).  Built here, on the CPU
# box -- it travels with gpurun.  NOTIMING=1: the switches without the ticks
# (libsofima_amd_measure.so: production timing behaviour, for A/B runs and the identity tests
# of those switches).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
python -c "from sofima_amd import _build; _build.build()"
O=$R/sofima_amd/build
T=-DSFM_MFMA_TIMING; NAME=timing
if [ -n "$NOTIMING" ]; then T=; NAME=measure; fi
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DSFM_MEASUREMENT_SWITCHES"
hipcc $F $T $SFM_MFMA_FLAGS -c $R/sofima_amd/csrc/sfm_xcorr_mfma.hip -o $O/sfm_xcorr_mfma_$NAME.o
hipcc $F -c $R/sofima_amd/csrc/sfm_core.hip -o $O/sfm_core_$NAME.o
OBJS=$(ls $O/*.hip.o | grep -v "sfm_xcorr_mfma.hip.o\|sfm_core.hip.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/sofima_amd/lib/libsofima_amd_$NAME.so \
  $OBJS $O/sfm_xcorr_mfma_$NAME.o $O/sfm_core_$NAME.o -L/opt/rocm/lib -ldl -Wl,-rpath,/opt/rocm/lib
ls -la $R/sofima_amd/lib/
