#!/bin/bash
# Timing build of the correlation kernels (clock64 phase ticks + device printf) as a SECOND
# library beside the production one: sofima_amd/lib/libsofima_amd_timing.so
# (select it with SOFIMA_AMD_LIB=...; built here, on the CPU box -- it travels with gpurun).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
python -c "from sofima_amd import _build; _build.build()"
O=$R/sofima_amd/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DSFM_MFMA_TIMING $SFM_MFMA_FLAGS \
  -c $R/sofima_amd/csrc/sfm_xcorr_mfma.hip -o $O/sfm_xcorr_mfma_timing.o
OBJS=$(ls $O/*.hip.o | grep -v sfm_xcorr_mfma.hip.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/sofima_amd/lib/libsofima_amd_timing.so \
  $OBJS $O/sfm_xcorr_mfma_timing.o -L/opt/rocm/lib -ldl -Wl,-rpath,/opt/rocm/lib
ls -la $R/sofima_amd/lib/
