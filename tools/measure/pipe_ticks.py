"""Phase ticks (timing build) of one whole-pair launch: SFM_MFMA_PIPE from the environment.
  SOFIMA_AMD_LIB=sofima_amd/lib/libsofima_amd_timing.so SFM_MFMA_PIPE=1 python tools/measure/pipe_ticks.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np, torch
from sofima_amd import flow_field
from bench import synth_pair, WARP
pre, post = synth_pair(8192, 1002, warp=WARP)
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
f = calc.flow_field(a, b, 160, 40, batch_size=1024); torch.cuda.synchronize()
