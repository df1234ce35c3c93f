# phase ticks of the correlation kernel (timing build) on the bench workload:
# the whole 8192^2 warped pair in one launch (79 patches per workgroup)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/early
SFM_MFMA_TIMING=1 python -c "
from sofima_amd import _build; _build.build(force=True)"
python - > gpurun_out/early/ticks_full.txt 2>&1 <<PY
import sys, os; sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
from sofima_amd import flow_field
from bench import synth_pair, WARP
pre, post = synth_pair(8192, 0, warp=WARP)
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
f = calc.flow_field(a, b, 160, 40, batch_size=1024); torch.cuda.synchronize()
PY
grep "^wave" gpurun_out/early/ticks_full.txt | tail -4
grep "^WG" gpurun_out/early/ticks_full.txt | awk '{n++; p+=$10; w+=$12; c+=$14; m+=$18; e+=$20} END {print "WGs", n, "patches/WG", p/n, "wall ticks", w/n, "cycles", c/n, "mfma/patch", m/n, "epi/patch", e/n}'
