cd $GRAFT_REPO_ROOT
SFM_MFMA_TIMING=1 python -c "
from sofima_amd import _build; _build.build(force=True)"
python - <<PY 2>&1 | grep -v amdgpu.ids | tail -600
import sys, time, os; sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
from sofima_amd import flow_field
from bench import synth_pair
pre, post = synth_pair(4096, 5)
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
flow_field.LAUNCH_PATCHES = 4096
f = calc.flow_field(a, b, 160, 40, batch_size=1024); torch.cuda.synchronize()
PY
