cd $GRAFT_REPO_ROOT
run() { python - <<PY 2>&1 | tail -1
import sys, time, os; sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
from sofima_amd import flow_field
from bench import synth_pair
pre, post = synth_pair(8192, 5)
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
calc.flow_field(a, b, 160, 40, batch_size=1024); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(4): calc.flow_field(a, b, 160, 40, batch_size=1024)
torch.cuda.synchronize()
print('%.2f ms' % ((time.perf_counter() - t) / 4 * 1e3))
PY
}
echo -n "baseline: "; run
SFM_MFMA_FLAGS="-DSFM_ABLATE_EPILOGUE" python -c "
from sofima_amd import _build; import os; os.utime('sofima_amd/csrc/sfm_xcorr_mfma.hip'); _build.build()"
echo -n "no epilogue/hot: "; run
