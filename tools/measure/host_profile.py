import sys, os, time, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from sofima_amd import flow_field, mesh
from bench import synth_pair, WARP
pre, post = synth_pair(8192, 0, warp=WARP)
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
for _ in range(3): f = calc.flow_field(a, b, 160, 40, batch_size=1024)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): f = calc.flow_field(a, b, 160, 40, batch_size=1024)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
