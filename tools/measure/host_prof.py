import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import flow_field
from bench import synth_pair
pre, post = synth_pair(8192, 5)
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
for _ in range(2): calc.flow_field(a, b, 160, 40, batch_size=1024)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
for _ in range(10): calc.flow_field(a, b, 160, 40, batch_size=1024)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
pr.disable()
print('pair %.3f ms' % (dt * 1e3))
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
