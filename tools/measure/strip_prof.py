import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import flow_field
from bench import synth_pair
pre, post = synth_pair(4096, 5)
a = torch.from_numpy(np.ascontiguousarray(pre[:, :400])).cuda(); b = torch.from_numpy(np.ascontiguousarray(post[:, :400])).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
for _ in range(3): calc.flow_field(a, b, 120, 20, batch_size=256)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
for _ in range(50): calc.flow_field(a, b, 120, 20, batch_size=256)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
pr.disable()
print('strip pair %.3f ms' % (dt * 1e3))
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
