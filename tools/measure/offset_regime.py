import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scipy import ndimage
from sofima_amd import flow_field
from oracle import flow_oracle
rng = np.random.default_rng(2)
base = ndimage.gaussian_filter(rng.standard_normal((4200, 420)), 2.0)
base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
pre = base[20:4116, 20:320]; post = base[31:4127, 14:314]       # 4096 x 300 overlaps
pm = ndimage.maximum_filter(pre, 7) - ndimage.minimum_filter(pre, 7) < 10
qm = ndimage.maximum_filter(post, 7) - ndimage.minimum_filter(post, 7) < 10
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
for name, kw in (('unmasked', {}), ('masked', dict(pre_mask=pm, post_mask=qm))):
  f = calc.flow_field(pre, post, pre.shape, 1, batch_size=1, **kw); torch.cuda.synchronize()
  t = time.perf_counter(); f = calc.flow_field(pre, post, pre.shape, 1, batch_size=1, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t
  print('whole-overlap 4096x300 %s: %.1f ms, flow %s' % (name, dt * 1e3, f[:, 0, 0]))
t = time.perf_counter(); w = flow_oracle.flow_field(pre, post, pre.shape, 1, batch_size=1); print('oracle (CPU FFT) %.0f ms' % ((time.perf_counter() - t) * 1e3), w[:, 0, 0])
