import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scipy import ndimage
from sofima_amd import flow_field
rng = np.random.default_rng(1)
img = ndimage.gaussian_filter(rng.standard_normal((2100, 2100)), 2).astype(np.float32)
a = torch.from_numpy(img[:2048, :2048].copy()).cuda(); b = torch.from_numpy(img[4:2052, 7:2055].copy()).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
for _ in range(3):
  f = calc.flow_field(a, b, 160, 40, batch_size=256); torch.cuda.synchronize()
