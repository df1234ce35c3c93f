import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scipy import ndimage
from sofima_amd import flow_field
rng = np.random.default_rng(1)
vol = ndimage.gaussian_filter(rng.standard_normal((320, 320, 320)), 1.5)
vol = ((vol - vol.min()) / (vol.max() - vol.min()) * 255).astype(np.uint8)
pre = torch.from_numpy(vol[:300, :300, :300].copy()).cuda(); post = torch.from_numpy(vol[3:303, 2:302, 5:305].copy()).cuda()
for m, name in ((0, 'auto(FFT)'), (1, 'direct')):
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator(method=m)
  sub = slice(None) if m == 0 else slice(0, 160)
  a, b = pre[sub, sub, sub], post[sub, sub, sub]
  f = calc.flow_field(a, b, (80, 80, 80), 40, batch_size=16); torch.cuda.synchronize()
  t = time.perf_counter(); f = calc.flow_field(a, b, (80, 80, 80), 40, batch_size=16); torch.cuda.synchronize(); dt = time.perf_counter() - t
  n = f[0].size
  print('3-D 80^3 %s: %d patches, %.1f ms, %.3f ms/patch, flow x unique %s' % (name, n, dt * 1e3, dt * 1e3 / n, np.unique(f[0])))
# large float 2-D patches
img = ndimage.gaussian_filter(rng.standard_normal((2100, 2100)), 2).astype(np.float32)
a = torch.from_numpy(img[:2048, :2048].copy()).cuda(); b = torch.from_numpy(img[4:2052, 7:2055].copy()).cuda()
for m, name in ((0, 'auto(FFT)'), (1, 'direct')):
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator(method=m)
  f = calc.flow_field(a, b, 160, 40, batch_size=256); torch.cuda.synchronize()
  t = time.perf_counter(); f = calc.flow_field(a, b, 160, 40, batch_size=256); torch.cuda.synchronize(); dt = time.perf_counter() - t
  print('2-D float 160^2 %s: %d patches %.1f ms, %.2f us/patch' % (name, f[0].size, dt * 1e3, dt * 1e6 / f[0].size))
