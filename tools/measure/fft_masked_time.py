"""Masked volumetric flow (80^3 patches, FFT form): hand-written transforms vs hipFFT plans."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scipy import ndimage
from sofima_amd import flow_field
rng = np.random.default_rng(1)
vol = ndimage.gaussian_filter(rng.standard_normal((250, 250, 250)), 1.5)
vol = ((vol - vol.min()) / (vol.max() - vol.min()) * 255).astype(np.uint8)
pre = torch.from_numpy(vol[:240, :240, :240].copy()).cuda(); post = torch.from_numpy(vol[3:243, 2:242, 5:245].copy()).cuda()
pm = torch.from_numpy(rng.random((240, 240, 240)) < 0.02).cuda(); qm = torch.from_numpy(rng.random((240, 240, 240)) < 0.02).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
for env in ('1', '0'):
  os.environ['SFM_FFT_OWN'] = env
  run = lambda: calc.flow_field(pre, post, (80, 80, 80), 40, pre_mask=pm, post_mask=qm, batch_size=8,
                                mask_only_for_patch_selection=False)
  f = run(); torch.cuda.synchronize()
  t = time.perf_counter(); f = run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
  n = f[0].size
  print('masked 3-D 80^3 %s: %d patches, %.1f ms, %.3f ms/patch, flow x %s' % (
      'hand-written' if env == '1' else 'hipFFT plans', n, dt * 1e3, dt * 1e3 / n, np.unique(f[0])))
os.environ.pop('SFM_FFT_OWN')
