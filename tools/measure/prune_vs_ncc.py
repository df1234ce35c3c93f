"""Exact pruning vs data quality: flow time of an 8192^2 pair, pruned / un-pruned, as the
second image gets noisier (lower NCC peak) and with a smooth deformation instead of a
constant shift."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np, torch
from scipy import ndimage
from sofima_amd import flow_field as ff

size = 8192
rng = np.random.default_rng(1002)
m = 16
base = ndimage.gaussian_filter(rng.standard_normal((size + 2 * m, size + 2 * m), dtype=np.float32), 2.0)
base = (base - base.min()) / (base.max() - base.min()) * 255
pre = np.clip(np.round(base[m:m + size, m:m + size]), 0, 255).astype(np.uint8)
shifted = base[m + 3:m + 3 + size, m - 5:m - 5 + size]
yy, xx = np.mgrid[:size:8, :size:8].astype(np.float32)
calc = ff.JAXMaskedXCorrWithStatsCalculator()
pre_d = torch.from_numpy(pre).cuda()


def ncc_peak(a, b):
  a = a[4000:4160, 4000:4160].astype(np.float64); b = b[4000:4160, 4000:4160].astype(np.float64)
  a -= a.mean(); b -= b.mean()
  from scipy.signal import fftconvolve
  return fftconvolve(a, b[::-1, ::-1]).max() / np.sqrt((a * a).sum() * (b * b).sum())


def run(tag, post):
  post_d = torch.from_numpy(post).cuda()
  out = {}
  for mode in ('pruned', 'full'):
    if mode == 'full':
      os.environ['SFM_MFMA_PRUNE'] = '0'
    else:
      os.environ.pop('SFM_MFMA_PRUNE', None)
    calc.flow_field(pre_d, post_d, 160, 40, batch_size=1024); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
      f = calc.flow_field(pre_d, post_d, 160, 40, batch_size=1024)
    torch.cuda.synchronize()
    out[mode] = ((time.perf_counter() - t0) / 3 * 1e3, np.asarray(f))
  same = np.array_equal(out['pruned'][1], out['full'][1], equal_nan=True)
  print(f'{tag}: NCC peak {ncc_peak(pre, post):.2f}; pruned {out["pruned"][0]:.2f} ms, '
        f'un-pruned {out["full"][0]:.2f} ms ({out["pruned"][0] / out["full"][0]:.2f}x), identical {same}',
        flush=True)


for sigma in (4, 16, 32, 64):
  noisy = shifted + rng.standard_normal(shifted.shape, dtype=np.float32) * sigma
  run(f'constant shift, noise sigma {sigma}', np.clip(np.round(noisy), 0, 255).astype(np.uint8))
# smooth deformation of amplitude 6 px, wavelength 2048 px (SURVEY 8d "realistic pair"), noise 4
gy, gx = np.mgrid[:size, :size].astype(np.float32)
d = 6.0 * np.sin(2 * np.pi * gx / 2048) * np.cos(2 * np.pi * gy / 2048)
warped = ndimage.map_coordinates(base, [gy + m + d, gx + m - d], order=1, mode='nearest')
warped = warped + rng.standard_normal(warped.shape, dtype=np.float32) * 4
run('smooth deformation 6 px / 2048 px, noise sigma 4', np.clip(np.round(warped), 0, 255).astype(np.uint8))
