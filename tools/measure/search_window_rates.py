"""Search-window mode (pre patch P > post patch Q, processor/flow.py:577,792-803): time per patch of the
FFT form and of the matrix-core form, by pre-patch size.  python tools/measure/search_window_rates.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import flow_field as ff, _abi
from tests.util import em_texture
rng = np.random.default_rng(3)
h = w = 4096
base = em_texture(rng, (h + 16, w + 16))
pre = torch.from_numpy(np.ascontiguousarray(base[8:8 + h, 8:8 + w])).cuda()
post = torch.from_numpy(np.ascontiguousarray(base[10:10 + h, 5:5 + w])).cuda()
Q, S, B = 160, 40, 1024
sizes = [int(s) for s in sys.argv[1:]] or [160, 192, 240, 320]
for P in sizes:
  res = {}
  for method in (3, 2):
    calc = ff.JAXMaskedXCorrWithStatsCalculator(method=method)
    try:
      out = calc.flow_field(pre, post, P, S, batch_size=B, post_patch_size=Q)
    except Exception as e:   # not eligible
      print(f'P={P} Q={Q} method={method}: {type(e).__name__}: {str(e)[:120]}', flush=True)
      continue
    torch.cuda.synchronize()
    n = 3
    t = time.perf_counter()
    for _ in range(n):
      out = calc.flow_field(pre, post, P, S, batch_size=B, post_patch_size=Q)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    npatch = out.shape[1] * out.shape[2]
    res[method] = out
    alg = 2.0 * P * P * Q * Q * npatch
    print(f'P={P} Q={Q} method={method}: {npatch} patches, {dt * 1e3:.2f} ms per call, '
          f'{dt / npatch * 1e6:.3f} us per patch, {alg / dt / 1e12:.0f} TOP/s algorithmic', flush=True)
  if 2 in res and 3 in res:
    a, b = res[2], res[3]
    same = np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[:2][~np.isnan(a[:2])], b[:2][~np.isnan(b[:2])])
    print(f'P={P}: flow vectors of the two forms identical: {same}; speed-up '
          f'(see the lines above)', flush=True)
