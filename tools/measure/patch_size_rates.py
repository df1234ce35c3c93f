"""Hardware / algorithmic rate of the correlation kernel by patch size (pruned and un-pruned)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
from sofima_amd import flow_field as ff, _abi
from tests.util import em_texture
rng = np.random.default_rng(3)
lib = _abi.load()
for (h, w, P, S, B) in ((4096, 4096, 120, 20, 1024), (4096, 4096, 128, 20, 1024), (4096, 4096, 112, 20, 1024), (4096, 4096, 96, 20, 1024), (4096, 4096, 80, 20, 1024), (8192, 8192, 160, 40, 1024)):
  base = em_texture(rng, (h + 16, w + 16))
  pre = torch.from_numpy(np.ascontiguousarray(base[8:8 + h, 8:8 + w])).cuda()
  post = torch.from_numpy(np.ascontiguousarray(base[10:10 + h, 5:5 + w])).cuda()
  calc = ff.JAXMaskedXCorrWithStatsCalculator()
  for prune in (1, 0):
    with _abi.option('SFM_MFMA_PRUNE', prune):
      calc.flow_field(pre, post, P, S, batch_size=B); torch.cuda.synchronize()
      pf = _abi.SfmProfile(); lib.sfm_profile_read(C.byref(pf)); lib.sfm_profile_enable(1)
      n = 3
      t = time.perf_counter()
      for _ in range(n): out = calc.flow_field(pre, post, P, S, batch_size=B)
      torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
      lib.sfm_profile_enable(0); lib.sfm_profile_read(C.byref(pf))
    npatch = out.shape[1] * out.shape[2]
    alg = 2.0 * P**4 * npatch
    issued = pf.mfma_issued[0] * 32768.0 / n
    print('P=%d prune=%d: %d patches, wall %.2f ms, kernel %.2f ms -> %.0f TOP/s alg (%.3f), issued %.0f TOP/s (%.3f), issued/alg %.2f, clk %.0f' % (
      P, prune, npatch, dt * 1e3, pf.kernel_ms[0] / n, alg / (pf.kernel_ms[0] / n * 1e-3) / 1e12, alg / (pf.kernel_ms[0] / n * 1e-3) / 5e15,
      issued / (pf.kernel_ms[0] / n * 1e-3) / 1e12, issued / (pf.kernel_ms[0] / n * 1e-3) / 5e15, issued / alg, pf.clock_mhz[0]), flush=True)
