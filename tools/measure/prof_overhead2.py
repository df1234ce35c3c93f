"""Flow of the bench pair back to back with the library's timing hooks off / on (what bench.py's timed region pays for its own instrumentation)."""
import sys, time, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import flow_field, _abi
import bench
pre, post = bench.synth_pair(8192, 1002, warp=bench.WARP)
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
lib = _abi.load()
prof = _abi.SfmProfile()
for rep in range(2):
  for on in (0, 1):
    lib.sfm_profile_read(C.byref(prof)); lib.sfm_profile_enable(on)
    for _ in range(3): calc.flow_field(a, b, 160, 40, batch_size=1024)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f = calc.flow_field(a, b, 160, 40, batch_size=1024)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20 * 1e3
    lib.sfm_profile_enable(0); lib.sfm_profile_read(C.byref(prof))
    print('hooks %d: %.3f ms per pair' % (on, dt), ('(kernel %.3f ms)' % (prof.kernel_ms[0] / max(1, prof.launches[0]))) if on else '')
