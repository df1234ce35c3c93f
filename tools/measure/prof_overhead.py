"""Wall time of flow_field() on the bench pair with the kernel-timing hooks off / on."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np, torch
from bench import synth_pair
from sofima_amd import flow_field as ff, _abi
lib = _abi.load()
pre, post = synth_pair(8192, 1002)
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
calc = ff.JAXMaskedXCorrWithStatsCalculator()
prof = _abi.SfmProfile()
for tag, on in (('hooks off', 0), ('hooks on', 1), ('hooks off', 0), ('hooks on', 1)):
  lib.sfm_profile_enable(on)
  calc.flow_field(a, b, 160, 40, batch_size=1024); torch.cuda.synchronize()
  ts = []
  for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f = calc.flow_field(a, b, 160, 40, batch_size=1024)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
  lib.sfm_profile_read(C.byref(prof))
  print(f'{tag}: {np.median(ts) * 1e3:.2f} ms per call (synchronised), kernel {prof.kernel_ms[0] / max(prof.launches[0], 1):.2f} ms')
