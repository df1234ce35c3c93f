"""Per-kernel register / scratch / occupancy report of one HIP translation unit
(hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.

  python tools/measure/resource_report.py sofima_amd/csrc/sfm_mesh.hip [-ffp-contract=off ...]
"""
import re
import shutil
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
       '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'] + extra
text = subprocess.run(cmd, capture_output=True, text=True).stderr
filt = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
for b in re.split(r'remark: [^\n]*Function Name: ', text)[1:]:
  name = b.split()[0]
  if filt:
    name = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip()
  name = re.sub(r'\(anonymous namespace\)::', '', name)
  name = re.sub(r'\(.*', '', name)
  g = lambda k: re.search(k + r': (\d+)', b).group(1)
  print('%-52s SGPR %3s VGPR %3s AGPR %3s scratch %4s occ %s spill s/v %3s/%3s LDS %6s' % (
      name[:52], g('TotalSGPRs'), g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'),
      g(r'Occupancy \[waves/SIMD\]'), g('SGPRs Spill'), g('VGPRs Spill'),
      g(r'LDS Size \[bytes/block\]')))
