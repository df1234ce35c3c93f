"""Builds libsofima_amd.so (HIP, gfx950) and the C oracle helpers in-tree."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB_DIR = os.path.join(PKG_DIR, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libsofima_amd.so')
OBJ_DIR = os.path.join(PKG_DIR, 'build')

# Translation units and their extra flags.  The mesh kernels follow the
# reference's f32 operation order, so FMA contraction is disabled there.
SOURCES = {
    'sfm_core.hip': [],
    'sfm_mesh.hip': ['-ffp-contract=off'] + os.environ.get('SFM_MESH_FLAGS', '').split(),
    'sfm_xcorr.hip': [],
    'sfm_xcorr_fft.hip': [],
    'sfm_fft_own.hip': [],
    # SFM_MFMA_TIMING / SFM_MFMA_FLAGS: instrumentation and tuning experiments
    'sfm_xcorr_mfma.hip': ((['-DSFM_MFMA_TIMING']
                            if os.environ.get('SFM_MFMA_TIMING') else []) +
                           os.environ.get('SFM_MFMA_FLAGS', '').split()),
    'sfm_maps.hip': ['-ffp-contract=off'] + os.environ.get('SFM_MAPS_FLAGS', '').split(),
    'sfm_flowutils.hip': ['-ffp-contract=off'],
    'sfm_comm.hip': [],
    'sfm_warp.hip': ['-ffp-contract=off'],
}
# SFM_BUILD_FLAGS: extra flags for every unit (-DSFM_MEASUREMENT_SWITCHES: the library
# honours the measurement-only switches, see csrc/sfm_common.h)
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
          '-Wno-unused-result'] + os.environ.get('SFM_BUILD_FLAGS', '').split()


def _hipcc() -> str:
  exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  if not os.path.exists(exe):
    raise RuntimeError('hipcc not found; cannot build libsofima_amd.so')
  return exe


def _stale(target: str, deps: list[str], cmd: list[str] | None = None) -> bool:
  """Out of date by mtime, or built with a different command line (the extra
  flags come from environment variables: an instrumented build must not
  silently stay the production library, nor the reverse)."""
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  if any(os.path.getmtime(d) > t for d in deps):
    return True
  if cmd is not None:
    try:
      with open(target + '.cmd') as f:
        return f.read() != ' '.join(cmd)
    except OSError:
      return True
  return False


def _record(target: str, cmd: list[str]) -> None:
  with open(target + '.cmd', 'w') as f:
    f.write(' '.join(cmd))


def build(force: bool = False, verbose: bool = False) -> str:
  """Compiles every HIP translation unit for gfx950 and links the library."""
  os.makedirs(LIB_DIR, exist_ok=True)
  os.makedirs(OBJ_DIR, exist_ok=True)
  headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
             if f.endswith('.h')]
  headers.append(os.path.join(PKG_DIR, '..', 'include', 'sofima_amd.h'))
  hipcc = _hipcc()
  objs = []
  relink = force
  for src, extra in SOURCES.items():
    src_path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ_DIR, src + '.o')
    objs.append(obj)
    cmd = [hipcc] + COMMON + extra + ['-c', src_path, '-o', obj]
    if force or _stale(obj, [src_path] + headers, cmd):
      if verbose:
        print(' '.join(cmd))
      subprocess.run(cmd, check=True)
      _record(obj, cmd)
      relink = True
  if relink or _stale(LIB_PATH, objs):
    # no library dependency beyond the HIP runtime (RCCL is resolved with dlsym)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH
           ] + objs + ['-L/opt/rocm/lib', '-ldl', '-Wl,-rpath,/opt/rocm/lib']
    if verbose:
      print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    _stamp()
  return LIB_PATH


def source_hash() -> str:
  """Hash of the kernel sources (csrc/ + the C header): names the BUILD that
  measurements belong to.  Unlike a git revision it survives documentation
  commits and rebuilds of unchanged sources."""
  import hashlib
  h = hashlib.sha256()
  files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                 if f.endswith(('.hip', '.h')))
  files.append(os.path.join(PKG_DIR, '..', 'include', 'sofima_amd.h'))
  for path in files:
    h.update(os.path.basename(path).encode())
    with open(path, 'rb') as f:
      h.update(f.read())
  return h.hexdigest()[:12]


def _stamp() -> None:
  """Records the source hash the library was linked from (.build_sha, not
  tracked): the profile scripts stamp their counter summaries with it.  The
  snapshot a GPU box receives has no .git, so the file travels with the .so."""
  try:
    with open(os.path.join(PKG_DIR, '..', '.build_sha'), 'w') as f:
      f.write(source_hash() + '\n')
  except OSError:
    pass


if __name__ == '__main__':
  print(build(force=False, verbose=True))
