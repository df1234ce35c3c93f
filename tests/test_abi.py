"""The C-ABI library loads on a CPU-only box and exports what the header says."""
import ctypes
import os
import re

import pytest

from sofima_amd import _abi, _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'sofima_amd.h')


@pytest.fixture(scope='module')
def lib():
  if not os.path.exists(_abi.lib_path()):
    _build.build()
  return _abi.load()


def _declared_functions():
  text = open(HEADER).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(sfm_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree(lib):
  declared = _declared_functions()
  assert declared, 'no functions parsed from the header'
  assert sorted(_abi.SIGNATURES) == declared
  for name in declared:
    assert hasattr(lib, name), f'{name} is declared but not exported'
  # ... and the other way round: every C symbol the library exports is in the header
  # (C++ internals are mangled and not part of the boundary)
  import subprocess
  out = subprocess.run(['nm', '-D', '--defined-only', _abi.lib_path()], capture_output=True,
                       text=True, check=True).stdout
  exported = sorted({line.split()[-1] for line in out.splitlines()
                     if ' T ' in line and line.split()[-1].startswith('sfm_')})
  assert exported == declared, (set(exported) ^ set(declared))


def test_version_and_error_channel(lib):
  assert lib.sfm_version() == 8
  # A NULL descriptor is rejected with a message, not a crash.
  rc = lib.sfm_mesh_force(None, None)
  assert rc == -1
  assert b'NULL' in lib.sfm_last_error()
  rc = lib.sfm_xcorr_peaks(None, None)
  assert rc == -1
  n = ctypes.c_int(-5)
  assert lib.sfm_device_count(ctypes.byref(n)) == 0
  assert n.value >= 0


def test_option_table(lib):
  """Explicit switches win over the environment, which stays the default."""
  assert lib.sfm_set_option(b'NOT_OURS', b'1') == -1
  os.environ['SFM_TEST_SWITCH'] = 'env'
  try:
    assert _abi.get_option('SFM_TEST_SWITCH') == 'env'
    with _abi.option('SFM_TEST_SWITCH', 7):
      assert _abi.get_option('SFM_TEST_SWITCH') == '7'
      with _abi.option('SFM_TEST_SWITCH', 8):
        assert _abi.get_option('SFM_TEST_SWITCH') == '8'
      assert _abi.get_option('SFM_TEST_SWITCH') == '7'
    assert _abi.get_option('SFM_TEST_SWITCH') == 'env'
    os.environ['SFM_TEST_SWITCH'] = 'env2'          # still just the default
    assert _abi.get_option('SFM_TEST_SWITCH') == 'env2'
    _abi.set_option('SFM_TEST_SWITCH', 'x')
    assert _abi.get_option('SFM_TEST_SWITCH') == 'x'
    _abi.set_option('SFM_TEST_SWITCH', None)
    assert _abi.get_option('SFM_TEST_SWITCH') == 'env2'
  finally:
    del os.environ['SFM_TEST_SWITCH']
    _abi.set_option('SFM_TEST_SWITCH', None)
  assert _abi.get_option('SFM_TEST_SWITCH') is None


def test_struct_layouts_match_header():
  """Field order of the ctypes structs == field order in the header."""
  text = open(HEADER).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  for cname, cls in (('SfmXcorrDesc', _abi.SfmXcorrDesc),
                     ('SfmPeaksDesc', _abi.SfmPeaksDesc),
                     ('SfmMeshDesc', _abi.SfmMeshDesc),
                     ('SfmFireState', _abi.SfmFireState),
                     ('SfmChunkStats', _abi.SfmChunkStats),
                     ('SfmProfile', _abi.SfmProfile),
                     ('SfmMaskCountDesc', _abi.SfmMaskCountDesc),
                     ('SfmComposeDesc', _abi.SfmComposeDesc),
                     ('SfmCleanFlowDesc', _abi.SfmCleanFlowDesc),
                     ('SfmMaskIrregularDesc', _abi.SfmMaskIrregularDesc),
                     ('SfmRangeMaskDesc', _abi.SfmRangeMaskDesc),
                     ('SfmWarpDesc', _abi.SfmWarpDesc),
                     ('SfmNdWarpDesc', _abi.SfmNdWarpDesc),
                     ('SfmFlowStartsDesc', _abi.SfmFlowStartsDesc),
                     ('SfmFlowScatterDesc', _abi.SfmFlowScatterDesc),
                     ('SfmMeshShard', _abi.SfmMeshShard),
                     ('SfmBandedDesc', _abi.SfmBandedDesc),
                     ('SfmTargetMeshDesc', _abi.SfmTargetMeshDesc)):
    body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (cname, cname), text,
                     re.S).group(1)
    names = []
    for decl in body.split(';'):
      decl = decl.strip()
      if not decl:
        continue
      for part in decl.split(','):
        m = re.search(r'([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])*\s*$', part.strip())
        names.append(m.group(1))
    assert names == [f[0] for f in cls._fields_], cname


def test_no_gpu_means_loud_failure():
  import torch
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  import numpy as np
  from sofima_amd import flow_field, mesh
  with pytest.raises(_abi.SofimaAmdError):
    mesh.inplane_force(np.zeros((2, 1, 4, 4)), 0.1, (10, 10))
  with pytest.raises(_abi.SofimaAmdError):
    flow_field.JAXMaskedXCorrWithStatsCalculator().flow_field(
        np.zeros((64, 64), np.uint8), np.zeros((64, 64), np.uint8), 32, 16)


def test_product_code_never_imports_the_oracle():
  pkg = os.path.join(ROOT, 'sofima_amd')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(('.py', '.hip', '.h', '.cpp')):
        src = open(os.path.join(dirpath, f)).read()
        assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f


def test_integration_md_binding_matches_header():
  """The maintainer-side ctypes stub shown in INTEGRATION.md lists the fields of
  SfmXcorrDesc in the order of include/sofima_amd.h."""
  import os
  import re
  from sofima_amd import _abi
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  text = open(os.path.join(root, 'INTEGRATION.md')).read()
  block = text[text.index('class SfmXcorrDesc(C.Structure):'):]
  block = block[:block.index('lib.sfm_xcorr_workspace_bytes.restype')]
  names = re.findall(r"\('(\w+)',", block)
  assert names == [f[0] for f in _abi.SfmXcorrDesc._fields_]
