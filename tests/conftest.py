import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run via gpurun)')


@pytest.fixture(scope='session')
def golden():
  import numpy as np

  def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)

  return load


@pytest.fixture(scope='session')
def gpu():
  """Loud failure when a gpu-marked test runs without the HIP path."""
  import torch
  from sofima_amd import _abi
  _abi.load()
  assert torch.cuda.is_available(), 'gpu tests need a visible MI355X'
  return torch.device('cuda', 0)


# SFM_TOL_REPORT=<file>: every np.testing.assert_allclose of the run appends the
# slack it had -- call site, the largest |got - want| / (atol + rtol |want|) (1.0
# = at the limit), the largest absolute and relative deviation -- so tolerances
# are set from measurements (tools/measure/tolerance_report.py summarises it).
if os.environ.get('SFM_TOL_REPORT'):
  import json
  import traceback

  import numpy as np

  _orig_allclose = np.testing.assert_allclose

  def _logged_allclose(actual, desired, rtol=1e-7, atol=0, *args, **kwargs):
    try:
      a = np.asarray(actual, dtype=np.float64)
      d = np.asarray(desired, dtype=np.float64)
      a, d = np.broadcast_arrays(a, d)
      ok = np.isfinite(a) & np.isfinite(d)
      if ok.any():
        diff = np.abs(a[ok] - d[ok])
        lim = atol + rtol * np.abs(d[ok])
        with np.errstate(divide='ignore', invalid='ignore'):
          used = float(np.max(np.where(lim > 0, diff / lim, np.where(diff > 0, np.inf, 0))))
          rel = float(np.max(np.where(d[ok] != 0, diff / np.abs(d[ok]), 0)))
        site = next((f for f in reversed(traceback.extract_stack()[:-1])
                     if '/tests/' in f.filename), None)
        rec = {'site': f'{os.path.basename(site.filename)}:{site.lineno}' if site else '?',
               'test': os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0],
               'rtol': rtol, 'atol': atol, 'used': used, 'max_abs': float(diff.max()),
               'max_rel': rel, 'scale': float(np.abs(d[ok]).max()), 'n': int(ok.sum())}
        with open(os.environ['SFM_TOL_REPORT'], 'a') as f:
          f.write(json.dumps(rec) + '\n')
    except Exception:   # the report must never change a test's outcome
      pass
    return _orig_allclose(actual, desired, rtol, atol, *args, **kwargs)

  np.testing.assert_allclose = _logged_allclose
