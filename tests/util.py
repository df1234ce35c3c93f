"""Shared comparison helpers for the parity tests."""
import json
import types

import numpy as np


def check_flow(got, want, sharp_rtol=2e-4, ratio_rtol=1e-4, ratio_atol=1e-6):
  """Flow-field parity: NaN pattern identical, vector components exact,
  sharpness / ratio within float32 tolerance."""
  assert got.shape == want.shape
  assert got.dtype == np.float32
  nd = got.shape[0] - 2
  np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
  np.testing.assert_array_equal(got[:nd], want[:nd])
  ok = np.isfinite(want[nd]) & np.isfinite(got[nd])
  np.testing.assert_array_equal(np.isfinite(want[nd]), np.isfinite(got[nd]))
  np.testing.assert_allclose(got[nd][ok], want[nd][ok], rtol=sharp_rtol)
  m = ~np.isnan(want[nd + 1])
  np.testing.assert_allclose(got[nd + 1][m], want[nd + 1][m], rtol=ratio_rtol,
                             atol=ratio_atol)


def cfg_from(d, cls=None):
  d = dict(d)
  d.pop('_force_cap', None)
  d['stride'] = tuple(d['stride'])
  if cls is None:
    return types.SimpleNamespace(**d)
  return cls(**d)


def load_cfgs(npz):
  return json.loads(str(npz['cfgs']))
