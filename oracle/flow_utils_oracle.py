"""CPU oracle for the flow clean-up step (TEST INFRASTRUCTURE ONLY).

NumPy / SciPy restatement of `flow_utils.clean_flow` of the reference
(flow_utils.py:37-78).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may import this module; the product path never does.

Pinned against: the reference's own known-answer test
(tests/flow_utils_test.py:38-64, re-typed in tests/test_reference_kats.py) and
tests/golden/clean_flow.npz, produced by running the reference's clean_flow
(pure NumPy / SciPy, no stand-in needed) in tests/golden/make_golden.py.
"""
import numpy as np
from scipy import ndimage


def clean_flow(flow, min_peak_ratio, min_peak_sharpness, max_magnitude,
               max_deviation, dim=2):
  """flow_utils.py:37-78: NaN out vectors that fail the quality criteria."""
  flow = np.asarray(flow)
  assert dim in (2, 3)
  assert dim <= flow.shape[0] <= dim + 2
  with np.errstate(invalid='ignore'):
    if flow.shape[0] == dim + 2:
      # flow_utils.py:61-65
      bad = np.abs(flow[dim]) < min_peak_sharpness
      pr = np.abs(flow[dim + 1])
      bad |= (pr > 0.0) & (pr < min_peak_ratio)
    else:
      bad = np.zeros(flow[0].shape, dtype=bool)
    vec = flow[:dim]
    if max_magnitude > 0:  # flow_utils.py:70-71
      bad |= np.max(np.abs(vec), axis=0) > max_magnitude
    if max_deviation > 0:  # flow_utils.py:73-76
      size = (1, 1, 3, 3) if dim == 2 else (1, 3, 3, 3)
      med = ndimage.median_filter(np.nan_to_num(vec), size=size)
      bad |= np.max(np.abs(med - vec), axis=0) > max_deviation
  ret = vec.copy()
  ret[:, bad] = np.nan
  return ret
