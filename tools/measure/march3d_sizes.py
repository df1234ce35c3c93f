"""Per-step time of integrate_kernel<3> against integrate_march3d_kernel over volume
sizes (where the default switches from one to the other: kMarch3dMinNodes).

  python tools/measure/march3d_sizes.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from march3d_ab import timed

for shape in [(3, 1, 64, 64, 64), (3, 8, 48, 48, 48), (3, 2, 80, 80, 80), (3, 4, 64, 64, 64),
              (3, 1, 100, 100, 100), (3, 1, 128, 128, 128), (3, 2, 100, 100, 100),
              (3, 3, 100, 100, 100), (3, 4, 100, 100, 100), (3, 1, 160, 160, 160),
              (3, 1, 200, 200, 200), (3, 1, 60, 300, 300)]:
  n = 1
  for s in shape[1:]:
    n *= s
  a, _ = timed(shape, False, iters=100)
  b, _ = timed(shape, True, iters=100)
  print('%-24s %8d nodes: per-node %.1f us per step, z-march %.1f (%.2fx)' %
        (shape, n, a, b, a / b), flush=True)
