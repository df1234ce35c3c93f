"""The inequalities behind the exact pruning of the correlation kernel
(DESIGN.md 1.3), restated in NumPy and checked against directly computed
surfaces: every bound the prep kernel emits must dominate |surface| on the
region it speaks for (tile + guard band).  CPU only."""
import numpy as np
import pytest
from scipy.signal import fftconvolve


def _surface(a, b):
  """out[dy + P - 1, dx + P - 1] = sum (a - mean)(b - mean) over the overlap:
  the un-masked surface of flow_field.py:78-89 for one patch pair."""
  a0 = a - a.mean()
  b0 = b - b.mean()
  return fftconvolve(a0, b0[::-1, ::-1])


def _energies(x):
  e = (x - x.mean()) ** 2
  return e


def _rows_of_shift(d, p):
  """overlap rows of operand A / B for a shift d (out[d] = sum a[y + d] b[y])"""
  if d >= 0:
    return slice(d, p), slice(0, p - d)
  return slice(0, p + d), slice(-d, p)


def _patches(kind, rng, p):
  if kind == 'noise':
    return (rng.integers(0, 256, (p, p)).astype(np.float64),
            rng.integers(0, 256, (p, p)).astype(np.float64))
  from scipy import ndimage
  base = ndimage.gaussian_filter(rng.standard_normal((p + 40, p + 40)), 2.0)
  base = np.round((base - base.min()) / (base.max() - base.min()) * 255)
  a = base[20:20 + p, 20:20 + p].copy()
  b = base[23:23 + p, 15:15 + p].copy()
  if kind == 'edges':
    a[: p // 2] = 90
    b[:, : p // 3] = 200
  return a, b


@pytest.mark.parametrize('kind', ['em', 'noise', 'edges'])
@pytest.mark.parametrize('p', [48, 80])
def test_row_column_and_block_bounds_dominate_the_surface(kind, p):
  rng = np.random.default_rng(p + len(kind))
  a, b = _patches(kind, rng, p)
  s = np.abs(_surface(a, b))
  ea, eb = _energies(a), _energies(b)
  guard = 10
  n_tiles = (2 * p - 1 + 15) // 16
  for t in range(n_tiles):
    lo, hi = 16 * t - (p - 1), min(16 * t + 15, 2 * p - 2) - (p - 1)
    d = 0
    if lo > 0:
      d = max(0, lo - guard)
    if hi < 0:
      d = min(0, hi + guard)
    ra, rb = _rows_of_shift(d, p)
    bound_rows = np.sqrt(ea[ra].sum() * eb[rb].sum())
    # region the bound speaks for: the tile's rows widened by the guard (towards
    # the centre it is covered by the nesting of the row sets)
    k0 = max(0, 16 * t - guard)
    k1 = min(2 * p - 1, 16 * t + 16 + guard)
    if lo > 0:
      k0 = max(k0, d + p - 1)
    if hi < 0:
      k1 = min(k1, d + p - 1 + 1)
    if lo <= 0 <= hi:
      continue  # the centre tile is never pruned (bound of the whole patch)
    assert s[k0:k1].max() <= bound_rows * (1 + 1e-9) + 1e-6
    # 2-D: the same rows x the outer column tiles, block energies rounded outwards
    for ks in (1, 2):
      for side in (0, 1):
        nq = (2 * p - 1 + 15) // 16
        dx = (min(0, 16 * ks - p + guard) if side == 0
              else max(0, 16 * (nq - ks) - (p - 1) - guard))
        ca, cb = _rows_of_shift(dx, p)   # same nesting along x
        def blocks(sl):
          return slice(sl.start // 16 * 16, min(p, (sl.stop + 15) // 16 * 16))
        if ca.stop <= ca.start or cb.stop <= cb.start:
          continue
        e2a = ea[blocks(ra), blocks(ca)].sum()
        e2b = eb[blocks(rb), blocks(cb)].sum()
        bound2 = np.sqrt(e2a * e2b)
        # columns the bound speaks for: |dx'| >= |dx| on that side
        cols = slice(0, dx + p) if side == 0 else slice(dx + p - 1, 2 * p - 1)
        if cols.stop <= cols.start:
          continue
        assert s[k0:k1, cols].max() <= bound2 * (1 + 1e-9) + 1e-6


def test_correction_bound_of_the_seed_probe():
  """|surface - S| <= |mA'| sqrt(N sum b'^2) + |mB'| sqrt(N sum a'^2) + |mA' mB'| N
  with a' = pixel - integer centre, S = sum a' b' over the overlap."""
  rng = np.random.default_rng(3)
  p = 40
  a, b = _patches('em', rng, p)
  ca, cb = np.round(a.mean()), np.round(b.mean())
  a1, b1 = a - ca, b - cb
  raw = fftconvolve(a1, b1[::-1, ::-1])
  full = _surface(a, b)
  ma, mb = a.mean() - ca, b.mean() - cb
  n = p * p
  bound = (abs(ma) * np.sqrt(n * (b1 ** 2).sum()) + abs(mb) * np.sqrt(n * (a1 ** 2).sum()) +
           abs(ma * mb) * n)
  assert np.abs(full - raw).max() <= bound + 1e-6
