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


# ---------------------------------------------------------------------------
# Round-4 inequalities ("cold tiles abandoned inside their row
# loop", "row loops that narrow themselves", "the guard band is counted in
# rows"): NumPy restatements of what sfm_xcorr_mfma.hip's check_after and the
# end-of-patch need mask rely on, against directly computed partial surfaces.
# ---------------------------------------------------------------------------
def _int_centre(x):
  """Integer centre of the int8 operands: clamp(round(mean), max - 127, min + 128)."""
  return float(np.clip(np.round(x.mean()), x.max() - 127, x.min() + 128))


def _corr(a, b):
  """out[dy + Q - 1, dx + Q - 1] = sum_y sum_x a[y + dy, x + dx] b[y, x]."""
  return fftconvolve(a, b[::-1, ::-1])


def _prefix4(e_rows):
  """Row-energy prefix sums the prep kernel stores at every fourth row
  (tbound[kRowPre + k] = sum over rows < min(4 k, rows))."""
  p = len(e_rows)
  cum = np.concatenate([[0.0], np.cumsum(e_rows)])
  return np.array([cum[min(4 * k, p)] for k in range(p // 4 + 2)])


def _rest_bound(pre_a, pre_b, p, dy0, y, yhi):
  """check_after's bound of what the patch rows y .. yhi - 1 can still add to ANY
  shift of the row tile that starts at dy0: operand rows of A that those rows
  meet under the tile's 16 shifts, prefixes rounded outwards."""
  a_lo, a_hi = max(0, y + dy0), min(p, yhi + dy0 + 15)
  ea = pre_a[(a_hi + 3) >> 2] - pre_a[a_lo >> 2]
  eb = pre_b[(yhi + 3) >> 2] - pre_b[y >> 2]
  return np.sqrt(max(ea, 0.0) * max(eb, 0.0))


def _corr_bound(a, b, a1, b1):
  n = a.size
  ma, mb = a.mean() - _int_centre(a), b.mean() - _int_centre(b)
  return (abs(ma) * np.sqrt(n * (b1 ** 2).sum()) + abs(mb) * np.sqrt(n * (a1 ** 2).sum()) +
          abs(ma * mb) * n)


@pytest.mark.parametrize('kind', ['em', 'noise', 'edges'])
@pytest.mark.parametrize('p', [48, 80])
def test_in_loop_rest_bound_dominates_what_the_remaining_rows_add(kind, p):
  """After the patch rows yb < y the accumulators hold exact partial sums; the
  rows y .. yhi - 1 add at most sqrt(E_A E_B) of the operand rows that are left
  to any of the tile's shifts, so max(partial) + rest + |correction|max bounds
  every FINAL element of the tile (the abandon test), and the same column tile
  by column tile (the narrowing test)."""
  rng = np.random.default_rng(7 * p + len(kind))
  a, b = _patches(kind, rng, p)
  a1, b1 = a - _int_centre(a), b - _int_centre(b)
  assert np.abs(a1).max() <= 128 and np.abs(b1).max() <= 128
  pre_a = _prefix4((a1 ** 2).sum(axis=1))
  pre_b = _prefix4((b1 ** 2).sum(axis=1))
  final_s = _corr(a1, b1)                    # exact integer sums S
  final = _surface(a, b)                     # S + mean correction: what is thresholded
  corr = _corr_bound(a, b, a1, b1)
  n_tiles = (2 * p - 1 + 15) // 16
  nq = n_tiles
  checked = 0
  for t in range(n_tiles):
    dy0 = 16 * t - (p - 1)
    ylo, yhi = max(0, -dy0 - 15), min(p, p - dy0)
    r0, r1 = 16 * t, min(16 * t + 16, 2 * p - 1)       # surface rows of the tile
    for y in range(ylo, yhi, 4):
      rest_rows = b1.copy()
      rest_rows[:y] = 0                                # rows the loop has visited
      remaining = _corr(a1, rest_rows)[r0:r1]
      partial = final_s[r0:r1] - remaining
      rest = _rest_bound(pre_a, pre_b, p, dy0, y, yhi)
      # (rows >= yhi never meet the patch under this tile's shifts)
      gone = b1.copy()
      gone[:yhi] = 0
      assert np.abs(_corr(a1, gone)[r0:r1]).max() < 1e-6
      assert np.abs(remaining).max() <= rest * (1 + 1e-9) + 1e-6
      ub = max(partial.max(), 0.0) + rest + corr       # the kernel's `ub` (+ its margins)
      assert final[r0:r1].max() <= ub * (1 + 1e-9) + 1e-6
      # narrowing: outer K + 1 column tiles on either side
      for k in range(0, nq // 2 - 1):
        cols = np.r_[0:16 * (k + 1), 16 * (nq - 1 - k):2 * p - 1]
        m_k = max(partial[:, cols].max(), 0.0)
        assert final[r0:r1][:, cols].max() <= (m_k + rest + corr) * (1 + 1e-9) + 1e-6
      checked += 1
    # before the first row group the bound is the tile's own a-priori bound
    assert _rest_bound(pre_a, pre_b, p, dy0, ylo, yhi) * (1 + 1e-9) + corr + 1e-6 >= \
        final[r0:r1].max()
  assert checked > 4 * n_tiles


def test_in_loop_bound_is_tight_enough_to_abandon_cold_tiles():
  """The bound is useful, not only valid: on an EM-like pair with a clear peak
  most non-central tiles fall below 0.5 x the maximum before their last row."""
  rng = np.random.default_rng(11)
  p = 80
  a, b = _patches('em', rng, p)
  a1, b1 = a - _int_centre(a), b - _int_centre(b)
  pre_a = _prefix4((a1 ** 2).sum(axis=1))
  pre_b = _prefix4((b1 ** 2).sum(axis=1))
  final_s = _corr(a1, b1)
  final = _surface(a, b)
  corr = _corr_bound(a, b, a1, b1)
  thr = 0.5 * final.max()
  abandoned = 0
  n_tiles = (2 * p - 1 + 15) // 16
  for t in range(n_tiles):
    dy0 = 16 * t - (p - 1)
    ylo, yhi = max(0, -dy0 - 15), min(p, p - dy0)
    r0, r1 = 16 * t, min(16 * t + 16, 2 * p - 1)
    for y in range(ylo, yhi - 8, 4):
      rest_rows = b1.copy()
      rest_rows[:y] = 0
      partial = final_s[r0:r1] - _corr(a1, rest_rows)[r0:r1]
      if max(partial.max(), 0.0) + _rest_bound(pre_a, pre_b, p, dy0, y, yhi) + corr < thr:
        abandoned += 1
        assert final[r0:r1].max() < thr               # ... and rightly so
        break
  assert abandoned >= n_tiles - 4, abandoned


@pytest.mark.parametrize('guard', [4, 10, 24, 60])
def test_row_counted_guard_band_covers_every_window_of_a_hot_element(guard):
  """A hot tile asks for the tiles its HOT ROWS reach with `guard` rows (lo_t =
  (16 t + r_lo - guard) >> 4, hi_t = (16 t + r_hi + guard) >> 4 with r_lo / r_hi
  the first / last hot row of the tile) instead of ceil(guard / 16) whole tiles
  on either side.  Every row within `guard` of a hot element must lie in a
  requested tile -- what stays un-stored is then farther than `guard` rows from
  every element above threshold_rel x the maximum."""
  rng = np.random.default_rng(guard)
  p = 80
  for kind in ('em', 'noise', 'edges'):
    a, b = _patches(kind, rng, p)
    s = _surface(a, b)
    n_tiles = (s.shape[0] + 15) // 16
    for thr_rel in (0.5, 0.2):
      hot = s > thr_rel * s.max()
      need = np.zeros(n_tiles, bool)
      for t in range(n_tiles):
        rows = np.nonzero(hot[16 * t:16 * t + 16].any(axis=1))[0]
        if rows.size == 0:
          continue
        lo_t = max(16 * t + rows.min() - guard, 0) >> 4
        hi_t = min((16 * t + rows.max() + guard) >> 4, n_tiles - 1)
        need[lo_t:hi_t + 1] = True
      for r in np.nonzero(hot.any(axis=1))[0]:
        band = np.arange(max(0, r - guard), min(s.shape[0], r + guard + 1))
        assert need[band >> 4].all(), (kind, thr_rel, r)
      # and it is a real saving against whole tiles either side of a hot tile
      g_tiles = (guard + 15) >> 4
      whole = np.zeros(n_tiles, bool)
      for t in np.nonzero([hot[16 * t:16 * t + 16].any() for t in range(n_tiles)])[0]:
        whole[max(0, t - g_tiles):t + g_tiles + 1] = True
      assert need.sum() <= whole.sum() and not (need & ~whole).any()


def _den_all_shifts(a, b):
  """Padfield denominator sqrt(SSD_A SSD_B) of a CLEAN same-size pair for every
  shift, exact (float64 on integers): SSD = n sum x^2 - (sum x)^2 over the
  overlap rectangle of each side, divided by n (flow_field.py:113-131 with all
  pixels valid; the patch means cancel)."""
  p = a.shape[0]
  out = np.zeros((2 * p - 1, 2 * p - 1))
  ia = np.pad(a.cumsum(0).cumsum(1), ((1, 0), (1, 0)))
  iqa = np.pad((a * a).cumsum(0).cumsum(1), ((1, 0), (1, 0)))
  ib = np.pad(b.cumsum(0).cumsum(1), ((1, 0), (1, 0)))
  iqb = np.pad((b * b).cumsum(0).cumsum(1), ((1, 0), (1, 0)))

  def box(i, ys, xs):
    return i[ys.stop, xs.stop] - i[ys.start, xs.stop] - i[ys.stop, xs.start] + i[ys.start, xs.start]

  for dy in range(-(p - 1), p):
    ya, yb = _rows_of_shift(dy, p)
    for dx in range(-(p - 1), p):
      xa, xb = _rows_of_shift(dx, p)
      n = (ya.stop - ya.start) * (xa.stop - xa.start)
      pd = max(n * box(iqa, ya, xa) - box(ia, ya, xa) ** 2, 0.0)
      cd = max(n * box(iqb, yb, xb) - box(ib, yb, xb) ** 2, 0.0)
      out[dy + p - 1, dx + p - 1] = np.sqrt(pd * cd) / n
  return out


@pytest.mark.parametrize('kind', ['smooth', 'noise', 'mean_border', 'flat_but_block', 'flat'])
def test_denominator_of_a_clean_pair_is_bounded_by_its_axis_values(kind):
  """masked_axis_max_kernel: the batch maximum of the
  denominator of a clean same-size pair is found on the two shift axes plus the
  candidate cross product, because den(dy, dx) <= min(den(dy, 0), den(0, dx))
  (the sum of squared deviations only grows with the rectangle, and the overlap
  rectangles are rows(dy) x columns(dx) on both sides).  Checked here on the
  directly computed surface of denominators, with the kernel's candidate rule
  (axis value within 1e-5 of the axis maximum) and a float32 evaluation."""
  p = 24
  rng = np.random.default_rng(len(kind))
  a, b = _patches('noise' if kind == 'noise' else 'smooth', rng, p)
  if kind == 'mean_border':
    # border rows / columns AT the mean of the rest: dropping them leaves the sum of
    # squared deviations almost unchanged, so several shifts come within 1e-5 of the
    # zero shift's denominator
    for arr in (a, b):
      arr[:] = np.round((arr - arr.mean()) * 3 + 128).clip(0, 255)
      inner = arr[2:-3, 1:]
      arr[:2] = np.round(inner.mean()); arr[-3:] = np.round(inner.mean())
      arr[:, :1] = np.round(inner.mean())
  elif kind == 'flat_but_block':
    for arr in (a, b):
      blk = arr[8:14, 9:15].copy(); arr[:] = 77; arr[8:14, 9:15] = blk
  elif kind == 'flat':
    a[:] = 200; b[:] = 13
  den = _den_all_shifts(a, b)
  c = p - 1
  row_axis, col_axis = den[:, c], den[c, :]
  # the inequality, exactly (float64 on integers below 2^53: a few ulp of slack)
  bound = np.minimum(row_axis[:, None], col_axis[None, :])
  assert (den <= bound * (1 + 1e-12) + 1e-9).all()
  # the kernel's procedure on float32 values: axis maximum, candidates, cross product
  den32 = den.astype(np.float32)
  amax = max(den32[:, c].max(), den32[c, :].max())
  cy = np.nonzero(den32[:, c] * np.float32(1.00001) >= amax)[0]
  cx = np.nonzero(den32[c, :] * np.float32(1.00001) >= amax)[0]
  found = max(amax, den32[np.ix_(cy, cx)].max()) if amax > 0 else np.float32(0)
  assert found == den32.max()
  if kind in ('smooth', 'noise'):
    assert len(cy) == 1 and len(cx) == 1        # the zero shift alone
  if kind == 'mean_border':
    assert len(cy) * len(cx) > 1                # several shifts within 1e-5 of it
  if kind == 'flat':
    assert den.max() == 0.0
