"""CPU oracle for the patch cross-correlation flow estimator.  TEST INFRASTRUCTURE.

A NumPy/SciPy restatement of the algorithm of the reference's `flow_field.py`
(/root/reference/flow_field.py), written from its observable behaviour.  It is
the checker for the HIP path and the `cpu_baseline` leg of bench.py; it is
never imported by anything under `sofima_amd/`.

Parity pinning (see tests/test_oracle_golden.py, tests/golden/make_golden.py):
  * the reference's own known-answer tests (tests/flow_field_test.py:24-125),
    re-typed in tests/test_reference_kats.py;
  * golden vectors produced in the build container from the reference's
    genuine pure-NumPy path `masked_xcorr(use_jax=False)` (flow_field.py:65,
    106-111, 142-155) -- needs only import stubs;
  * golden vectors produced by executing the unmodified reference source over
    a NumPy stand-in for jax (tests/golden/_refshim) -- "reference over a
    stand-in", NOT XLA numbers.

Two evaluators of the correlation surface are provided:
  `xcorr_surface`         FFT form, float32/complex64 like the reference
                          (flow_field.py:36-156);
  `xcorr_surface_direct`  shift-by-shift summation in float64 / exact integers
                          -- the formulation the HIP kernels use; the two are
                          checked against each other in the CPU tests.
"""
from __future__ import annotations

import collections.abc
import itertools

import numpy as np
import scipy.fft
import scipy.signal

F32_EPS = np.finfo(np.float32).eps


def _seq(v, n):
  if isinstance(v, collections.abc.Sequence) or isinstance(v, np.ndarray):
    return tuple(int(a) for a in v)
  return (int(v),) * n


# ---------------------------------------------------------------------------
# correlation surface   (flow_field.py:36-156)
# ---------------------------------------------------------------------------
def xcorr_surface(prev, curr, prev_mask=None, curr_mask=None, dim=2,
                  workers=1, dtype=np.float32):
  """Full linear cross-correlation over the last `dim` axes, FFT form.

  Unmasked: RAW correlation (flow_field.py:88-89).  Masked: Padfield's
  normalised masked correlation with the reference's round / eps clamp /
  tolerance / clip / 0.3-overlap steps (flow_field.py:91-155); `tol` and the
  overlap threshold use maxima over the WHOLE input array (batch included).
  Layout: out[..., k] = sum_i prev[i + k - (Q-1)] * curr[i]; zero shift at
  index Q-1.

  dtype=float32 mirrors the jax path used by flow_field(); dtype=float64
  mirrors the reference's NumPy path (use_jax=False), where boolean masks
  promote the arithmetic to double (this only matters at exact ties of the
  0.3*max overlap threshold).
  """
  ft = np.dtype(dtype).type
  prev = np.asarray(prev, dtype=ft)
  curr = np.asarray(curr, dtype=ft)
  p_shape = prev.shape[-dim:]
  q_shape = curr.shape[-dim:]
  full = tuple(p + q - 1 for p, q in zip(p_shape, q_shape))
  fast = tuple(scipy.fft.next_fast_len(n, real=False) for n in full)
  axes = tuple(range(-dim, 0))
  crop = (Ellipsis,) + tuple(slice(0, n) for n in full)
  flip = (Ellipsis,) + (slice(None, None, -1),) * dim

  def fwd(a):
    return scipy.fft.rfftn(a.astype(ft), s=fast, axes=axes, workers=workers)

  def inv(a):
    return scipy.fft.irfftn(a, s=fast, axes=axes, workers=workers)

  have_mask = prev_mask is not None or curr_mask is not None
  if prev_mask is not None:
    prev = np.where(prev_mask, ft(0), prev)
  if curr_mask is not None:
    curr = np.where(curr_mask, ft(0), curr)
  curr = curr[flip]

  pf = fwd(prev)
  cf = fwd(curr)
  num = inv(pf * cf)
  if not have_mask:
    return np.ascontiguousarray(num[crop])

  p_valid = (np.ones(prev.shape, bool) if prev_mask is None
             else np.logical_not(prev_mask))
  c_valid = (np.ones(curr.shape, bool) if curr_mask is None
             else np.logical_not(curr_mask)[flip])
  pvf = fwd(p_valid)
  cvf = fwd(c_valid)

  n_ov = np.round(inv(cvf * pvf))
  n_ov = np.fmax(n_ov, ft(F32_EPS))
  inv_ov = ft(1.0) / n_ov

  sum_p = inv(cvf * pf)   # sum of prev over the overlap (curr-valid weighted)
  sum_c = inv(pvf * cf)
  num = num - sum_p * sum_c * inv_ov

  var_p = inv(cvf * fwd(np.square(prev))) - np.square(sum_p) * inv_ov
  var_p = np.fmax(var_p, ft(0))
  var_c = inv(pvf * fwd(np.square(curr))) - np.square(sum_c) * inv_ov
  var_c = np.fmax(var_c, ft(0))
  den = np.sqrt(var_p * var_c)

  num = num[crop]
  den = den[crop]
  n_ov = n_ov[crop]

  tol = ft(1e3) * ft(F32_EPS) * np.max(np.abs(den))
  ok = den > tol
  out = np.zeros_like(den)
  out[ok] = num[ok] / den[ok]
  np.clip(out, -1, 1, out=out)
  out[n_ov < ft(0.3) * np.max(n_ov)] = 0
  return out.astype(np.float32)


def _direct_corr(a, b, dim):
  """sum_i a[i + k - (Q-1)] * b[i] in float64 (exact for small integers)."""
  a = np.asarray(a, np.float64)
  b = np.asarray(b, np.float64)
  lead = a.shape[:-dim]
  out_sp = tuple(p + q - 1 for p, q in zip(a.shape[-dim:], b.shape[-dim:]))
  out = np.empty(lead + out_sp, np.float64)
  for idx in itertools.product(*[range(n) for n in lead]):
    out[idx] = scipy.signal.correlate(a[idx], b[idx], mode='full',
                                      method='direct')
  return out


def xcorr_surface_direct(prev, curr, prev_mask=None, curr_mask=None, dim=2):
  """Same quantity as `xcorr_surface`, evaluated shift by shift in float64.

  This is the formulation of the HIP kernels (DESIGN.md "xcorr"): the masked
  surface is assembled from six plain correlations.
  """
  prev = np.asarray(prev, np.float64)
  curr = np.asarray(curr, np.float64)
  if prev_mask is None and curr_mask is None:
    return _direct_corr(prev, curr, dim).astype(np.float32)
  va = (np.ones(prev.shape) if prev_mask is None
        else 1.0 - np.asarray(prev_mask, np.float64))
  vb = (np.ones(curr.shape) if curr_mask is None
        else 1.0 - np.asarray(curr_mask, np.float64))
  a0 = prev * va
  b0 = curr * vb
  n_ov = np.maximum(np.round(_direct_corr(va, vb, dim)), F32_EPS)
  s_a = _direct_corr(a0, vb, dim)
  s_b = _direct_corr(va, b0, dim)
  num = _direct_corr(a0, b0, dim) - s_a * s_b / n_ov
  var_a = np.maximum(_direct_corr(a0 * a0, vb, dim) - s_a * s_a / n_ov, 0)
  var_b = np.maximum(_direct_corr(va, b0 * b0, dim) - s_b * s_b / n_ov, 0)
  den = np.sqrt(var_a * var_b)
  tol = 1e3 * F32_EPS * np.max(np.abs(den))
  out = np.where(den > tol, num / np.where(den > tol, den, 1.0), 0.0)
  out = np.clip(out, -1, 1)
  out[n_ov < 0.3 * np.max(n_ov)] = 0
  return out.astype(np.float32)


# ---------------------------------------------------------------------------
# patch selection by mask   (flow_field.py:159-175, 575-589)
# ---------------------------------------------------------------------------
def integral_image(mask):
  """Summed-volume table with a leading zero row/column per axis."""
  ii = np.asarray(mask).astype(np.int64)
  for ax in range(ii.ndim):
    ii = np.cumsum(ii, axis=ax)
  return np.pad(ii, [(1, 0)] * ii.ndim)


def query_integral_image(svt, diam, stride):
  """Box sums of size `diam` sampled every `stride` (inclusion-exclusion)."""
  nd = svt.ndim
  hi = [np.s_[diam[i]::stride[i]] for i in range(nd)]
  lo = [np.s_[:-diam[i]:stride[i]] for i in range(nd)]
  out = 0
  for bits in itertools.product((0, 1), repeat=nd):
    sel = tuple(hi[i] if b else lo[i] for i, b in enumerate(bits))
    out = out + (-1) ** (nd - sum(bits)) * svt[sel]
  return out


# ---------------------------------------------------------------------------
# peaks   (flow_field.py:178-275)
# ---------------------------------------------------------------------------
def _maxfilter_zero_same(img, size):
  """Separable running max, window `size` per axis, ZERO 'same' padding."""
  out = img
  for ax in range(1, img.ndim):
    s = size[ax - 1]
    lo, hi = (s - 1) // 2, s // 2
    pad = [(0, 0)] * img.ndim
    pad[ax] = (lo, hi)
    padded = np.pad(out, pad)
    win = np.lib.stride_tricks.sliding_window_view(padded, s, axis=ax)
    out = win.max(axis=-1)
  return out


def batched_peaks(img, center_offset, min_distance, threshold_rel,
                  peak_radius=5):
  """Top-two peak statistics for a batch of surfaces -> [b, dim+2] float32.

  Reproduces the reference's batch-coupled behaviours: the first-peak flat
  indices of ALL surfaces in the batch are suppressed in EVERY surface before
  the second search (flow_field.py:263-265), while the second value is read
  from the un-suppressed array (:266-268).
  """
  img = np.asarray(img, np.float32)
  b = img.shape[0]
  dim = img.ndim - 1
  sp = img.shape[1:]
  size = (2 * int(min_distance) + 1,) * dim
  radius = np.array(_seq(peak_radius, dim))
  win = 2 * radius + 1

  mx = _maxfilter_zero_same(img, size)
  thr = np.float32(threshold_rel) * img.reshape(b, -1).max(axis=1)
  thr = thr.reshape((b,) + (1,) * dim)
  is_peak = (img == mx) & (img > thr)
  cand = np.where(is_peak, img, -np.inf).reshape(b, -1)

  i1 = np.argmax(cand, axis=1)
  v1 = cand[np.arange(b), i1]
  sup = cand.copy()
  sup[:, i1] = -np.inf
  i2 = np.argmax(sup, axis=1)
  v2 = cand[np.arange(b), i2]

  out = np.full((b, dim + 2), np.nan, np.float32)
  for n in range(b):
    if np.isinf(v1[n]):
      continue
    pos = np.array(np.unravel_index(i1[n], sp))
    centred = pos.astype(np.float32) - np.asarray(center_offset, np.float32)
    start = np.clip(pos - win // 2, 0, np.array(sp) - win)
    sl = tuple(slice(int(s), int(s + w)) for s, w in zip(start, win))
    with np.errstate(divide='ignore', invalid='ignore'):
      sharp = img[n][tuple(pos)] / np.min(img[n][sl])
      ratio = np.float32(0) if np.isinf(v2[n]) else v1[n] / v2[n]
    out[n, :dim] = centred[::-1]
    out[n, dim] = sharp
    out[n, dim + 1] = ratio
  return out


# ---------------------------------------------------------------------------
# batch = gather + mean + surface + peaks   (flow_field.py:278-441)
# ---------------------------------------------------------------------------
def _gather(arr, starts, size):
  """Patch gather with the start clamped so the patch stays inside `arr`."""
  out = np.empty((len(starts),) + tuple(size), arr.dtype)
  lim = np.array(arr.shape) - np.array(size)
  for n, st in enumerate(starts):
    st = np.clip(st, 0, lim)
    sl = tuple(slice(int(s), int(s + w)) for s, w in zip(st, size))
    out[n] = arr[sl]
  return out


def batched_xcorr(pre_image, post_image, pre_mask, post_mask, patch_size,
                  starts, mean, post_patch_size=None, post_starts=None,
                  workers=1):
  dim = len(patch_size)
  if post_patch_size is None:
    post_patch_size = patch_size
  if post_starts is None:
    post_starts = starts
  a = _gather(np.asarray(pre_image), starts, patch_size)
  b = _gather(np.asarray(post_image), post_starts, post_patch_size)
  am = None if pre_mask is None else _gather(
      np.asarray(pre_mask, bool), starts, patch_size)
  bm = None if post_mask is None else _gather(
      np.asarray(post_mask, bool), post_starts, post_patch_size)
  axes = tuple(range(-dim, 0))

  def centre(src, msk):
    src = src.astype(np.float32)
    if mean is not None:
      return src - np.float32(mean)
    if msk is None:
      mu = src.mean(axis=axes, keepdims=True, dtype=np.float32)
    else:
      with np.errstate(invalid='ignore', divide='ignore'):
        cnt = (~msk).sum(axis=axes, keepdims=True).astype(np.float32)
        mu = np.where(msk, np.float32(0), src).sum(
            axis=axes, keepdims=True, dtype=np.float32) / cnt
    return src - mu

  center_offset = (np.array(patch_size) + np.array(post_patch_size)) // 2 - 1
  surf = xcorr_surface(centre(a, am), centre(b, bm), am, bm, dim=dim,
                       workers=workers)
  return center_offset, surf


def batched_xcorr_peaks(pre_image, post_image, pre_mask, post_mask,
                        patch_size, starts, mean, min_distance=2,
                        threshold_rel=0.5, peak_radius=5,
                        post_patch_size=None, post_starts=None, workers=1):
  off, surf = batched_xcorr(pre_image, post_image, pre_mask, post_mask,
                            patch_size, starts, mean, post_patch_size,
                            post_starts, workers=workers)
  return batched_peaks(surf, off, min_distance, threshold_rel, peak_radius)


# ---------------------------------------------------------------------------
# host driver   (flow_field.py:449-712)
# ---------------------------------------------------------------------------
def plan_patches(pre_shape, post_shape, patch_size, step, post_patch_size,
                 pre_mask, post_mask, selection_mask, max_masked):
  """Output grid + row-major list of grid positions to evaluate."""
  step = np.array(step)
  out_shape = (np.array(post_shape) - (np.array(post_patch_size) - step)) // step
  sel_idx = tuple(slice(0, int(s)) for s in out_shape)
  if selection_mask is None:
    sel = np.ones(out_shape, bool)
  else:
    sel = np.array(selection_mask[sel_idx], dtype=bool)
  for msk, psz in ((pre_mask, patch_size), (post_mask, post_patch_size)):
    if msk is None:
      continue
    cnt = query_integral_image(integral_image(msk), psz, step)
    drop = (cnt / np.prod(psz) >= max_masked)[sel_idx]
    sel[drop] = False
  return out_shape, np.array(np.nonzero(sel)).T


def _target_offsets(field, tstep, starts, psize, img_shape):
  """Integer [z]yx shifts looked up in a targeting field, clipped in-bounds."""
  centre = np.array(psize) // 2
  q = np.round((starts + centre) / np.array(tstep)).astype(int)
  idx = tuple(np.clip(q[:, i], 0, field.shape[i + 1] - 1)
              for i in range(q.shape[1]))
  off = np.nan_to_num(field[(slice(None),) + idx].T).astype(int)[:, ::-1]
  new = starts + off
  off = off - np.minimum(new, 0)
  over = np.maximum(new + np.array(psize), np.array(img_shape)) - np.array(img_shape)
  return off - over


def flow_field(pre_image, post_image, patch_size, step, pre_mask=None,
               post_mask=None, mask_only_for_patch_selection=False,
               selection_mask=None, max_masked=0.75, batch_size=4096,
               post_patch_size=None, pre_targeting_field=None,
               pre_targeting_step=None, post_targeting_field=None,
               post_targeting_step=None, mean=None, min_distance=2,
               peak_radius=5, workers=1, max_batches=None, only_batches=None):
  """[dim+2, *grid] float32 flow (x, y[, z], sharpness, ratio); NaN = none.

  max_batches / only_batches (test infrastructure, not in the reference): stop
  after the first n batches / evaluate only the batches with these indices;
  the rest of the field stays NaN."""
  nd = pre_image.ndim
  patch_size = _seq(patch_size, nd)
  post_patch_size = patch_size if post_patch_size is None else _seq(
      post_patch_size, nd)
  step = _seq(step, nd)
  if pre_targeting_step is not None:
    pre_targeting_step = _seq(pre_targeting_step, nd)
  if post_targeting_step is not None:
    post_targeting_step = _seq(post_targeting_step, nd)

  out_shape, grid = plan_patches(pre_image.shape, post_image.shape, patch_size,
                                 step, post_patch_size, pre_mask, post_mask,
                                 selection_mask, max_masked)
  out = np.full([nd + 2] + out_shape.tolist(), np.nan, np.float32)
  if mask_only_for_patch_selection:
    pre_mask = post_mask = None
  shrink = (np.array(patch_size) - np.array(post_patch_size)) // 2

  n_done = 0
  for lo in range(0, len(grid), batch_size):
    if max_batches is not None and n_done >= max_batches:
      break
    if only_batches is not None and lo // batch_size not in only_batches:
      continue
    n_done += 1
    pos = grid[lo:lo + batch_size]
    real = len(pos)
    if real < batch_size:
      pos_p = np.concatenate([pos, np.repeat(pos[-1:], batch_size - real, 0)])
    else:
      pos_p = pos
    post_st = pos_p * np.array(step)
    pre_st = np.maximum(post_st - shrink, 0)
    pre_off = post_off = None
    if pre_targeting_field is not None and pre_targeting_step is not None:
      pre_off = _target_offsets(pre_targeting_field, pre_targeting_step,
                                pre_st, patch_size, pre_image.shape)
      pre_st = pre_st + pre_off
    if post_targeting_field is not None and post_targeting_step is not None:
      post_off = _target_offsets(post_targeting_field, post_targeting_step,
                                 post_st, post_patch_size, post_image.shape)
      post_st = post_st + post_off
    pre_st = np.maximum(pre_st, 0)
    post_st = np.maximum(post_st, 0)
    pk = batched_xcorr_peaks(pre_image, post_image, pre_mask, post_mask,
                             patch_size, pre_st, mean, min_distance, 0.5,
                             peak_radius, post_patch_size, post_st,
                             workers=workers)[:real].copy()
    if pre_off is not None:
      pk[:, :nd] += pre_off[:real, ::-1]
    if post_off is not None:
      pk[:, :nd] -= post_off[:real, ::-1]
    out[(slice(None),) + tuple(pos.T)] = pk.T
  return out
