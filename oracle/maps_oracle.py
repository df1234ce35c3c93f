"""CPU oracle for coordinate-map composition and the montage target mesh.
TEST INFRASTRUCTURE.

NumPy restatement of `map_utils.compose_maps_fast` (map_utils.py:616-734, with
the order-1 `jax.scipy.ndimage.map_coordinates` it calls) and of
`stitch_elastic.compute_target_mesh` (stitch_elastic.py:456-676) of the
reference.  Pinned by tests/golden/compose_maps.npz and montage.npz, which were
produced by executing the unmodified reference source over the NumPy stand-in
for jax (tests/golden/_refshim) -- "reference over a stand-in", not XLA -- and by
the reference's known-answer test tests/map_utils_test.py:266-301.
"""
from __future__ import annotations

import itertools

import numpy as np

f32 = np.float32


def map_coordinates_linear(inp, coords, mode, cval=np.nan):
  """order-1 interpolation with JAX's per-corner out-of-range handling."""
  inp = np.asarray(inp, f32)
  per_axis = []
  for c, size in zip(coords, inp.shape):
    c = np.asarray(c, f32)
    lo = np.floor(c)
    w_hi = (c - lo).astype(f32)
    with np.errstate(invalid='ignore'):
      lo_i = np.nan_to_num(lo, nan=-1e9).astype(np.int64)
    per_axis.append(((lo_i, f32(1) - w_hi), (lo_i + 1, w_hi), size))
  out = None
  for combo in itertools.product((0, 1), repeat=len(per_axis)):
    idx, w, valid = [], None, True
    for ax, pick in zip(per_axis, combo):
      i, wi = ax[pick]
      size = ax[2]
      valid = valid & (i >= 0) & (i < size)
      idx.append(np.clip(i, 0, size - 1))
      w = wi if w is None else (w * wi).astype(f32)
    val = inp[tuple(idx)]
    if mode == 'constant':
      val = np.where(valid, val, f32(cval))
    term = (w * val).astype(f32)
    out = term if out is None else (out + term).astype(f32)
  nan_q = np.zeros(out.shape, bool)
  for c in coords:
    nan_q |= np.isnan(c)
  return np.where(nan_q, f32(np.nan), out).astype(f32)


def compose_maps_fast(map1, start1, stride1, map2, start2, stride2,
                      mode='nearest'):
  map1 = np.asarray(map1, f32)
  map2 = np.asarray(map2, f32)
  dim = map1.shape[0]
  vec = lambda v: tuple(np.ravel(v)[-dim:]) if np.ndim(v) else (v,) * dim
  stride1, stride2 = vec(stride1), vec(stride2)
  s1 = np.asarray(start1, np.float64).ravel()[-dim:]
  s2 = np.asarray(start2, np.float64).ravel()[-dim:]
  origin = np.minimum(s1, s2)

  def ref(shape, start, stride):
    axes = [(np.arange(n) + (start[i] - origin[i])) * stride[i]
            for i, n in enumerate(shape)]
    return [g.astype(f32) for g in np.meshgrid(*axes, indexing='ij')]

  ref1 = ref(map1.shape[-dim:], s1, stride1)
  ref2 = ref(map2.shape[-dim:], s2, stride2)
  out = np.zeros_like(map1)
  if dim == 2:
    for z in range(map1.shape[1]):
      qx = (ref1[1] + map1[0, z]) / f32(stride2[1])
      qy = (ref1[0] + map1[1, z]) / f32(stride2[0])
      out[0, z] = map_coordinates_linear(map2[0, z] + ref2[1], [qy, qx],
                                         mode) - ref1[1]
      out[1, z] = map_coordinates_linear(map2[1, z] + ref2[0], [qy, qx],
                                         mode) - ref1[0]
    return out
  qx = (ref1[2] + map1[0]) / f32(stride2[2])
  qy = (ref1[1] + map1[1]) / f32(stride2[1])
  qz = (ref1[0] + map1[2]) / f32(stride2[0])
  q = [qz, qy, qx]
  out[0] = map_coordinates_linear(map2[0] + ref2[2], q, mode) - ref1[2]
  out[1] = map_coordinates_linear(map2[1] + ref2[1], q, mode) - ref1[1]
  out[2] = map_coordinates_linear(map2[2] + ref2[0], q, mode) - ref1[0]
  return out


def compute_target_mesh(nbor_data, x, fx, fy, stride):
  """[2, y, x] / [3, z, y, x] target positions of one tile
  (stitch_elastic.py:624-676 with _update_mesh :573-620, _apply_flow :456-570)."""
  x = np.asarray(x, f32)
  ncomp = x.shape[0]
  msz = x.shape[-ncomp:]                      # [z]yx size of one tile mesh
  my, mx = msz[-2:]
  ext = [msz[i] + max(fy.shape[-ncomp + i], fx.shape[-ncomp + i]) for i in range(ncomp)]
  canvas = np.full([ncomp] + ext, np.nan, f32)
  for nd in np.asarray(nbor_data):
    nbor, flow_idx, off_ortho, f_ortho, f_overlap, fine_x, fine_y, dim = (
        int(v) for v in nd[:8])
    if nbor == -1:
      continue
    mult = 1 if nbor == flow_idx else -1
    flow = fx if dim == 0 else fy
    par_n = mx if dim == 0 else my
    ortho_n = my if dim == 0 else mx
    start_par = par_n - f_overlap if mult == 1 else 0
    s_hi = (mult == 1 and off_ortho > 0) or (mult == -1 and off_ortho < 0)
    start_ortho = ortho_n - f_ortho if s_hi else 0
    start = (start_ortho, start_par) if dim == 0 else (start_par, start_ortho)
    tg_par = 0 if mult == 1 else par_n - f_overlap
    t_hi = (mult == 1 and off_ortho < 0) or (mult == -1 and off_ortho > 0)
    tg_ortho = ortho_n - f_ortho if t_hi else 0
    tg = (tg_ortho, tg_par) if dim == 0 else (tg_par, tg_ortho)
    nflow = mult * np.asarray(flow[:, flow_idx], f32)
    if ncomp == 3:
      # stitch_elastic.py:509-518, 556-561
      off_z, f_z, fine_z = int(nd[8]), int(nd[9]), int(nd[10])
      sz_hi = (mult == 1 and off_z > 0) or (mult == -1 and off_z < 0)
      start = (msz[0] - f_z if sz_hi else 0,) + start
      tz_hi = (mult == 1 and off_z < 0) or (mult == -1 and off_z > 0)
      tg = (msz[0] - f_z if tz_hi else 0,) + tg
      upd = compose_maps_fast(nflow, start, stride, x[:, nbor], (0, 0, 0), stride,
                              mode='constant')
      upd = upd + f32(mult) * np.array([fine_x, fine_y, fine_z], f32).reshape(3, 1, 1, 1)
    else:
      upd = compose_maps_fast(nflow[:, None], start, stride, x[:, nbor][:, None],
                              (0, 0), stride, mode='constant')[:, 0]
      upd = upd + f32(mult) * np.array([fine_x, fine_y], f32).reshape(2, 1, 1)
    sl = (slice(None),) + tuple(slice(t, t + n) for t, n in zip(tg, upd.shape[1:]))
    canvas[sl] = np.where(np.isnan(upd), canvas[sl], upd)
  return canvas[(slice(None),) + tuple(slice(0, n) for n in msz)]


def target_mesh_all(nbors, x, fx, fy, stride):
  """prev_fn of the montage relaxation: [2, N, y, x] or [3, N, z, y, x]."""
  return np.stack([compute_target_mesh(nb, x, fx, fy, stride) for nb in nbors],
                  axis=1)


def mask_irregular(coord_map, stride, frac, max_frac=None, dilation_iters=1):
  """map_utils.py:737-786: returns (masked copy of the map, bad mask).

  The reference masks in place; the oracle returns the masked copy instead.
  """
  from scipy import ndimage
  coord_map = np.array(coord_map, copy=True)
  assert coord_map.ndim == 3 and coord_map.shape[0] == 2
  if max_frac is None:
    max_frac = 2 - frac
  stride_x, stride_y = np.asarray(stride)
  # map_utils.py:768-771
  diff_x = np.pad(np.diff(coord_map[0], axis=-1), [[0, 0], [0, 1]]) + stride_x
  diff_y = np.pad(np.diff(coord_map[1], axis=-2), [[0, 1], [0, 0]]) + stride_y
  with np.errstate(invalid='ignore'):
    bad = (diff_x < frac * stride_x) | (diff_y < frac * stride_y)
    bad |= (diff_x > max_frac * stride_x) | (diff_y > max_frac * stride_y)
  if dilation_iters > 0:  # map_utils.py:776-781
    bad = ndimage.binary_dilation(bad, ndimage.generate_binary_structure(2, 2),
                                  iterations=dilation_iters)
  coord_map[:, bad] = np.nan
  return coord_map, bad
