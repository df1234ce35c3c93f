"""CPU oracle for the stitching callers of the hot path.  TEST INFRASTRUCTURE.

A NumPy restatement of the pieces of the reference's `stitch_rigid.py` and
`stitch_elastic.py` that feed or parameterise the two compute cores:

  estimate_offset        <-> stitch_rigid._estimate_offset      (stitch_rigid.py:39-67)
  elastic_tile_mesh      <-> stitch_rigid.elastic_tile_mesh      (:330-388)
  elastic_tile_mesh_3d   <-> stitch_rigid.elastic_tile_mesh_3d   (:391-473)
  optimize_coarse_mesh   <-> stitch_rigid.optimize_coarse_mesh   (:476-523)
  flow_map_strips        <-> the strip cropping of stitch_elastic.compute_flow_map
                             (stitch_elastic.py:198-282)

Never imported by anything under `sofima_amd/`.  Parity pinning: the reference
has NO tests for these functions; the oracle is pinned by
`tests/golden/stitch_cfg1.npz`, produced in the build container by executing
the unmodified reference source over the NumPy stand-in for jax
(tests/golden/make_golden.py: gen_stitch; "reference over a stand-in", not XLA).
"""
from __future__ import annotations

import types

import numpy as np
from scipy import ndimage

from oracle import flow_oracle
from oracle import mesh_oracle

f32 = np.float32


def range_mask(img, range_limit, filter_size=10):
  """Pixels whose neighbourhood has too little dynamic range
  (stitch_rigid.py:48-55); the subtraction is done in the image dtype like the
  reference (uint8 arithmetic cannot wrap here: max >= min)."""
  return (ndimage.maximum_filter(img, filter_size) -
          ndimage.minimum_filter(img, filter_size)) < range_limit


def estimate_offset(a, b, range_limit, filter_size=10, masks=None):
  """Global offset between two overlap strips -> ([x, y], |peak ratio|)."""
  a_mask = range_mask(a, range_limit, filter_size)
  b_mask = range_mask(b, range_limit, filter_size)
  if masks is not None:
    a_mask = a_mask | masks[0]
    b_mask = b_mask | masks[1]
  f = flow_oracle.flow_field(a, b, a.shape, (1, 1), pre_mask=a_mask,
                             post_mask=b_mask, batch_size=1)
  xo, yo, _, pr = f.squeeze()
  return [xo, yo], abs(pr)


def _pair_terms(x, c_arr, comp, axis):
  """nan_to_num((x_c[i+1] - x_c[i]) - c_c[i]) for every pair along `axis`
  (-1: x pairs, -2: y pairs); jnp.nan_to_num defaults (inf -> +-max)."""
  hi = [slice(None)] * (x.ndim - 1)
  lo = [slice(None)] * (x.ndim - 1)
  hi[axis] = slice(1, None)
  lo[axis] = slice(None, -1)
  d = x[comp][tuple(hi)] - x[comp][tuple(lo)]
  return np.nan_to_num(d - c_arr[comp][tuple(lo)]).astype(f32)


def _scatter(f_tot, comp, t, axis):
  hi = [slice(None)] * (f_tot.ndim - 1)
  lo = [slice(None)] * (f_tot.ndim - 1)
  hi[axis] = slice(1, None)
  lo[axis] = slice(None, -1)
  f_tot[comp][tuple(lo)] += t
  f_tot[comp][tuple(hi)] -= t


def _tile_force(x, cx, cy, order):
  x = np.asarray(x, f32)
  cx = np.asarray(cx, f32)
  cy = np.asarray(cy, f32)
  f_tot = np.zeros_like(x)
  for comp, which in order:
    if which == 'x':
      _scatter(f_tot, comp, _pair_terms(x, cx, comp, -1), -1)
    else:
      _scatter(f_tot, comp, _pair_terms(x, cy, comp, -2), -2)
  return f_tot


def elastic_tile_mesh(x, cx, cy, k=None, stride=None, prefer_orig_order=False,
                      links=None):
  """Force on the nodes of a 2-D tile mesh, [2, z, y, x]; the families are
  accumulated in the reference's order (stitch_rigid.py:357-386)."""
  del k, stride, prefer_orig_order, links
  return _tile_force(x, cx, cy, ((0, 'x'), (1, 'y'), (0, 'y'), (1, 'x')))


def elastic_tile_mesh_3d(x, cx, cy, k=None, stride=None, prefer_orig_order=False,
                         links=None):
  """Force on the nodes of a 3-D tile mesh, [3, z, y, x] (stitch_rigid.py:419-473)."""
  del k, stride, prefer_orig_order, links
  return _tile_force(x, cx, cy, ((0, 'x'), (1, 'y'), (0, 'y'), (1, 'x'),
                                 (2, 'x'), (2, 'y')))


def default_coarse_config():
  """The fallback IntegrationConfig of optimize_coarse_mesh (stitch_rigid.py:496-507)."""
  return types.SimpleNamespace(
      dt=0.001, gamma=0.0, k0=0.0, k=0.1, stride=(1, 1), num_iters=1000,
      max_iters=100000, stop_v_max=0.001, fire=True, f_alpha=0.99, f_inc=1.1,
      f_dec=0.5, alpha=0.1, n_min=5, dt_max=100, start_cap=1e6, final_cap=1e6,
      cap_scale=1.1, cap_upscale_every=100, prefer_orig_order=False,
      remove_drift=False)


def optimize_coarse_mesh(cx, cy, cfg=None, mesh_fn=elastic_tile_mesh):
  """Relaxed tile positions, same shape as cx / cy (stitch_rigid.py:476-523)."""
  if cfg is None:
    cfg = default_coarse_config()

  def force(x, *args, **kwargs):
    return mesh_fn(x, cx, cy, *args, **kwargs)

  res = mesh_oracle.relax_mesh(np.zeros_like(cx), None, cfg, mesh_force=force)
  return np.array(res[0])


def flow_map_strips(pre, post, offset, axis, stride):
  """The overlap strips compute_flow_map correlates for one tile pair and the
  offset it records (stitch_elastic.py:232-262, :277-280).

  offset: coarse (x, y) offset of the pair; axis 0: x neighbours, 1: y
  neighbours; stride is yx.  Returns (pre strip, post strip, (off_x, off_y)).
  """
  stride = np.asarray(stride)
  offset = np.asarray(offset, dtype=np.float64)
  rounded = stride[::-1] * np.round(offset / stride[::-1])
  overlap = -int(offset[axis])
  extent = pre.shape[1 - axis]
  overlap = extent - (extent - overlap) // stride[1 - axis] * stride[1 - axis]
  ortho = int(rounded[1 - axis])
  pre_sel = [slice(None), slice(None)]
  post_sel = [slice(None), slice(None)]
  pre_sel[1 - axis] = slice(-overlap, None)
  post_sel[1 - axis] = slice(None, overlap)
  if ortho > 0:
    pre_sel[axis] = slice(ortho, None)
    post_sel[axis] = slice(None, -ortho)
  elif ortho < 0:
    pre_sel[axis] = slice(None, ortho)
    post_sel[axis] = slice(-ortho, None)
  off = (-overlap, ortho) if axis == 0 else (ortho, -overlap)
  return pre[tuple(pre_sel)], post[tuple(post_sel)], off


def compute_flow_map(tile_map, offset_map, axis, patch_size=(120, 120),
                     stride=(20, 20), batch_size=256):
  """Fine flow of every adjacent tile pair (stitch_elastic.py:198-282)."""
  ret, offsets = {}, {}
  pad_y = patch_size[0] // 2 // stride[0]
  pad_x = patch_size[1] // 2 // stride[1]
  ny, nx = offset_map.shape[-2:]
  for y in range(0, ny - axis):
    for x in range(0, nx - (1 - axis)):
      if np.isnan(offset_map[0, y, x]):
        continue
      pre, post, off = flow_map_strips(
          tile_map[x, y], tile_map[x + (1 - axis), y + axis],
          offset_map[:, y, x], axis, stride)
      f = flow_oracle.flow_field(pre, post, patch_size, stride,
                                 batch_size=batch_size)
      ret[x, y] = np.pad(f, [[0, 0], [pad_y, pad_y - 1], [pad_x, pad_x - 1]],
                         constant_values=np.nan)
      offsets[x, y] = off
  return ret, offsets
