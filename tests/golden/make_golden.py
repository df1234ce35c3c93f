"""Generates tests/golden/*.npz by running the UNMODIFIED reference source.

Build-container only: needs /root/reference.  Two kinds of vectors:

  * `xcorr_np_*`: the reference's genuine pure-NumPy path
    `flow_field.masked_xcorr(use_jax=False)` -- needs only import stubs;
  * everything else: reference functions executed over the NumPy stand-in for
    jax in `_refshim/refshim.py` ("reference over a stand-in", NOT XLA).

Every file stores inputs AND expected outputs, so the tests never need the
reference.  Run:  python tests/golden/make_golden.py
"""
import dataclasses
import json
import os
import sys

import numpy as np
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '_refshim'))
import refshim  # noqa: E402

refshim.install()
from sofima import flow_field as rff  # noqa: E402
from sofima import mesh as rmesh  # noqa: E402
from sofima import map_utils as rmap  # noqa: E402


def save(name, **arrs):
  path = os.path.join(HERE, name + '.npz')
  np.savez_compressed(path, **arrs)
  print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


def em_like(rng, shape, sigma=2.0):
  img = ndimage.gaussian_filter(rng.standard_normal(shape), sigma)
  img = (img - img.min()) / (img.max() - img.min()) * 255
  return img.astype(np.uint8)


# ---------------------------------------------------------------------------
def gen_xcorr_np():
  rng = np.random.default_rng(11)
  a = rng.standard_normal((3, 24, 24)).astype(np.float32) * 30
  b = rng.standard_normal((3, 20, 20)).astype(np.float32) * 30
  am = rng.random(a.shape) < 0.15
  bm = rng.random(b.shape) < 0.15
  a3 = rng.standard_normal((2, 8, 10, 12)).astype(np.float32) * 10
  b3 = rng.standard_normal((2, 6, 10, 12)).astype(np.float32) * 10
  am3 = rng.random(a3.shape) < 0.1
  bm3 = rng.random(b3.shape) < 0.1
  save(
      'xcorr_np',
      a=a, b=b, am=am, bm=bm, a3=a3, b3=b3, am3=am3, bm3=bm3,
      unmasked=rff.masked_xcorr(a, b, use_jax=False),
      masked=rff.masked_xcorr(a.copy(), b.copy(), am, bm, use_jax=False),
      masked_prev_only=rff.masked_xcorr(a.copy(), b.copy(), am, None,
                                        use_jax=False),
      unmasked3=rff.masked_xcorr(a3, b3, use_jax=False, dim=3),
      masked3=rff.masked_xcorr(a3.copy(), b3.copy(), am3, bm3, use_jax=False,
                               dim=3),
  )


def gen_peaks():
  rng = np.random.default_rng(12)
  s = 31
  imgs = np.zeros((8, s, s), np.float32)
  # 0: two peaks 10 @ (10,10), 8 @ (20,20)
  imgs[0, 10, 10] = 10
  imgs[0, 20, 20] = 8
  # 1: single peak exactly where image 0 has its 2nd peak (batch coupling)
  imgs[1, 20, 20] = 5
  # 2: single corner peak on a floor (window shift + zero padding)
  imgs[2] = 0.1
  imgs[2, 0, 0] = 3.1
  # 3: all zero -> NaNs
  # 4: negative-only surface
  imgs[4] = -1.0
  imgs[4, 15, 15] = -0.5
  # 5..7: smooth random surfaces
  for i in (5, 6, 7):
    imgs[i] = ndimage.gaussian_filter(
        rng.standard_normal((s, s)), 1.5).astype(np.float32)
  out_all = np.array(rff._batched_peaks(imgs, (15, 15), 2, 0.5, 5))
  out_0 = np.array(rff._batched_peaks(imgs[:1], (15, 15), 2, 0.5, 5))
  out_r = np.array(rff._batched_peaks(imgs[5:], (15, 15), 2, 0.5, (2, 3)))
  vol = ndimage.gaussian_filter(
      rng.standard_normal((3, 9, 11, 13)), 1.0).astype(np.float32)
  out_3d = np.array(rff._batched_peaks(vol, (4, 5, 6), 1, 0.5, (1, 2, 2)))
  save('peaks', imgs=imgs, out_all=out_all, out_0=out_0, out_r=out_r,
       vol=vol, out_3d=out_3d)


def gen_flow():
  rng = np.random.default_rng(13)
  h, w, m = 192, 160, 12
  base = em_like(rng, (h + 2 * m, w + 2 * m))
  dy, dx = 3, -5
  pre = base[m:m + h, m:m + w].copy()
  post = base[m + dy:m + dy + h, m + dx:m + dx + w].copy()
  calc = rff.JAXMaskedXCorrWithStatsCalculator()
  out = {'pre': pre, 'post': post}
  out['plain'] = calc.flow_field(pre, post, 48, 24, batch_size=8)

  pre_mask = np.zeros((h, w), bool)
  pre_mask[:70, :60] = True
  pre_mask[100:130, 90:150] = rng.random((30, 60)) < 0.3
  post_mask = np.zeros((h + 8, w + 8), bool)   # larger than the image
  post_mask[150:, 100:] = True
  post_mask[20:60, 80:120] = rng.random((40, 40)) < 0.2
  out['pre_mask'] = pre_mask
  out['post_mask'] = post_mask
  out['masked'] = calc.flow_field(pre, post, 48, 24, pre_mask=pre_mask,
                                  post_mask=post_mask, batch_size=8)
  out['masksel'] = calc.flow_field(
      pre, post, 48, 24, pre_mask=pre_mask, post_mask=post_mask,
      mask_only_for_patch_selection=True, max_masked=0.5, batch_size=16)

  out['postpatch'] = calc.flow_field(pre, post, 48, 24, batch_size=8,
                                     post_patch_size=32)

  sel = rng.random((7, 5)) < 0.6
  out['sel'] = sel
  out['selected'] = calc.flow_field(pre, post, (48, 32), (24, 16),
                                    selection_mask=np.pad(sel, ((0, 0), (0, 4))),
                                    batch_size=5)

  # Targeting: a coarse field shifts where patches are taken from.
  tg_pre = np.zeros((2, 4, 4), np.float32)
  tg_pre[0] = 6.0
  tg_pre[1] = -4.0
  tg_pre[0, 0, 0] = np.nan
  tg_post = np.full((2, 3, 3), 0.0, np.float32)
  tg_post[0] = -3.0
  tg_post[1] = 5.0
  out['tg_pre'] = tg_pre
  out['tg_post'] = tg_post
  out['targeted'] = calc.flow_field(
      pre, post, 48, 24, batch_size=8, pre_targeting_field=tg_pre,
      pre_targeting_step=48, post_targeting_field=tg_post,
      post_targeting_step=64)

  calc_m = rff.JAXMaskedXCorrWithStatsCalculator(mean=120.0,
                                                 peak_min_distance=3,
                                                 peak_radius=(3, 4))
  pre_f = pre.astype(np.float32) + rng.standard_normal(pre.shape).astype(
      np.float32)
  post_f = post.astype(np.float32)
  out['pre_f'] = pre_f
  out['post_f'] = post_f
  out['float_mean'] = calc_m.flow_field(pre_f, post_f, 40, 20, batch_size=64)
  save('flow2d', **out)

  # 3-D
  vol = em_like(rng, (24 + 6, 40 + 6, 40 + 6), sigma=1.5)
  pre3 = vol[3:27, 3:43, 3:43].copy()
  post3 = vol[2:26, 5:45, 2:42].copy()
  f3 = calc.flow_field(pre3, post3, (16, 24, 24), 8, batch_size=4)
  save('flow3d', pre=pre3, post=post3, plain=f3)


# ---------------------------------------------------------------------------
def cfg_dict(cfg):
  return {k: (list(v) if isinstance(v, tuple) else v)
          for k, v in dataclasses.asdict(cfg).items()}


def gen_mesh():
  import json
  rng = np.random.default_rng(14)
  out = {}
  x2 = (rng.standard_normal((2, 2, 12, 14)) * 3).astype(np.float32)
  out['x2'] = x2
  for poo in (False, True):
    out[f'f2_{int(poo)}'] = np.array(
        rmesh.inplane_force(x2, 0.1, (40.0, 30.0), poo))
  x3 = (rng.standard_normal((3, 6, 7, 8)) * 2).astype(np.float32)
  x3b = (rng.standard_normal((3, 2, 5, 6, 7)) * 2).astype(np.float32)
  out['x3'] = x3
  out['x3b'] = x3b
  for poo in (False, True):
    out[f'f3_{int(poo)}'] = np.array(
        rmesh.elastic_mesh_3d(x3, 0.1, (20.0, 25.0, 14.0), poo))
    out[f'f3b_{int(poo)}'] = np.array(
        rmesh.elastic_mesh_3d(x3b, 0.05, 16.0, poo))
  planar = ((1, 0, 0), (0, 1, 0), (1, 1, 0), (-1, 1, 0))
  out['f3_planar'] = np.array(
      rmesh.elastic_mesh_3d(x3, 0.1, (20.0, 25.0, 14.0), False, links=planar))
  # A fold: neighbouring nodes swapped, exercises sign factors and nan_to_num.
  xf = np.zeros((2, 1, 6, 6), np.float32)
  xf[0, 0, 2, 2] = 45.0
  xf[1, 0, 3, 3] = -41.0
  xf[0, 0, 4, 1] = -40.0   # coincides with left neighbour -> zero length
  out['xf'] = xf
  for poo in (False, True):
    out[f'ff_{int(poo)}'] = np.array(
        rmesh.inplane_force(xf, 0.1, (40.0, 40.0), poo))
  save('mesh_force', **out)

  out = {}
  cfgs = {}
  x0 = (rng.standard_normal((2, 2, 20, 24)) * 2).astype(np.float32)
  prev = (rng.standard_normal((2, 2, 20, 24)) * 4).astype(np.float32)
  prev[:, 0, 3:6, 4:9] = np.nan
  v0 = np.zeros_like(x0)
  out['x0'] = x0
  out['prev'] = prev

  def run_vv(tag, cfg, cap, prev_=prev, x_=x0, force=rmesh.inplane_force,
             **kw):
    st = rmesh.velocity_verlet(x_.copy(), np.zeros_like(x_), prev_, cfg,
                               force_cap=cap, mesh_force=force, **kw)
    cfgs[tag] = cfg_dict(cfg)
    cfgs[tag]['_force_cap'] = cap
    for i, name in enumerate(('x', 'v', 'a')):
      out[f'{tag}_{name}'] = np.array(st[i], np.float32)
    if len(st) > 3:
      out[f'{tag}_scal'] = np.array([float(s) for s in st[3:]], np.float64)

  base = dict(dt=0.01, gamma=0.0, k0=0.05, k=0.1, stride=(20.0, 20.0),
              max_iters=1000, stop_v_max=0.001)
  for n in (1, 10, 100):
    run_vv(f'fire{n}', rmesh.IntegrationConfig(num_iters=n, **base), 1e6)
  run_vv('fire_cap', rmesh.IntegrationConfig(
      num_iters=60, start_cap=0.02, final_cap=1.0, cap_upscale_every=7,
      prefer_orig_order=True, dt_max=30.0, **base), 0.02)
  run_vv('fire_drift', rmesh.IntegrationConfig(
      num_iters=25, remove_drift=True, **base), 1e6)
  dbase = dict(base)
  dbase['gamma'] = 0.5
  run_vv('damped10', rmesh.IntegrationConfig(num_iters=10, fire=False, **dbase),
         1e6)
  run_vv('noprev10', rmesh.IntegrationConfig(num_iters=10, **base), 1e6,
         prev_=None)

  x30 = (rng.standard_normal((3, 5, 8, 9)) * 1.5).astype(np.float32)
  prev3 = (rng.standard_normal((3, 5, 8, 9)) * 2).astype(np.float32)
  out['x30'] = x30
  out['prev3'] = prev3
  b3 = dict(base)
  b3['stride'] = (20.0, 20.0, 12.0)
  run_vv('fire3d_20', rmesh.IntegrationConfig(num_iters=20, **b3), 1e6,
         prev_=prev3, x_=x30, force=rmesh.elastic_mesh_3d)
  out['cfgs'] = np.array(json.dumps(cfgs))
  save('mesh_vv', **out)

  # relax_mesh end-to-end
  out = {}
  cfgs = {}
  xr = np.zeros((2, 1, 30, 30))
  xr[0, 0, 10:20, 6] = 3
  xr[0, 0, 10:20, 24] = -4
  xr[1, 0, 20, 6:12] = 2
  out['xr'] = xr
  cfg = rmesh.IntegrationConfig(dt=0.01, gamma=0.0, k0=0.1, k=0.1,
                                stride=(10, 10), num_iters=100,
                                max_iters=10000, stop_v_max=0.001, fire=True)
  xs, ek, t = rmesh.relax_mesh(xr.copy(), np.zeros_like(xr), cfg)
  cfgs['fire'] = cfg_dict(cfg)
  out['fire_x'] = np.array(xs, np.float32)
  out['fire_ekin'] = np.array(ek)
  out['fire_t'] = np.array(t)

  cfg = rmesh.IntegrationConfig(
      dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40), num_iters=50,
      max_iters=400, stop_v_max=0.005, dt_max=1000, start_cap=0.01,
      final_cap=10, prefer_orig_order=True)
  xe = np.zeros((2, 2, 20, 24), np.float32)
  pe = (rng.standard_normal((2, 2, 20, 24)) * 5).astype(np.float32)
  pe = ndimage.gaussian_filter(pe, (0, 0, 2, 2)).astype(np.float32) * 4
  pe[:, 1, :4, :5] = np.nan
  out['xe'] = xe
  out['pe'] = pe
  xs, ek, t = rmesh.relax_mesh(xe.copy(), pe, cfg)
  cfgs['em2d'] = cfg_dict(cfg)
  out['em2d_x'] = np.array(xs, np.float32)
  out['em2d_ekin'] = np.array(ek)
  out['em2d_t'] = np.array(t)
  out['cfgs'] = np.array(json.dumps(cfgs))
  save('mesh_relax', **out)


def gen_maps():
  rng = np.random.default_rng(15)
  m1 = (rng.standard_normal((2, 2, 9, 11)) * 6).astype(np.float32)
  m2 = (rng.standard_normal((2, 2, 12, 10)) * 6).astype(np.float32)
  m1[:, 0, 2, 3] = np.nan
  m2[:, 1, 5:7, 4] = np.nan
  out = dict(m1=m1, m2=m2)
  for mode in ('nearest', 'constant'):
    out[f'c2_{mode}'] = np.array(rmap.compose_maps_fast(
        m1, (0, 10, 20), (16, 16), m2, (0, -5, 8), (20, 20), mode=mode))
  n1 = (rng.standard_normal((3, 4, 5, 6)) * 3).astype(np.float32)
  n2 = (rng.standard_normal((3, 5, 6, 7)) * 3).astype(np.float32)
  out['n1'] = n1
  out['n2'] = n2
  for mode in ('nearest', 'constant'):
    out[f'c3_{mode}'] = np.array(rmap.compose_maps_fast(
        n1, (1, 2, 3), (8, 10, 10), n2, (0, 1, 2), (8, 10, 10), mode=mode))
  save('compose_maps', **out)


def gen_montage3d():
  """compute_target_mesh for a synthetic 2 x 2 montage of 3-D tiles: hand-made
  neighbour table (11 fields), smooth random meshes and flows."""
  import functools as ft
  from sofima import stitch_elastic
  jax = sys.modules['jax']
  jnp = sys.modules['jax.numpy']
  rng = np.random.default_rng(77)
  n, (mz, my, mx) = 4, (6, 10, 12)
  stride = (8.0, 16.0, 16.0)  # zyx

  def smooth(shape, amp):
    a = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 1, 1.5, 1.5))
    return (a / np.abs(a).max() * amp).astype(np.float32)

  x = smooth((3, n, mz, my, mx), 6.0)
  fx = smooth((3, n, 5, 9, 4), 5.0)    # horizontal pairs: z, ortho (y), overlap (x)
  fy = smooth((3, n, 5, 3, 11), 5.0)   # vertical pairs:   z, overlap (y), ortho (x)
  fx[:, 0, 1, 2, 1] = np.nan
  fy[:, 2, 3, 1, 5] = np.nan
  x[:, 3, 2, 4, 4] = np.nan
  nb = -np.ones((n, 4, 11), np.int32)
  NI = stitch_elastic.NeighborInfo

  def entry(nbor, flow_idx, dim, flow, off_o, off_z):
    e = -np.ones(11, np.int32)
    e[NI.nbor_idx], e[NI.flow_idx], e[NI.dim] = nbor, flow_idx, dim
    fz, fyy, fxx = flow.shape[-3:]
    e[NI.flow_size_z] = fz
    e[NI.flow_size_overlap] = fxx if dim == 0 else fyy
    e[NI.flow_size_ortho] = fyy if dim == 0 else fxx
    e[NI.coarse_offset_ortho], e[NI.coarse_offset_z] = off_o, off_z
    e[NI.fine_off_x], e[NI.fine_off_y], e[NI.fine_off_z] = rng.integers(-3, 4, 3)
    return e

  offs = {0: (3, -2), 1: (-4, 1), 2: (0, 0), 3: (2, 2)}
  for t in range(n):
    tx, ty = t % 2, t // 2
    k = 0
    if tx == 1:   # left neighbour: flow stored with the left tile
      nb[t, k] = entry(t - 1, t - 1, 0, fx, *offs[t - 1]); k += 1
    if tx == 0:
      nb[t, k] = entry(t + 1, t, 0, fx, *offs[t]); k += 1
    if ty == 1:
      nb[t, k] = entry(t - 2, t - 2, 1, fy, *offs[(t - 2 + 1) % 4]); k += 1
    if ty == 0:
      nb[t, k] = entry(t + 2, t, 1, fy, *offs[(t + 1) % 4]); k += 1
  tf = ft.partial(stitch_elastic.compute_target_mesh, x=jnp.asarray(x),
                  fx=jnp.asarray(fx), fy=jnp.asarray(fy), stride=stride)
  r = np.asarray(jax.vmap(tf)(jnp.asarray(nb)))
  tg = np.transpose(r, [1, 0, 2, 3, 4]).astype(np.float32)
  assert np.isfinite(tg).mean() > 0.1
  save('montage3d', x=x, fx=fx, fy=fy, nbors=nb, stride=np.array(stride, np.float32),
       tg=tg)


def gen_mask_irregular():
  """map_utils.mask_irregular (pure NumPy / SciPy in the reference)."""
  rng = np.random.default_rng(33)
  yy, xx = np.mgrid[:37, :45]
  m = np.stack([6 * np.sin(yy / 6.0) * np.cos(xx / 9.0),
                5 * np.cos(yy / 7.0 + xx / 11.0)]).astype(np.float32)
  m += rng.standard_normal(m.shape).astype(np.float32) * 0.8
  for _ in range(12):   # folds / stretches
    y, x = rng.integers(0, 37), rng.integers(0, 45)
    m[rng.integers(0, 2), y, x] += rng.choice([-1, 1]) * rng.uniform(15, 40)
  m[:, 3, 4] = np.nan
  m[0, 30, 44] = np.nan
  out = {'m': m}
  for tag, kw in (('a', dict(frac=0.25, max_frac=1.1)),
                  ('b', dict(frac=0.4)),
                  ('c', dict(frac=0.25, max_frac=1.5, dilation_iters=0)),
                  ('d', dict(frac=0.3, max_frac=1.3, dilation_iters=3))):
    mm = m.copy()
    bad = rmap.mask_irregular(mm, (20.0, 16.0), **kw)
    out['bad_' + tag] = bad
    out['map_' + tag] = mm
  save('mask_irregular', **out)


def gen_clean_flow():
  """flow_utils.clean_flow (pure NumPy / SciPy in the reference)."""
  from sofima import flow_utils as rfu
  rng = np.random.default_rng(21)
  out = {}
  # 2-D: a smooth field with outliers, NaNs, infs, weak peaks
  f = np.zeros((4, 2, 33, 41), np.float32)
  yy, xx = np.mgrid[:33, :41]
  f[0] = 3 * np.sin(xx / 7.0) + rng.standard_normal((2, 33, 41)) * 0.3
  f[1] = 2 * np.cos(yy / 5.0) + rng.standard_normal((2, 33, 41)) * 0.3
  f[2] = 1.0 + rng.random((2, 33, 41)) * 3
  f[3] = rng.random((2, 33, 41)) * 3
  f[3][rng.random((2, 33, 41)) < 0.2] = 0.0
  for _ in range(40):
    z, y, x = rng.integers(0, 2), rng.integers(0, 33), rng.integers(0, 41)
    f[rng.integers(0, 2), z, y, x] += rng.choice([-1, 1]) * rng.uniform(4, 30)
  f[:, 0, 0, 0] = np.nan
  f[0, 1, 5, 5] = np.nan
  f[1, 0, 32, 40] = np.inf
  f[0, 1, 16, 0] = -np.inf
  f[:2, 0, 10:14, 20:23] = np.nan
  out['f2'] = f
  out['p2'] = np.array([1.5, 1.6, 8.0, 2.5], np.float32)
  out['c2'] = rfu.clean_flow(f.copy(), 1.5, 1.6, 8.0, 2.5)
  out['c2_nomag'] = rfu.clean_flow(f.copy(), 1.5, 1.6, 0.0, 2.5)
  out['c2_nodev'] = rfu.clean_flow(f.copy(), 1.5, 1.6, 8.0, 0.0)
  out['c2_2ch'] = rfu.clean_flow(f[:2].copy(), 1.5, 1.6, 8.0, 2.5)
  # 3-D: [5, z, y, x]
  g = (rng.standard_normal((5, 6, 9, 11)) * 1.5).astype(np.float32)
  g[3] = 1.0 + rng.random((6, 9, 11)) * 3
  g[4] = rng.random((6, 9, 11)) * 3
  for _ in range(25):
    z, y, x = rng.integers(0, 6), rng.integers(0, 9), rng.integers(0, 11)
    g[rng.integers(0, 3), z, y, x] += rng.choice([-1, 1]) * rng.uniform(5, 20)
  g[:, 2, 3, 4] = np.nan
  g[1, 5, 8, 10] = np.nan
  out['f3'] = g
  out['p3'] = np.array([1.2, 1.4, 9.0, 3.0], np.float32)
  out['c3'] = rfu.clean_flow(g.copy(), 1.2, 1.4, 9.0, 3.0, dim=3)
  out['c3_3ch'] = rfu.clean_flow(g[:3].copy(), 1.2, 1.4, 9.0, 3.0, dim=3)
  save('clean_flow', **out)


def _cfg1_tiles():
  """The four 512^2 tiles of the configs[0] montage (same recipe and seed as
  gen_montage, so both fixtures describe one chain)."""
  rng = np.random.default_rng(1001)
  t, ov = 512, 64
  canvas = ndimage.gaussian_filter(rng.standard_normal((2 * t, 2 * t)), 2.0)
  canvas = ((canvas - canvas.min()) / (canvas.max() - canvas.min()) * 255
            ).astype(np.uint8)
  jit = {(0, 0): (0, 0), (1, 0): (3, -2), (0, 1): (-4, 5), (1, 1): (2, 1)}
  tile_map = {}
  for (tx, ty), (dy, dx) in jit.items():
    y0 = 20 + ty * (t - ov) + dy
    x0 = 20 + tx * (t - ov) + dx
    tile_map[(tx, ty)] = canvas[y0:y0 + t, x0:x0 + t]
  return tile_map


def gen_stitch():
  """configs[0] flow leg and the stitch_rigid callers of the hot path, through
  the reference: _estimate_offset / compute_coarse_offsets (whole-overlap masked
  correlation), elastic_tile_mesh[_3d] + optimize_coarse_mesh (relax_mesh with a
  custom mesh_force), compute_flow_map (the 512^2 strips, patch 64 step 32)."""
  import json
  from sofima import stitch_rigid, stitch_elastic
  jnp = sys.modules['jax.numpy']
  rng = np.random.default_rng(2024)
  tile_map = _cfg1_tiles()
  out = {}
  keys = sorted(tile_map)
  out['tile_keys'] = np.array(keys, np.int32)
  out['tiles'] = np.stack([tile_map[k] for k in keys])
  cx, cy = stitch_rigid.compute_coarse_offsets(
      (2, 2), tile_map, overlaps_xy=((96, 128), (96, 128)), min_overlap=32)
  out['cx'], out['cy'] = cx, cy
  # direct _estimate_offset calls: plain, custom masks, other filter / range
  a = tile_map[(0, 0)][:, -96:]
  b = tile_map[(1, 0)][:, :96]
  off, pr = stitch_rigid._estimate_offset(a, b, 10)
  out['eo0'] = np.array(off + [pr], np.float64)
  ma = np.zeros(a.shape, bool)
  ma[:100, :40] = True
  mb = rng.random(b.shape) < 0.05
  out['eo1_ma'], out['eo1_mb'] = ma, mb
  off, pr = stitch_rigid._estimate_offset(a, b, 40, filter_size=7, masks=(ma, mb))
  out['eo1'] = np.array(off + [pr], np.float64)
  off, pr = stitch_rigid._estimate_offset(a, b, 250, filter_size=7, masks=(ma, mb))
  out['eo3'] = np.array(off + [pr], np.float64)   # everything masked -> NaN
  a2 = tile_map[(0, 0)][-128:, :]
  b2 = tile_map[(0, 1)][:128, :]
  off, pr = stitch_rigid._estimate_offset(a2, b2, 0)
  out['eo2'] = np.array(off + [pr], np.float64)
  coarse = stitch_rigid.optimize_coarse_mesh(cx, cy)
  out['coarse'] = np.asarray(coarse, np.float32)

  # tile-mesh forces on a 3 x 4 tile grid (NaN = missing tile pair)
  tx = (rng.standard_normal((2, 1, 3, 4)) * 20).astype(np.float32)
  tcx = (rng.standard_normal((2, 1, 3, 4)) * 30 - [[[[400]]], [[[0]]]]).astype(np.float32)
  tcy = (rng.standard_normal((2, 1, 3, 4)) * 30 - [[[[0]]], [[[400]]]]).astype(np.float32)
  tcx[:, 0, 1, 2] = np.nan
  tcy[:, 0, 0, 3] = np.nan
  out['tm_x'], out['tm_cx'], out['tm_cy'] = tx, tcx, tcy
  out['tm_f'] = np.asarray(stitch_rigid.elastic_tile_mesh(
      jnp.asarray(tx), jnp.asarray(tcx), jnp.asarray(tcy)), np.float32)
  cfg = rmesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.0, k=0.1, stride=(1, 1),
                                num_iters=1000, max_iters=20000, stop_v_max=0.001,
                                dt_max=100)
  out['tm_cfg'] = np.array(json.dumps(cfg_dict(cfg)))
  out['tm_relaxed'] = np.asarray(
      stitch_rigid.optimize_coarse_mesh(tcx, tcy, cfg), np.float32)
  t3x = (rng.standard_normal((3, 2, 3, 4)) * 20).astype(np.float32)
  t3cx = (rng.standard_normal((3, 2, 3, 4)) * 30).astype(np.float32)
  t3cy = (rng.standard_normal((3, 2, 3, 4)) * 30).astype(np.float32)
  t3cx[:, 1, 2, 1] = np.nan
  out['tm3_x'], out['tm3_cx'], out['tm3_cy'] = t3x, t3cx, t3cy
  out['tm3_f'] = np.asarray(stitch_rigid.elastic_tile_mesh_3d(
      jnp.asarray(t3x), jnp.asarray(t3cx), jnp.asarray(t3cy)), np.float32)
  out['tm3_relaxed'] = np.asarray(stitch_rigid.optimize_coarse_mesh(
      t3cx, t3cy, cfg, mesh_fn=stitch_rigid.elastic_tile_mesh_3d), np.float32)

  # configs[0] flow leg: fine flow of every adjacent tile pair
  stride = (32, 32)
  for name, conn, axis in (('fx', cx[:, 0], 0), ('fy', cy[:, 0], 1)):
    flows, offs = stitch_elastic.compute_flow_map(
        tile_map, conn, axis, patch_size=(64, 64), stride=stride, batch_size=64)
    ks = sorted(flows)
    out[name + '_keys'] = np.array(ks, np.int32)
    out[name + '_offsets'] = np.array([offs[k] for k in ks], np.int32)
    for i, k in enumerate(ks):
      out[f'{name}_{i}'] = np.asarray(flows[k], np.float32)
  save('stitch_cfg1', **out)


def gen_montage():
  """2 x 2 montage of 512^2 tiles through the reference's stitch_rigid /
  stitch_elastic chain: the inputs and outputs of
  compute_target_mesh (the prev_fn of the montage relaxation) and the relaxed
  mesh."""
  import functools as ft
  import json
  from sofima import stitch_rigid, stitch_elastic, flow_utils
  jax = sys.modules['jax']
  jnp = sys.modules['jax.numpy']
  rng = np.random.default_rng(1001)
  t, ov = 512, 64
  canvas = ndimage.gaussian_filter(rng.standard_normal((2 * t, 2 * t)), 2.0)
  canvas = ((canvas - canvas.min()) / (canvas.max() - canvas.min()) * 255
            ).astype(np.uint8)
  jit = {(0, 0): (0, 0), (1, 0): (3, -2), (0, 1): (-4, 5), (1, 1): (2, 1)}
  tile_map = {}
  for (tx, ty), (dy, dx) in jit.items():
    y0 = 20 + ty * (t - ov) + dy
    x0 = 20 + tx * (t - ov) + dx
    tile_map[(tx, ty)] = canvas[y0:y0 + t, x0:x0 + t]
  cx, cy = stitch_rigid.compute_coarse_offsets(
      (2, 2), tile_map, overlaps_xy=((96, 128), (96, 128)), min_overlap=32)
  coarse = stitch_rigid.optimize_coarse_mesh(cx, cy)
  stride = (32, 32)
  cx2, cy2 = cx[:, 0], cy[:, 0]
  fx_, offx = stitch_elastic.compute_flow_map(
      tile_map, cx2, 0, patch_size=(64, 64), stride=stride, batch_size=64)
  fy_, offy = stitch_elastic.compute_flow_map(
      tile_map, cy2, 1, patch_size=(64, 64), stride=stride, batch_size=64)
  kw = dict(min_peak_ratio=1.4, min_peak_sharpness=1.4, max_deviation=5,
            max_magnitude=0)
  fine_x = {k: flow_utils.clean_flow(v[:, None], **kw)[:, 0]
            for k, v in fx_.items()}
  fine_y = {k: flow_utils.clean_flow(v[:, None], **kw)[:, 0]
            for k, v in fy_.items()}
  fx, fy, x, nbors, key_to_idx = stitch_elastic.aggregate_arrays(
      (cx2, fine_x, offx), (cy2, fine_y, offy), list(tile_map.keys()),
      coarse[:, 0], stride=stride, tile_shape=(t, t))
  fx = np.asarray(fx, np.float32)
  fy = np.asarray(fy, np.float32)
  x = np.asarray(x, np.float32)
  # a non-trivial mesh state so the bilinear part is exercised
  xs = x + (rng.standard_normal(x.shape) * 1.5).astype(np.float32)

  def prev_fn(xx):
    tf = ft.partial(stitch_elastic.compute_target_mesh, x=xx, fx=fx, fy=fy,
                    stride=stride)
    r = jax.vmap(tf)(nbors)
    return jnp.transpose(r, [1, 0, 2, 3])

  tg0 = np.asarray(prev_fn(jnp.asarray(x)))
  tg1 = np.asarray(prev_fn(jnp.asarray(xs)))
  cfg = rmesh.IntegrationConfig(
      dt=0.001, gamma=0., k0=0.01, k=0.1, stride=stride, num_iters=100,
      max_iters=200, stop_v_max=0.001, dt_max=100, prefer_orig_order=True,
      start_cap=0.1, final_cap=10., remove_drift=True)
  xr, ek, tt = rmesh.relax_mesh(x.copy(), None, cfg, prev_fn=prev_fn)
  save('montage', fx=fx, fy=fy, x=x, xs=xs, nbors=np.asarray(nbors, np.int32),
       tg0=tg0, tg1=tg1, relaxed=np.asarray(xr, np.float32),
       ekin=np.asarray(ek), t=np.asarray(tt),
       cfg=np.array(json.dumps(cfg_dict(cfg))), stride=np.asarray(stride))
  print('target finite fraction', np.isfinite(tg0[0]).mean(axis=(1, 2)))


def flow_map3d_tiles():
  """2 x 2 grid of 3-d uint8 tiles cut from one EM-like volume with jittered
  positions; returns (tile_map, tile_shape xyz, offsets_x, offsets_y)."""
  rng = np.random.default_rng(55)
  tz, ty, tx = 40, 56, 64
  vol = ndimage.gaussian_filter(rng.standard_normal((tz + 24, 2 * ty + 24, 2 * tx + 24)), 1.5)
  vol = ((vol - vol.min()) / (vol.max() - vol.min()) * 255).astype(np.uint8)
  # xyz position of every tile in the volume: grid step (tx - 22, ty - 18) + jitter
  # (ortho / z jitters of up to 14 px: rounded to the stride on either side of 0)
  pos = {(0, 0): (8, 9, 8), (1, 0): (8 + tx - 23, 22, 1), (0, 1): (2, 9 + ty - 17, 20),
         (1, 1): (2 + tx - 25, 9 + ty - 19 - 14, 13)}
  tiles = {k: vol[None, z:z + tz, y:y + ty, x:x + tx].copy() for k, (x, y, z) in pos.items()}
  ox = np.full((3, 1, 2, 2), np.nan)
  oy = np.full((3, 1, 2, 2), np.nan)
  for (x, y), p in pos.items():
    if (x + 1, y) in pos:
      ox[:, 0, y, x] = np.array(pos[x + 1, y]) - np.array(p) - (tx, 0, 0)
    if (x, y + 1) in pos:
      oy[:, 0, y, x] = np.array(pos[x, y + 1]) - np.array(p) - (0, ty, 0)
  return tiles, (tx, ty, tz), ox, oy


def gen_flow_map3d():
  """stitch_elastic.compute_flow_map3d (stitch_elastic.py:85-194) on the tiles
  above: flow arrays and recorded offsets of every horizontal / vertical pair."""
  from sofima import stitch_elastic
  tiles, shape, ox, oy = flow_map3d_tiles()
  out = dict(tile_keys=np.array(list(tiles)), tiles=np.stack(list(tiles.values())),
             tile_shape=np.array(shape), ox=ox, oy=oy,
             patch=np.array((16, 20, 20)), stride=np.array((8, 10, 10)))
  for name, om, axis in (('fx', ox, 0), ('fy', oy, 1)):
    flows, offs = stitch_elastic.compute_flow_map3d(
        tiles, shape, om, axis, patch_size=(16, 20, 20), stride=(8, 10, 10), batch_size=8)
    keys = list(flows)
    out[name + '_keys'] = np.array(keys)
    out[name + '_offsets'] = np.array([offs[k] for k in keys], dtype=np.float64)
    for i, k in enumerate(keys):
      out[f'{name}_{i}'] = flows[k].astype(np.float32)
      print('flow_map3d', name, k, flows[k].shape, 'offset', offs[k],
            'valid', np.isfinite(flows[k][0]).sum())
  save('flow_map3d', **out)


def relax_passes_cases():
  """Inputs of the three-pass driver cases (shared with tests/test_gpu_mesh.py)."""
  cases = {}
  for case in ('regularized', 'regular', 'prep_failed', 'masked', 'median'):
    rng = np.random.default_rng(8)
    shape = (2, 1, 40, 44)
    prev = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 4, 4)) * 60
    if case in ('regularized', 'masked', 'median'):
      prev[0, 0, 18:22, 20:24] += 60          # a local fold in the flow field
    if case == 'median':
      prev[0] += 25.0                         # a global offset: PREV_MEDIAN start state
    frac = 0.7
    if case == 'prep_failed':
      # a 100 px tear: the free band of the soft pass is stretched beyond 1.1
      prev = ndimage.gaussian_filter(rng.standard_normal(shape), (0, 0, 4, 4)) * 20
      prev[0, 0, :, 22:] += 100
      frac = 0.9
    mask = None
    if case == 'masked':
      mask = np.zeros((1, 40, 44), bool)
      mask[0, :3, :] = True
    cases[case] = dict(prev=prev.astype(np.float32), frac=frac, mask=mask,
                       median=(case == 'median'))
  kw = dict(dt=0.001, gamma=0.0, k0=0.3, k=0.1, stride=(40, 40), num_iters=100,
            max_iters=400, stop_v_max=0.005, dt_max=1000, start_cap=1e6,
            final_cap=1e6, prefer_orig_order=True)
  return cases, kw


def gen_relax_passes():
  """processor/mesh.py:428-513 RelaxMesh.relax_mesh, the REFERENCE method run on
  an instance that only carries `_config` (all it touches): relax -> fold test ->
  soft relaxation -> fold test -> final relaxation."""
  import types
  from sofima.processor import mesh as rproc
  cases, kw = relax_passes_cases()
  cfg = rmesh.IntegrationConfig(**kw)
  out = dict(cfg=json.dumps(kw), names=np.array(list(cases)))
  for name, c in cases.items():
    proc = rproc.RelaxMesh.__new__(rproc.RelaxMesh)
    proc._config = types.SimpleNamespace(
        mesh_min_frac=c['frac'],
        options=rproc.MeshOptions(
            init_state=rproc.MeshInitState.PREV_MEDIAN if c['median']
            else rproc.MeshInitState.ZEROS))
    x0 = np.zeros_like(c['prev'])
    x, e_kin, steps, status = proc.relax_mesh(x0, c['prev'].copy(), cfg, c['mask'])
    print('relax_passes', name, 'status', int(status), 'steps', steps)
    out[f'{name}_prev'] = c['prev']
    out[f'{name}_frac'] = np.float64(c['frac'])
    out[f'{name}_mask'] = (np.zeros((0,), bool) if c['mask'] is None else c['mask'])
    out[f'{name}_median'] = np.bool_(c['median'])
    out[f'{name}_x'] = np.asarray(x, dtype=np.float32)
    out[f'{name}_ekin'] = np.asarray(e_kin, dtype=np.float64)
    out[f'{name}_steps'] = np.int64(steps)
    out[f'{name}_status'] = np.int64(int(status))
  save('relax_passes', **out)


def ndimage_warp_cases():
  """Inputs of the ndimage_warp fixtures (shared with nothing: stored in the
  .npz).  Boxes are xyz (start, size) pairs; None = not given."""
  rng = np.random.default_rng(4242)

  def field(shape, amp, sig):
    f = ndimage.gaussian_filter(rng.standard_normal(shape), (0,) + (sig,) * (len(shape) - 1))
    return (f / np.abs(f).max() * amp).astype(np.float32)

  img2 = em_like(rng, (96, 120))
  map2 = field((2, 9, 11), 5.0, 1.5)
  img3 = em_like(rng, (14, 44, 52), 1.5)
  map3 = field((3, 6, 9, 10), 2.5, 1.0)
  cases = {
      # name: image, map, stride zyx, work xyz, overlap xyz, order, boxes, out_scale
      'u8_2d': (img2, map2, (12, 12), (40, 32), (8, 8), 1, None, None),
      'u8_2d_nearest': (img2, map2, (12.0, 12.0), (64, 64), (0, 0), 0, None, None),
      'f32_2d': (img2.astype(np.float32) / 3, map2 * 2, (12, 12), (50, 50), (10, 4), 1, None,
                 (1.0, 1.0)),
      'u16_2d': (img2.astype(np.uint16) * 200, map2, (12, 12), (128, 128), (0, 0), 1, None, None),
      'u8_3d': (img3, map3, (3, 6, 6), (20, 20, 6), (4, 4, 2), 1, None, None),
      # map with context around the output box, output box inside the image box
      'u8_3d_boxes': (img3, map3, (3, 6, 6), (24, 16, 5), (6, 4, 1), 1,
                      dict(image=((100, 200, 10), (52, 44, 14)), map=((16, 33, 3), (10, 9, 6)),
                           out=((104, 203, 11), (40, 36, 10))), (1.0, 1.0, 1.0)),
      # output voxels twice the size of the image voxels in x and y
      'f32_3d_scale': (img3.astype(np.float32), map3 * 0.5, (3, 6, 6), (32, 32, 8), (0, 0, 0), 1,
                       dict(image=((0, 0, 0), (52, 44, 14)), map=((0, 0, 0), (10, 9, 6)),
                            out=((1, 2, 1), (22, 18, 12))), (2.0, 2.0, 1.0)),
  }
  return cases


def gen_ndimage_warp():
  """warp.ndimage_warp (warp.py:189-335), the SciPy-only rendering path: the
  reference function itself on small 2-d / 3-d images (uint8, uint16, float32;
  linear and nearest; work boxes with overlap; map / image / output boxes;
  out_scale)."""
  from sofima import warp as rwarp
  out = {'names': np.array(list(ndimage_warp_cases()))}
  for name, (img, cmap, stride, work, ov, order, boxes, scale) in ndimage_warp_cases().items():
    kw = {}
    if boxes is not None:
      for k in ('image', 'map', 'out'):
        kw[k + '_box'] = refshim.BoundingBox(start=boxes[k][0], size=boxes[k][1])
        out[f'{name}_{k}_box'] = np.array(boxes[k], np.int64)
    if scale is not None:
      kw['out_scale'] = scale
      out[f'{name}_out_scale'] = np.array(scale, np.float64)
    got = rwarp.ndimage_warp(img, cmap, stride, work, ov, order=order, parallelism=2, **kw)
    print('ndimage_warp', name, got.shape, got.dtype, 'non-zero %.3f' % (got != 0).mean())
    out[f'{name}_image'] = img
    out[f'{name}_map'] = cmap
    out[f'{name}_stride'] = np.array(stride, np.float64)
    out[f'{name}_stride_is_int'] = np.bool_(all(isinstance(v, int) for v in stride))
    out[f'{name}_work'] = np.array(work, np.int64)
    out[f'{name}_overlap'] = np.array(ov, np.int64)
    out[f'{name}_order'] = np.int64(order)
    out[f'{name}_warped'] = got
  save('ndimage_warp', **out)


if __name__ == '__main__':
  which = sys.argv[1:] or ['xcorr', 'peaks', 'flow', 'mesh', 'maps', 'clean', 'irregular', 'montage3d', 'montage', 'stitch', 'passes', 'flowmap3d', 'ndwarp']
  if 'ndwarp' in which:
    gen_ndimage_warp()
  if 'xcorr' in which:
    gen_xcorr_np()
  if 'peaks' in which:
    gen_peaks()
  if 'flow' in which:
    gen_flow()
  if 'mesh' in which:
    gen_mesh()
  if 'maps' in which:
    gen_maps()
  if 'clean' in which:
    gen_clean_flow()
  if 'irregular' in which:
    gen_mask_irregular()
  if 'montage3d' in which:
    gen_montage3d()
  if 'montage' in which:
    gen_montage()
  if 'stitch' in which:
    gen_stitch()
  if 'passes' in which:
    gen_relax_passes()
  if 'flowmap3d' in which:
    gen_flow_map3d()
