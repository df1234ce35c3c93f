"""Build-container-only stand-ins that let the UNMODIFIED reference files under
/root/reference be imported and executed where `jax`, `absl`, `connectomics`
and `dataclasses_json` are not installed.

This is test tooling (golden-vector generation), not product code and not a
copy of any reference code: it contains

  * no-op stubs for absl.logging / dataclasses_json,
  * `connectomics.common.{utils.batch, geom_utils.integral_image,
    geom_utils.query_integral_image, bounding_box.BoundingBox}` restated from
    their published behaviour (the package is not in this container),
  * a NumPy-backed subset of the `jax` API (`jit`, `vmap`, `lax.dynamic_slice`
    with JAX's start clamping, `lax.conv_general_dilated_patches` 'same',
    `lax.fori_loop/scan/cond`, `jnp.*` forwarding to NumPy with float64 results
    down-cast to float32 like JAX with x64 disabled, `.at[].set()`,
    `jax.scipy.ndimage.map_coordinates` order 1 with JAX's per-corner
    'constant' semantics).

Consequently golden vectors produced through it are "reference source executed
over a NumPy stand-in for JAX" -- NOT XLA numbers.  It never travels to the GPU
box in a form that is used there (nothing in tests -m gpu / smoke / bench
imports it) and it needs /root/reference to do anything.

Usage:   import refshim; refshim.install(); from sofima import flow_field
"""
import functools
import itertools
import operator
import os
import sys
import tempfile
import types

import numpy as np

_INSTALLED = False


def _mod(name, **attrs):
  m = types.ModuleType(name)
  m.__dict__.update(attrs)
  sys.modules[name] = m
  return m


# --------------------------------------------------------------------------
# jax.numpy stand-in
# --------------------------------------------------------------------------
class JArr(np.ndarray):
  """ndarray with the functional `.at[idx].set(v)` update of jax arrays."""

  @property
  def at(self):
    arr = self

    class _At:

      def __getitem__(self, idx):
        class _Upd:

          def set(self, v):
            out = np.array(arr, copy=True).view(JArr)
            out[idx] = v
            return out

          def add(self, v):
            out = np.array(arr, copy=True).view(JArr)
            np.add.at(out, idx, v)
            return out

        return _Upd()

    return _At()


def _down(x):
  """float64 -> float32, complex128 -> complex64 (JAX default, x64 off)."""
  if isinstance(x, np.ndarray):
    if x.dtype == np.float64:
      x = x.astype(np.float32)
    elif x.dtype == np.complex128:
      x = x.astype(np.complex64)
    elif x.dtype == np.int64:
      x = x.astype(np.int32)
    return x.view(JArr)
  if isinstance(x, np.floating) and not isinstance(x, np.float32):
    return np.float32(x)
  if isinstance(x, (tuple, list)) and x and all(
      isinstance(e, np.ndarray) for e in x
  ):
    return type(x)(_down(e) for e in x)
  return x


def _wrap(fn):
  @functools.wraps(fn)
  def w(*a, **k):
    return _down(fn(*a, **k))

  return w


class _NP(types.ModuleType):
  """Forwarding proxy: jnp.foo -> down-casting wrapper around np.foo."""

  def __init__(self, name, base):
    super().__init__(name)
    self._base = base

  def __getattr__(self, n):
    v = getattr(self._base, n)
    if isinstance(v, types.ModuleType):
      return _NP(self.__name__ + '.' + n, v)
    if callable(v) and not isinstance(v, type):
      return _wrap(v)
    return v


def _build_jax():
  jnp = _NP('jax.numpy', np)
  jnp.ndarray = np.ndarray
  jnp.float32 = np.float32
  jnp.int32 = np.int32
  jnp.inf = np.inf
  jnp.nan = np.nan
  jnp.r_ = np.r_

  def _asarray(x, dtype=None):
    return _down(np.array(x, dtype=dtype))

  jnp.asarray = _asarray
  jnp.array = _asarray

  def _nan_to_num(x, copy=True, nan=0.0, posinf=None, neginf=None):
    del copy
    return _down(
        np.nan_to_num(
            np.asarray(x), copy=True, nan=nan, posinf=posinf, neginf=neginf
        )
    )

  jnp.nan_to_num = _nan_to_num

  def _clip(x, min=None, max=None):  # pylint: disable=redefined-builtin
    return _down(np.clip(np.asarray(x), min, max))

  jnp.clip = _clip

  def jit(fn=None, **kw):
    del kw
    if fn is None:
      return lambda f: f
    return fn

  def vmap(fn):
    def w(*args):
      n = len(args[0])
      outs = [fn(*[a[i] for a in args]) for i in range(n)]
      return _down(np.stack([np.asarray(o) for o in outs]))

    return w

  def dynamic_slice(x, start, size):
    start = [int(s) for s in np.asarray(start).ravel()]
    sl = []
    for st, sz, dim in zip(start, size, np.shape(x)):
      st = max(0, min(st, dim - int(sz)))  # JAX clamps the start index
      sl.append(slice(st, st + int(sz)))
    return _down(np.asarray(x)[tuple(sl)])

  def dynamic_update_slice(x, upd, start):
    x = np.array(x, copy=True)
    start = [int(s) for s in start]
    sl = []
    for st, sz, dim in zip(start, np.shape(upd), x.shape):
      st = max(0, min(st, dim - sz))
      sl.append(slice(st, st + sz))
    x[tuple(sl)] = upd
    return _down(x)

  def dynamic_index_in_dim(x, idx, axis=0, keepdims=True):
    idx = int(idx)
    n = np.shape(x)[axis]
    idx = idx + n if idx < 0 else idx
    idx = max(0, min(idx, n - 1))
    r = np.take(np.asarray(x), idx, axis=axis)
    return _down(np.expand_dims(r, axis) if keepdims else r)

  def conv_general_dilated_patches(x, patch, strides, padding):
    # x: [b, 1, *spatial]; zero 'same' padding -> [b, prod(patch), *spatial]
    del strides
    assert padding == 'same' and x.shape[1] == 1
    x = np.asarray(x)[:, 0]
    dim = x.ndim - 1
    pads = [(0, 0)] + [((p - 1) // 2, p // 2) for p in patch]
    xp = np.pad(x, pads)
    win = np.lib.stride_tricks.sliding_window_view(
        xp, patch, axis=tuple(range(1, dim + 1))
    )
    win = win.reshape(win.shape[: dim + 1] + (-1,))
    return _down(np.moveaxis(win, -1, 1))

  def fori_loop(lo, hi, body, init):
    s = init
    for i in range(lo, hi):
      s = body(i, s)
    return s

  def cond(p, tf, ff, *ops):
    return tf(*ops) if bool(p) else ff(*ops)

  def scan(f, init, xs):
    c = init
    ys = []
    for i in range(len(xs)):
      c, y = f(c, xs[i])
      ys.append(y)
    return c, ys

  def map_coordinates(inp, coords, order, mode='constant', cval=0.0):
    # jax.scipy.ndimage.map_coordinates, order 1: every corner sample that is
    # out of range is replaced by `cval` BEFORE weighting ('constant'), or the
    # index is clamped ('nearest').
    assert order == 1
    inp = np.asarray(inp)
    coords = [np.asarray(c) for c in coords]
    per_dim = []
    for c, size in zip(coords, inp.shape):
      lo = np.floor(c)
      w_hi = c - lo
      lo = lo.astype(np.int64)
      items = []
      for idx, w in ((lo, 1 - w_hi), (lo + 1, w_hi)):
        if mode == 'nearest':
          items.append((np.clip(idx, 0, size - 1), None, w))
        elif mode == 'constant':
          valid = (idx >= 0) & (idx < size)
          items.append((np.clip(idx, 0, size - 1), valid, w))
        else:
          raise NotImplementedError(mode)
      per_dim.append(items)
    out = 0
    for combo in itertools.product(*per_dim):
      idxs = tuple(c[0] for c in combo)
      ws = functools.reduce(operator.mul, [c[2] for c in combo])
      contrib = inp[idxs]
      valids = [c[1] for c in combo if c[1] is not None]
      if valids:
        allv = functools.reduce(operator.and_, valids)
        contrib = np.where(allv, contrib, cval)
      out = out + ws * contrib
    return _down(np.asarray(out, dtype=np.float32))

  jax = types.ModuleType('jax')
  jax.jit = jit
  jax.vmap = vmap
  jax.Array = np.ndarray
  jax.numpy = jnp
  jax.lax = types.SimpleNamespace(
      dynamic_slice=dynamic_slice,
      dynamic_update_slice=dynamic_update_slice,
      dynamic_index_in_dim=dynamic_index_in_dim,
      conv_general_dilated_patches=conv_general_dilated_patches,
      fori_loop=fori_loop,
      cond=cond,
      scan=scan,
  )
  jax.tree_util = types.SimpleNamespace(
      register_dataclass=lambda *a, **k: None
  )
  jax.scipy = types.SimpleNamespace(
      ndimage=types.SimpleNamespace(map_coordinates=map_coordinates)
  )
  sys.modules['jax'] = jax
  sys.modules['jax.numpy'] = jnp
  return jax


# --------------------------------------------------------------------------
# connectomics / absl / dataclasses_json stubs
# --------------------------------------------------------------------------
def _integral_image(val):
  pads = []
  ii = val
  for axis in range(val.ndim):
    ii = ii.cumsum(axis=axis)
    pads.append([1, 0])
  return np.pad(ii, pads, mode='constant')


def _query_integral_image(svt, diam, stride):
  svt = np.asarray(svt)
  d, s = diam, stride
  if svt.ndim == 2:
    return (
        svt[d[0] :: s[0], d[1] :: s[1]]
        - svt[d[0] :: s[0], : -d[1] : s[1]]
        - svt[: -d[0] : s[0], d[1] :: s[1]]
        + svt[: -d[0] : s[0], : -d[1] : s[1]]
    )
  if svt.ndim == 3:
    hi = [np.s_[d[i] :: s[i]] for i in range(3)]
    lo = [np.s_[: -d[i] : s[i]] for i in range(3)]
    out = 0
    for bits in itertools.product((0, 1), repeat=3):
      sel = tuple(hi[i] if b else lo[i] for i, b in enumerate(bits))
      sign = (-1) ** (3 - sum(bits))
      out = out + sign * svt[sel].astype(np.int64)
    return out
  raise NotImplementedError


def _batch(iterable, n):
  it = iter(iterable)
  while True:
    b = list(itertools.islice(it, n))
    if not b:
      return
    yield b


class BoundingBox:
  """Minimal xyz-ordered box (start, size) sufficient for stitch_*.py."""

  def __init__(self, start=None, size=None, end=None):
    if start is not None:
      start = np.array(start)
    if size is not None:
      size = np.array(size)
    if end is not None:
      end = np.array(end)
    if start is None:
      start = end - size
    if size is None:
      size = end - start
    self.start = start
    self.size = size

  @property
  def end(self):
    return self.start + self.size

  def adjusted_by(self, start=None, end=None):
    s = self.start + (0 if start is None else np.array(start))
    e = self.end + (0 if end is None else np.array(end))
    return BoundingBox(start=s, end=e)

  def translate(self, off):
    return BoundingBox(start=self.start + np.array(off), size=self.size)

  def intersection(self, o):
    s = np.maximum(self.start, o.start)
    e = np.minimum(self.end, o.end)
    if np.any(e <= s):
      return None
    return BoundingBox(start=s, end=e)

  def to_slice3d(self):
    return tuple(
        slice(int(a), int(b)) for a, b in zip(self.start[::-1], self.end[::-1])
    )

  def to_slice4d(self):
    return (slice(None),) + self.to_slice3d()

  def __eq__(self, o):
    return bool(np.all(self.start == o.start) and np.all(self.size == o.size))

  def __repr__(self):
    return f'BB(start={self.start}, size={self.size})'


class BoxGenerator:
  """Stand-in for connectomics.common.box_generator.BoxGenerator with the three
  members warp.ndimage_warp touches (warp.py:279-321): an outer box cut into
  overlapping work boxes of `box_size` (stride box_size - overlap; boxes that
  would stick out are shifted back inside, `back_shift_small_boxes`), and for
  every work box the part of it that is written to the output -- here: half the
  overlap is given to either neighbour, so the cropped boxes tile the outer box
  exactly once.  For interpolation orders <= 1 (no spline prefilter over the
  work box) the warped output does not depend on how the work boxes are cut, so
  any exact tiling gives the reference's result; this is NOT a restatement of
  the connectomics class."""

  def __init__(self, outer_box, box_size, box_overlap=None, back_shift_small_boxes=False):
    self.outer = outer_box
    size = np.array(box_size)
    ov = np.zeros_like(size) if box_overlap is None else np.array(box_overlap)
    outer = np.array(outer_box.size)
    size = np.minimum(size, outer)
    self._starts, self._crops = [], []
    for d in range(len(size)):
      step = max(int(size[d] - ov[d]), 1)
      st = list(range(0, max(int(outer[d] - ov[d]), 1), step))
      st = sorted(set(min(s, int(outer[d] - size[d])) for s in st))
      cuts = [0] + [st[i] + int(ov[d]) // 2 for i in range(1, len(st))] + [int(outer[d])]
      self._starts.append([(s, int(size[d])) for s in st])
      self._crops.append([(cuts[i], cuts[i + 1]) for i in range(len(st))])
    self._shape = [len(s) for s in self._starts]

  @property
  def num_boxes(self):
    return int(np.prod(self._shape))

  def _coords(self, i):
    out = []
    for n in self._shape:
      out.append(i % n)
      i //= n
    return out

  def generate(self, i):
    c = self._coords(i)
    start = [self._starts[d][c[d]][0] for d in range(len(c))]
    size = [self._starts[d][c[d]][1] for d in range(len(c))]
    return tuple(c), BoundingBox(start=np.array(start) + self.outer.start, size=size)

  def index_to_cropped_box(self, i):
    c = self._coords(i)
    start = [self._crops[d][c[d]][0] for d in range(len(c))]
    end = [self._crops[d][c[d]][1] for d in range(len(c))]
    return BoundingBox(start=np.array(start) + self.outer.start, end=np.array(end) + self.outer.start)


def install(reference_root='/root/reference'):
  """Installs the stand-ins and makes `import sofima` resolve to the reference."""
  global _INSTALLED
  if _INSTALLED:
    return
  if not os.path.isdir(reference_root):
    raise RuntimeError(f'{reference_root} is not available')

  class _Log:

    def info(self, *a, **k):
      pass

    warning = error = debug = info

  absl = _mod('absl')
  absl.logging = _Log()
  _mod('absl.logging', info=_Log().info, warning=_Log().info)

  class DataClassJsonMixin:
    pass

  _mod('dataclasses_json', DataClassJsonMixin=DataClassJsonMixin)

  c = _mod('connectomics')
  cc = _mod('connectomics.common')
  cc.geom_utils = _mod(
      'connectomics.common.geom_utils',
      integral_image=_integral_image,
      query_integral_image=_query_integral_image,
  )
  class NPDataClassJsonMixin:
    pass

  cc.utils = _mod('connectomics.common.utils', batch=_batch,
                  NPDataClassJsonMixin=NPDataClassJsonMixin)
  cc.file = _mod('connectomics.common.file', PathLike=str)

  # connectomics.volume.*: only names that the class bodies / annotations of
  # processor/mesh.py touch at import time (RelaxMesh.relax_mesh itself uses
  # none of them: it reads self._config and calls sofima.mesh / map_utils).
  def _cls(name):
    return type(name, (), {})

  cv = _mod('connectomics.volume')
  cv.mask = _mod('connectomics.volume.mask', MaskConfigs=_cls('MaskConfigs'))
  cv.metadata = _mod('connectomics.volume.metadata',
                     DecoratedVolume=_cls('DecoratedVolume'))
  cv.subvolume = _mod('connectomics.volume.subvolume', Subvolume=_cls('Subvolume'))
  cv.subvolume_processor = _mod(
      'connectomics.volume.subvolume_processor',
      SubvolumeProcessor=_cls('SubvolumeProcessor'),
      SuggestedXyz=_cls('SuggestedXyz'), TupleOrSuggestedXyz=tuple)
  c.volume = cv
  cc.bounding_box = _mod(
      'connectomics.common.bounding_box',
      BoundingBox=BoundingBox,
      BoundingBoxBase=BoundingBox,
  )
  cc.box_generator = _mod('connectomics.common.box_generator', BoxGenerator=BoxGenerator)
  c.common = cc
  # warp.py imports these at module level; ndimage_warp (the only function the
  # goldens run) touches none of them for non-uint64 images
  cs = _mod('connectomics.segmentation')
  cs.labels = _mod('connectomics.segmentation.labels')
  c.segmentation = cs
  if 'cv2' not in sys.modules:
    _mod('cv2', INTER_NEAREST=0, INTER_LINEAR=1, INTER_CUBIC=2, INTER_LANCZOS4=4)
  if 'skimage' not in sys.modules:
    sk = _mod('skimage')
    sk.exposure = _mod('skimage.exposure')
  _build_jax()

  link_dir = tempfile.mkdtemp(prefix='sofima_ref_')
  os.symlink(reference_root, os.path.join(link_dir, 'sofima'))
  sys.path.insert(0, link_dir)
  _INSTALLED = True
