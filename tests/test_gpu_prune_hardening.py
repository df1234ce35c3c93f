"""Parity evidence for the timing-dependent exact pruning of the correlation
kernel (reference semantics: flow_field.py:205-275) -- -m gpu.

* the WHOLE field of the LITERAL bench input (bench.synth_pair with sensor
  noise, the pair `bench.py` times) against the oracle;
* many patches per workgroup (SFM_MFMA_GRID=1 / 2): the state a workgroup
  carries from patch to patch -- previous need mask, previous hot columns, the
  seed block, the probe's self-switch-off -- on a batch that INTERLEAVES all
  seven adversarial image kinds, so that every prediction is wrong;
* identity of every run-time switch of the kernel;
* a time-boxed soak + random-geometry fuzz with thousands of patches per call.
"""
import os
import time

import numpy as np
import pytest

from oracle import flow_oracle
from tests.util import check_sharpness

pytestmark = pytest.mark.gpu

KINDS = ['em', 'far', 'periodic', 'noise', 'edges', 'smooth', 'fine']
# kinds whose first peak is unambiguous: the float32 FFT of the oracle and the
# exact integer kernel rank the candidates of a lattice / a plateau differently
DISTINCT = ('em', 'far', 'edges')


def _mosaic(seed, h=460, w=500):
  """The seven adversarial image pairs of test_gpu_flow._prune_images side by
  side: region k holds kind KINDS[k]."""
  from tests.test_gpu_flow import _prune_images
  pairs = [_prune_images(k, seed + i, h, w) for i, k in enumerate(KINDS)]
  return (np.ascontiguousarray(np.concatenate([p[0] for p in pairs], axis=1)),
          np.ascontiguousarray(np.concatenate([p[1] for p in pairs], axis=1)))


def _interleaved_starts(rng, b, py, px, h=460, w=500):
  """Patch k lies in region k % 7: consecutive patches of a workgroup never
  share an image kind."""
  kind = np.arange(b) % len(KINDS)
  y = rng.integers(-10, h - py + 10, b)
  x = kind * w + rng.integers(0, w - px + 1, b)
  return np.stack([y, x], axis=1).astype(np.int32), kind


def test_literal_bench_input_whole_field_vs_oracle(gpu, tmp_path):
  """The pair bench.py times -- synth_pair(8192, 1002, warp=WARP): content
  shift, smooth 6 px / 2048 px deformation AND sigma-4 sensor noise, which lowers
  the NCC peak and with it every pruning / abandon decision -- all 40 reference
  batches (40401 patches) against the oracle, vector for vector."""
  import bench
  from sofima_amd import _abi, flow_field as ff
  from tests.util import oracle_flow_batches
  pre, post = bench.synth_pair(8192, 1002, warp=bench.WARP)
  calc = ff.JAXMaskedXCorrWithStatsCalculator()
  got = calc.flow_field(pre, post, bench.PATCH, bench.STEP, batch_size=bench.BATCH)
  assert got.shape == (4, 201, 201)
  parts = oracle_flow_batches(tmp_path, pre, post, bench.PATCH, bench.STEP,
                              bench.BATCH, range(40))
  want = np.concatenate([parts[b] for b in range(40)], axis=1).reshape(4, 201, 201)
  np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
  np.testing.assert_array_equal(got[:2], want[:2])
  np.testing.assert_allclose(got[3], want[3], rtol=1e-4, atol=1e-6)
  ok = np.isfinite(want[2])
  check_sharpness(got[2][ok], want[2][ok])
  assert ok.mean() > 0.99
  with _abi.option('SFM_MFMA_PRUNE', 0):
    full = calc.flow_field(pre, post, bench.PATCH, bench.STEP, batch_size=bench.BATCH)
  np.testing.assert_array_equal(got, full)


@pytest.mark.parametrize('py,px', [(160, 160), (96, 96), (50, 70)])
@pytest.mark.parametrize('mean', [None, 100.0])
def test_many_patches_per_workgroup_on_interleaved_kinds(gpu, py, px, mean):
  """98 patches, image kind changing from patch to patch, through ONE and TWO
  workgroups: pruned == un-pruned bit for bit, and the vectors of the kinds with
  an unambiguous peak equal the oracle's."""
  from sofima_amd import _abi, flow_field
  pre, post = _mosaic(41)
  rng = np.random.default_rng(py + px)
  b = 98
  starts, kind = _interleaved_starts(rng, b, py, px)
  kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5,
            post_patch_size=(py, px), post_starts=starts)
  args = (pre, post, None, None, (py, px), starts, mean)
  with _abi.option('SFM_MFMA_PRUNE', 0):
    full = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
  for grid in (1, 2, 0):
    with _abi.option('SFM_MFMA_GRID', grid):
      pruned = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
      np.testing.assert_array_equal(pruned, full, err_msg=f'grid {grid}')
      for name, val in (('SFM_MFMA_EARLY', 1), ('SFM_MFMA_NARROW', 0),
                        ('SFM_MFMA_LAZY', 0), ('SFM_MFMA_PROBE', 0)):
        with _abi.option(name, val):
          np.testing.assert_array_equal(
              flow_field.batched_xcorr_peaks(*args, method=2, **kw), full,
              err_msg=f'grid {grid} {name}={val}')
  # the oracle on the same batch (float64 surfaces: the ranking of near-equal
  # candidates is then the exact one; batch coupling as in the reference)
  off, surf = flow_oracle.batched_xcorr(pre, post, None, None, (py, px), starts, mean,
                                        (py, px), starts, workers=8)
  want = flow_oracle.batched_peaks(surf, off, 2, 0.5, 5)
  sel = np.isin(kind, [KINDS.index(k) for k in DISTINCT])
  np.testing.assert_array_equal(np.isnan(full[sel]), np.isnan(want[sel]))
  np.testing.assert_array_equal(full[sel, :2], want[sel, :2])
  ok = sel & np.isfinite(want[:, 2])
  check_sharpness(full[ok, 2], want[ok, 2])


@pytest.mark.parametrize('py,px', [(160, 160), (128, 128), (112, 112), (96, 96), (64, 64), (48, 48),
                                   (32, 48), (160, 128), (96, 160), (144, 80), (48, 160)])
@pytest.mark.parametrize('mean', [None, 100.0])
def test_prep_without_staging_matches_the_staged_form(gpu, py, px, mean):
  """The prep pass of the lazy-table mode reduces half blocks of pixels in
  registers (no patch in LDS, packed row / column words, a second pass) where the
  staged form sweeps an LDS copy: the same integers in the same tables, so the
  flow vectors and peak statistics are the same bits (SFM_MFMA_LAZYG=0: staged
  form + prep-built table; SFM_MFMA_PRUNE=0: every tile).  Patch sizes of every
  instantiated chunk geometry, non-square patches, starts clamped at the image
  borders and at every byte alignment; clipped / saturated content (integer
  centres away from the mean) in two of the seven regions."""
  from sofima_amd import _abi, flow_field
  pre, post = _mosaic(47)
  pre = pre.copy()
  post = post.copy()
  pre[:, 500:1000] = np.where(pre[:, 500:1000] > 128, 255, pre[:, 500:1000])   # saturated
  post[:, 1500:2000] = (post[:, 1500:2000] // 64) * 64                            # few levels
  rng = np.random.default_rng(3 * py + px)
  b = 301
  starts, _ = _interleaved_starts(rng, b, py, px)
  starts[:7, 1] = np.arange(7)                      # every alignment at the left border
  starts[7:14, 1] = pre.shape[1] - px - np.arange(7)   # ... and at the right one
  starts[14, 0] = -40                               # clamped to the top
  starts[15, 0] = pre.shape[0] + 5                  # ... the bottom
  kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5,
            post_patch_size=(py, px), post_starts=starts)
  args = (pre, post, None, None, (py, px), starts, mean)
  got = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
  with _abi.option('SFM_MFMA_LAZYG', 0):
    staged = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
  with _abi.option('SFM_MFMA_PRUNE', 0):
    full = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
  np.testing.assert_array_equal(got, staged)
  np.testing.assert_array_equal(got, full)
  for grid in (1, 3):
    with _abi.option('SFM_MFMA_GRID', grid):
      np.testing.assert_array_equal(flow_field.batched_xcorr_peaks(*args, method=2, **kw), got)


@pytest.mark.parametrize('mean', [None, 100.0])
def test_cross_patch_pipeline_is_bit_identical(gpu, mean):
  """SFM_MFMA_PIPE=1 (kModePipe: one workgroup of eight waves per CU, two patch
  slots in LDS, the tile queue running across the patch boundary, the closing wave
  recomputing / publishing / re-opening alone, openings cut into units any idle wave
  takes) against the two-workgroup kernel and the un-pruned run: the same bits on the
  batch that interleaves all seven adversarial image kinds, through 1, 3 and all
  workgroups (1 workgroup = both slots of ONE CU carry the whole batch: every slot
  hand-over, retirement with tiles still in flight in the other slot, batches of 1
  and 2 patches = a slot that never opens), with every admission limit and with the
  kernel's other switches on top."""
  from sofima_amd import _abi, flow_field
  pre, post = _mosaic(53)
  rng = np.random.default_rng(17)
  py = px = 160
  for b in (1, 2, 3, 98, 301):
    starts, _ = _interleaved_starts(rng, b, py, px)
    kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5,
              post_patch_size=(py, px), post_starts=starts)
    args = (pre, post, None, None, (py, px), starts, mean)
    ref = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
    if b == 98:
      with _abi.option('SFM_MFMA_PRUNE', 0):
        np.testing.assert_array_equal(flow_field.batched_xcorr_peaks(*args, method=2, **kw), ref)
    with _abi.option('SFM_MFMA_PIPE', 1):
      for grid in (0, 2, 6):
        with _abi.option('SFM_MFMA_GRID', grid):
          for admit in ((255, 2, 1) if b == 98 else (0,)):
            with _abi.option('SFM_MFMA_PIPE_ADMIT', admit):
              np.testing.assert_array_equal(
                  flow_field.batched_xcorr_peaks(*args, method=2, **kw), ref,
                  err_msg=f'batch {b} grid {grid} admit {admit}')
      if b == 98:
        for name, val in (('SFM_MFMA_EARLY', 0), ('SFM_MFMA_NARROW', 0), ('SFM_MFMA_PROBE', 0),
                          ('SFM_MFMA_WIDEN', 1)):
          with _abi.option(name, val), _abi.option('SFM_MFMA_GRID', 2):
            np.testing.assert_array_equal(
                flow_field.batched_xcorr_peaks(*args, method=2, **kw), ref,
                err_msg=f'{name}={val}')
  # other peak geometries (guard bands of 60 rows: the closer recomputes band tiles)
  starts, _ = _interleaved_starts(rng, 70, py, px)
  for radius, md, thr in ((30, 2, 0.9), (5, 2, 0.2), (12, 6, 0.5)):
    kw = dict(min_distance=md, threshold_rel=thr, peak_radius=radius,
              post_patch_size=(py, px), post_starts=starts)
    args = (pre, post, None, None, (py, px), starts, mean)
    ref = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
    with _abi.option('SFM_MFMA_PIPE', 1), _abi.option('SFM_MFMA_GRID', 4):
      np.testing.assert_array_equal(flow_field.batched_xcorr_peaks(*args, method=2, **kw), ref)


def test_cross_patch_pipeline_whole_bench_field(gpu):
  """The literal bench pair through the pipeline kernel: the whole [4, 201, 201]
  field equals the production kernel's (which the oracle test above pins)."""
  import bench
  from sofima_amd import _abi, flow_field as ff
  pre, post = bench.synth_pair(8192, 1002, warp=bench.WARP)
  calc = ff.JAXMaskedXCorrWithStatsCalculator()
  ref = calc.flow_field(pre, post, bench.PATCH, bench.STEP, batch_size=bench.BATCH)
  with _abi.option('SFM_MFMA_PIPE', 1):
    got = calc.flow_field(pre, post, bench.PATCH, bench.STEP, batch_size=bench.BATCH)
  np.testing.assert_array_equal(got, ref)


def test_every_kernel_switch_returns_the_same_bits(gpu):
  """The run-time switches of the correlation kernel select schedules, never
  results: PROBE, TOUCH_ALL, EXACT, QUEUE, PRIO, MAX_WG_PER_CU (measurement
  switches) next to PRUNE / LAZY / EARLY / WIDEN / NARROW / XCD / GRID."""
  from sofima_amd import _abi, flow_field
  pre, post = _mosaic(43)
  rng = np.random.default_rng(9)
  for (py, px), b in (((160, 160), 140), ((96, 96), 210)):
    starts, _ = _interleaved_starts(rng, b, py, px)
    kw = dict(min_distance=2, threshold_rel=0.5, peak_radius=5,
              post_patch_size=(py, px), post_starts=starts)
    args = (pre, post, None, None, (py, px), starts, None)
    want = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
    # (the first six are measurement-only: a production library ignores them -- run this
    # test with SOFIMA_AMD_LIB=.../libsofima_amd_measure.so, NOTIMING=1
    # tools/measure/build_timing_lib.sh, to exercise them)
    measure = (('SFM_MFMA_PROBE', (0,)), ('SFM_MFMA_TOUCH_ALL', (1,)),
               ('SFM_MFMA_EXACT', (0,)), ('SFM_MFMA_QUEUE', (0,)),
               ('SFM_MFMA_PRIO', (1, 2, 3)), ('SFM_MFMA_MAX_WG_PER_CU', (1,)))
    if _abi.get_option('SFM_BUILD_MEASUREMENT_SWITCHES') != '1':
      measure = ()
    for name, values in measure + (
                         ('SFM_MFMA_PRUNE', (0,)), ('SFM_MFMA_LAZY', (0,)), ('SFM_MFMA_LAZYG', (0,)),
                         ('SFM_MFMA_EARLY', (0, 1, 4)), ('SFM_MFMA_WIDEN', (1,)),
                         ('SFM_MFMA_NARROW', (0, 4)), ('SFM_MFMA_XCD', (1,)),
                         ('SFM_MFMA_GRID', (1, 3)), ('SFM_MFMA_PIPE', (1,))):
      for v in values:
        with _abi.option(name, v):
          got = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
        np.testing.assert_array_equal(got, want, err_msg=f'{name}={v} at {py}x{px}')


def _fuzz_case(rng):
  """Random texture / shift / noise / geometry / peak parameters with thousands
  of patches per call (every workgroup runs many patches)."""
  from scipy import ndimage
  h, w = (int(v) for v in rng.integers(700, 1500, 2))
  sig = float(rng.choice([0.8, 1.5, 2.0, 4.0, 10.0]))
  base = ndimage.gaussian_filter(rng.standard_normal((h + 200, w + 200), dtype=np.float32), sig)
  base = (base - base.min()) / (base.max() - base.min()) * 255
  if rng.random() < 0.3:
    dy, dx = (int(v) for v in rng.integers(-90, 91, 2))
  else:
    dy, dx = (int(v) for v in rng.integers(-8, 9, 2))
  pre = base[100:100 + h, 100:100 + w]
  post = (base[100 + dy:100 + dy + h, 100 + dx:100 + dx + w] +
          rng.standard_normal((h, w), dtype=np.float32) * float(rng.choice([0, 2, 8, 30])))
  if rng.random() < 0.2:
    pre = pre.copy()
    pre[: h // 3] = 77
  if rng.random() < 0.2:
    post = post.copy()
    post[:, ::int(rng.integers(5, 40))] = 255
  if rng.random() < 0.25:      # two unrelated halves: predictions flip from patch to patch
    post = post.copy()
    post[:, w // 2:] = rng.integers(0, 256, (h, w - w // 2))
  pre = np.clip(np.round(pre), 0, 255).astype(np.uint8)
  post = np.clip(np.round(post), 0, 255).astype(np.uint8)
  py = int(rng.choice([160, 160, 128, 96, 64, 48, 120, 100, 50, 33]))
  px = py if rng.random() < 0.7 else int(rng.choice([160, 128, 96, 64, 70, 112]))
  b = int(rng.integers(4096, 6145))
  starts = np.stack([rng.integers(-10, h - py + 10, b),
                     rng.integers(-10, w - px + 10, b)], axis=1).astype(np.int32)
  kw = dict(min_distance=int(rng.choice([1, 2, 2, 3, 7])),
            threshold_rel=float(rng.choice([0.5, 0.5, 0.3, 0.8, 0.1])),
            peak_radius=int(rng.choice([5, 5, 2, 9, 20])),
            post_patch_size=(py, px), post_starts=starts)
  mean = None if rng.random() < 0.7 else float(rng.uniform(0, 255))
  return (pre, post, None, None, (py, px), starts, mean), kw


def test_soak_and_fuzz_on_the_shipped_build(gpu):
  """Time-boxed (SFM_SOAK_SECONDS, default 45 s; tools/measure/soak.py and
  prune_fuzz.py are the long forms).  Soak: the 4096^2 bench-like field (10201
  patches per call, every workgroup runs ~20 patches) repeated -- the decisions
  depend on wave timing, the field must not.  Fuzz: random geometries with b >=
  4096 patches per call, pruned == un-pruned."""
  import torch
  import bench
  from sofima_amd import _abi, flow_field
  budget = float(os.environ.get('SFM_SOAK_SECONDS', '45'))
  pre, post = bench.synth_pair(4096, 7, warp=bench.WARP)
  a, b = torch.from_numpy(pre).cuda(), torch.from_numpy(post).cuda()
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  with _abi.option('SFM_MFMA_PRUNE', 0):
    ref = calc.flow_field(a, b, 160, 40, batch_size=1024)
  t_end = time.time() + budget / 3
  runs = 0
  while time.time() < t_end or runs < 20:
    got = calc.flow_field(a, b, 160, 40, batch_size=1024)
    assert np.array_equal(got, ref, equal_nan=True), f'soak run {runs} differs'
    runs += 1
  rng = np.random.default_rng(20260929)
  t_end = time.time() + 2 * budget / 3
  cases = 0
  while time.time() < t_end or cases < 6:
    args, kw = _fuzz_case(rng)
    pruned = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
    with _abi.option('SFM_MFMA_PRUNE', 0):
      full = flow_field.batched_xcorr_peaks(*args, method=2, **kw)
    assert np.array_equal(pruned, full, equal_nan=True), (
        f'fuzz case {cases}: patch {args[4]}, b {len(args[5])}, mean {args[6]}, '
        f'{ {k: v for k, v in kw.items() if k != "post_starts"} }')
    cases += 1
  print(f'soak: {runs} identical runs; fuzz: {cases} cases with b >= 4096, 0 mismatches')
