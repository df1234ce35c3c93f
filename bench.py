#!/usr/bin/env python3
"""Headline benchmark: patch-xcorr Mpix/s + mesh node-updates/s on 8192^2 tiles.

One "step" = one pass of the hot path over one synthetic 8192x8192 EM tile
pair (BASELINE.json configs[1]):
  1. flow_field(pre, post, patch 160, step 40, batch 1024)  -> [4, 201, 201]
  2. relax_mesh([2, 1, 205, 205], prev from that flow, em_2d FIRE config,
     1000 iterations)
with the images already resident in HBM.  The headline pair is SURVEY.md 8d's
"realistic pair" (content shift + smooth 6 px deformation + noise); --pair
exact selects the rigid-shift pair.  With --gpus N every rank processes
its own independent tile pair (weak scaling, no data-path collective); the time
is the max over ranks.

Prints ONE JSON line (rank 0).  `value` is patch-xcorr Mpix/s = post-image
pixels / wall time of flow_field(); the mesh figure is in the
`mesh` object.  `roofline` describes the dominant kernel (the patch-correlation
kernel), timed with HIP events through the library's sfm_profile_* hooks:
`roofline.frac` is the HARDWARE fraction -- the same kernel on the same pair
with the exact tile pruning switched off (a second leg of the same run), so
every algorithmic operation is executed; `frac_pruned` is the production launch
of the timed region (algorithmic operations / time: includes the skipped work);
`issued_frac` counts the matrix instructions the kernel really issued.  After
the timed region the step is repeated for >= --sustain seconds (`sustained`).
`aux_rooflines` carries HBM roofline lines for the FFT form of the correlation
and the volumetric / large in-plane mesh steps.  `cpu_baseline` times the CPU oracle (a NumPy/SciPy port
of the reference algorithm) on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PATCH, STEP, BATCH = 160, 40, 1024
MESH_ITERS = 1000
# MI355X peaks (MI355X_MICROARCH.md): dense int8 MFMA ~= 2x bf16 = 5.0 POP/s,
# f32 vector/MFMA 157.3 TFLOP/s, HBM 8 TB/s.
PEAK_I8_TOPS = 5000.0
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def synth_pair(size, seed, shift=(3, -5), warp=None):
  """EM-like texture pair: low-pass noise; the second image is
  the same content shifted by an integer vector, optionally sampled through the
  smooth deformation d(x, y) = A sin(2 pi x / L) cos(2 pi y / L) (warp = (A, L),
  bilinear), plus fresh sensor noise (sigma 4)."""
  from scipy import ndimage
  rng = np.random.default_rng(seed)
  m = 16
  base = ndimage.gaussian_filter(
      rng.standard_normal((size + 2 * m, size + 2 * m), dtype=np.float32), 2.0)
  base = (base - base.min()) / (base.max() - base.min()) * 255
  dy, dx = shift
  pre = base[m:m + size, m:m + size]
  post = base[m + dy:m + dy + size, m + dx:m + dx + size]
  if warp is not None:
    amp, lam = warp
    out = np.empty((size, size), np.float32)
    xx = np.arange(size, dtype=np.float32)[None, :]
    for y0 in range(0, size, 1024):
      yy = np.arange(y0, min(y0 + 1024, size), dtype=np.float32)[:, None]
      d = amp * np.sin(2 * np.pi * xx / lam) * np.cos(2 * np.pi * yy / lam)
      out[y0:y0 + 1024] = ndimage.map_coordinates(
          post, [yy + d, xx - d], order=1, mode='nearest', output=np.float32)
    post = out
  post = post + rng.standard_normal(post.shape, dtype=np.float32) * 4
  return (np.clip(pre, 0, 255).astype(np.uint8),
          np.clip(post, 0, 255).astype(np.uint8))


WARP = (6.0, 2048.0)   # SURVEY.md 8d "realistic pair"


def mesh_inputs(flow, pad):
  """prev = flow padded with NaN by patch//2//step nodes (em_alignment nb)."""
  f = np.pad(flow[:2], ((0, 0), (pad, pad), (pad, pad)),
             constant_values=np.nan)
  return f[:, None].astype(np.float32)


def build_sha():
  """Hash of the kernel sources in this tree (sofima_amd._build.source_hash)."""
  try:
    from sofima_amd import _build
    return _build.source_hash()
  except OSError:
    return None


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=3)
  ap.add_argument('--warmup', type=int, default=1)
  ap.add_argument('--size', type=int, default=8192)
  ap.add_argument('--method', type=int, default=0,
                  help='0 auto, 1 direct f32, 2 int8 MFMA, 3 FFT form')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--cpu-baseline-only', action='store_true',
                  help='internal: the CPU leg, run in a child process')
  ap.add_argument('--seed', type=int, default=1002)
  ap.add_argument('--mesh-iters', type=int, default=MESH_ITERS)
  ap.add_argument('--pair', choices=('warped', 'exact'), default='warped',
                  help='headline pair: smooth 6 px deformation (SURVEY 8d) or the '
                       'rigid integer shift')
  ap.add_argument('--sustain', type=float, default=5.0,
                  help='seconds of back-to-back steps after the timed region '
                       '(steady-state power / clocks); 0 = off')
  ap.add_argument('--no-legs', action='store_true',
                  help='skip the un-pruned and other-pair roofline legs')
  ap.add_argument('--multi-gpu-timeout', type=float, default=240.0,
                  help='N > 1: seconds a communicating leg may take before it is '
                       'abandoned (the line is printed without it)')
  ap.add_argument('--no-multi-gpu-legs', action='store_true',
                  help='N > 1: skip the communicating legs (configs[3] section '
                       'chain with the boundary hand-off, one mesh in bands '
                       'across the ranks)')
  ap.add_argument('--force-multi-gpu-legs', action='store_true',
                  help='N = 1: run the communicating legs anyway, over an nccl (RCCL) '
                       'process group of ONE rank, so that the nccl-only branches '
                       '(device all-gather of the boundary meshes, the library '
                       'communicator from a broadcast id, the watchdog) execute on a '
                       'single-GPU box; the banded mesh runs 2 local bands through '
                       'RCCL self send / recv')
  ap.add_argument('--aux-leg', default=None, metavar='LEG',
                  help='run ONE leg of aux_rooflines (1 warm-up + its timed calls) and '
                       'exit: the process rocprofv3 --pmc counts for that leg\'s '
                       '`traffic` (tools/measure/pmc_aux.sh)')
  ap.add_argument('--mesh-sharded', type=int, default=0, metavar='BANDS',
                  help='extra leg: one [2,64,204,204] mesh split into BANDS bands '
                       'per rank, stepped by the C-side banded loop (RCCL halo '
                       'exchange between ranks)')
  args = ap.parse_args()

  if args.cpu_baseline_only:
    cpu_baseline_child(args.size, args.seed, args.pair)
    return
  # wall clock of this process by phase (rank 0's view; `wall_s` in the line): what a
  # driver that bounds the whole run needs to know, beside the timed region
  wall = {}
  w_mark = [time.perf_counter()]

  def lap(name):
    now = time.perf_counter()
    wall[name] = round(wall.get(name, 0.0) + now - w_mark[0], 3)
    w_mark[0] = now
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    # `python bench.py --gpus N` without a launcher: start the N ranks here
    # (one process per GPU, RCCL), exactly as the driver's torchrun line does.
    sys.exit(spawn_ranks(args.gpus))

  # (multi-process GPU work on this driver stack needs dmabuf IPC: harmless when
  # already exported, decisive when a launcher forgot it)
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  import torch
  import torch.distributed as dist
  from sofima_amd import _abi, flow_field, mesh

  if args.aux_leg:
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    print(json.dumps({'aux_leg': aux_legs(dev, args.seed, only=args.aux_leg)}), flush=True)
    return

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  # SFM_BENCH_BACKEND=gloo + SFM_BENCH_ONE_DEVICE=1 let the multi-rank logic be
  # smoke-tested on a single-GPU box (all ranks share cuda:0).
  backend = os.environ.get('SFM_BENCH_BACKEND', 'nccl')
  if os.environ.get('SFM_BENCH_ONE_DEVICE'):
    local_rank = 0
  forced = bool(args.force_multi_gpu_legs) and world == 1
  if forced:
    import socket
    from sofima_amd import dist as _sdist
    _sdist.FORCE_COLLECTIVES = True
    if 'MASTER_PORT' not in os.environ:
      sock = socket.socket()
      sock.bind(('127.0.0.1', 0))
      os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
      sock.close()
  if world > 1 or forced:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend == 'nccl':
      dist.init_process_group('nccl', rank=rank, world_size=world,
                              device_id=torch.device('cuda', local_rank))
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)
  if world != args.gpus:
    raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
  ranks_seen = 1
  if world > 1 or forced:
    if backend == 'nccl':
      # every rank must sit on its own GPU and RCCL must see all of them
      assert torch.cuda.device_count() >= world, (torch.cuda.device_count(), world)
    probe = torch.ones(1, device=torch.device('cuda', local_rank) if backend == 'nccl'
                       else 'cpu')
    dist.all_reduce(probe)
    ranks_seen = int(probe.item())
    assert ranks_seen == world, (ranks_seen, world)
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  lib = _abi.load()

  lap('import_init')
  size = args.size
  warp = WARP if args.pair == 'warped' else None
  pre, post = synth_pair(size, args.seed + rank, warp=warp)
  lap('synth_pair')
  pre_t = torch.from_numpy(pre).to(dev)
  post_t = torch.from_numpy(post).to(dev)
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator(method=args.method)
  cfg = mesh.IntegrationConfig(
      dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(STEP, STEP),
      num_iters=args.mesh_iters, max_iters=args.mesh_iters, stop_v_max=0.005,
      dt_max=1000, start_cap=0.01, final_cap=10, prefer_orig_order=True)
  pad = PATCH // 2 // STEP

  def flow_step(a=None, b=None):
    return calc.flow_field(pre_t if a is None else a, post_t if b is None else b,
                           PATCH, STEP, batch_size=BATCH)

  def mesh_step(prev):
    x0 = torch.zeros(prev.shape, dtype=torch.float32, device=dev)
    return mesh.relax_mesh(x0, prev, cfg)

  def barrier():
    torch.cuda.synchronize(dev)
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize(dev)

  flow = None
  for _ in range(args.warmup):
    flow = flow_step()
    prev_t = torch.from_numpy(mesh_inputs(flow, pad)).to(dev)
    mesh_step(prev_t)
  if flow is None:
    flow = flow_step()
    prev_t = torch.from_numpy(mesh_inputs(flow, pad)).to(dev)

  lap('warmup')
  prof = _abi.SfmProfile()
  lib.sfm_profile_read(C.byref(prof))  # reset
  lib.sfm_profile_enable(1)
  barrier()
  t_flow = t_mesh = 0.0
  t0 = time.perf_counter()
  mesh_steps_done = 0
  for _ in range(args.steps):
    a = time.perf_counter()
    flow = flow_step()
    torch.cuda.synchronize(dev)
    b = time.perf_counter()
    _, _, t_iters = mesh_step(prev_t)
    torch.cuda.synchronize(dev)
    c = time.perf_counter()
    t_flow += b - a
    t_mesh += c - b
    mesh_steps_done += t_iters
  barrier()
  elapsed = time.perf_counter() - t0
  lib.sfm_profile_enable(0)
  _abi.check(lib.sfm_profile_read(C.byref(prof)))

  lap('timed_region')
  times = torch.tensor([elapsed, t_flow, t_mesh], dtype=torch.float64,
                       device=dev if backend == 'nccl' else 'cpu')
  if world > 1:
    dist.all_reduce(times, op=dist.ReduceOp.MAX)
  elapsed, t_flow, t_mesh = [float(v) for v in times.cpu()]

  n_grid = (size - (PATCH - STEP)) // STEP
  n_patches = n_grid * n_grid
  mesh_nodes = (n_grid + 2 * pad) ** 2
  pix = float(size) * size
  mpix_s = world * pix * args.steps / t_flow / 1e6
  node_updates_s = world * mesh_nodes * mesh_steps_done / t_mesh

  # Roofline of the dominant kernel: algorithmic work per launch = 2 P^4 ops
  # per patch (every pixel pair contributes to exactly one shift) x patches
  # per launch, over the HIP-event time of the launches in the timed region.
  flop_per_patch = 2.0 * PATCH ** 4
  uses_mfma = args.method != 1 and bool(
      getattr(flow_field, 'MFMA_I8_AVAILABLE', False))
  peak = PEAK_I8_TOPS if uses_mfma else PEAK_F32_TFLOPS

  def leg_figures(pf, n_pairs):
    """achieved / issued fractions of one measured leg (SfmProfile, pairs run)."""
    n = int(pf.launches[0])
    if not n:
      return None
    avg_ms = pf.kernel_ms[0] / n
    ppl = n_patches * n_pairs / n
    ach = flop_per_patch * ppl / (avg_ms * 1e-3) / 1e12
    o = {'avg_launch_ms': round(avg_ms, 4), 'launches': n,
         'patches_per_launch': round(ppl, 1), 'achieved': round(ach, 2),
         'frac': round(ach / peak, 4)}
    issued = int(pf.mfma_issued[0])
    if issued:
      # matrix instructions the kernel really issued (16x16x64 int8 = 32768
      # ops each): tile padding adds to the algorithmic count, pruning removes
      issued_ops = issued * 32768.0 / n
      o['issued_frac'] = round(issued_ops / (avg_ms * 1e-3) / 1e12 / peak, 4)
      o['issued_over_algorithmic'] = round(issued_ops / (flop_per_patch * ppl), 4)
    clk = float(pf.clock_mhz[0])
    if clk > 0:
      o['sustained_clock_mhz'] = round(clk, 1)
    drawn = int(pf.tiles_drawn[0])
    if drawn:
      o['row_tiles_skipped_frac'] = round(int(pf.tiles_skipped[0]) / drawn, 4)
      computed = drawn - int(pf.tiles_skipped[0])
      # (of the drawn row tiles) given up inside their row loop: proved cold by the
      # energy of the rows still to come, the rest of their matrix loop unissued
      o['row_tiles_abandoned_frac'] = round(int(pf.tiles_abandoned[0]) / drawn, 4)
      if computed:
        o['col_tiles_skipped_per_row_tile'] = round(
            int(pf.col_tiles_skipped[0]) / computed, 3)
    return o

  def timed_flow_leg(a_t, b_t, prune, n):
    """n flow passes with the pruning on / off: (SfmProfile, wall ms / pair)."""
    pf = _abi.SfmProfile()
    with _abi.option('SFM_MFMA_PRUNE', 1 if prune else 0):
      flow_step(a_t, b_t)                       # warm-up of this variant
      torch.cuda.synchronize(dev)
      lib.sfm_profile_read(C.byref(pf))
      lib.sfm_profile_enable(1)
      w0 = time.perf_counter()
      for _ in range(n):
        flow_step(a_t, b_t)
      torch.cuda.synchronize(dev)
      wall = (time.perf_counter() - w0) / n * 1e3
      lib.sfm_profile_enable(0)
      _abi.check(lib.sfm_profile_read(C.byref(pf)))
    return pf, wall

  roof = None
  timed = leg_figures(prof, args.steps)
  if timed:
    roof = {
        'kernel': 'xcorr_mfma_kernel<10,11,same-exact>' if uses_mfma
                  else 'corr_direct_kernel<f32>',
        'bound': 'mfma', 'peak': peak,
        # int8 multiply-accumulates counted as 2 ops each, against the dense
        # int8 MFMA peak (the contract's unit name for a matrix-core bound)
        'unit': 'TFLOP/s', 'op_dtype': 'int8' if uses_mfma else 'f32',
        'flop_per_patch': flop_per_patch, 'traffic': None,
        'pair': args.pair,
    }
    # `achieved` / `frac` describe the HARDWARE: the same kernel on the same pair
    # with the exact pruning switched off, i.e. every algorithmic operation
    # really executed.  The production (pruned) launch of the timed region is
    # `frac_pruned`: algorithmic operations over its time, an algorithmic gain.
    roof['pruned'] = timed
    roof['frac_pruned'] = timed['frac']
    # the reference's FFT formulation needs ~3 transforms x 2.5 N log2 N
    # flop per patch (N = 320^2): ~100x less arithmetic than the direct sum.  The
    # same patch rate priced at that count, so the direct form is not flattered
    # (the FFT form itself is timed in aux_rooflines.xcorr_fft_2d).
    n_fft = float((2 * PATCH) ** 2)
    fft_flop = 3 * 2.5 * n_fft * math.log2(n_fft)
    roof['fft_equivalent'] = {
        'flop_per_patch': round(fft_flop),
        'tflops_at_this_patch_rate': round(
            fft_flop * world * n_patches * args.steps / t_flow / 1e12, 3),
        'direct_over_fft_flop': round(flop_per_patch / fft_flop, 1)}
    legs = uses_mfma and not args.no_legs and world == 1
    if legs:
      n_leg = max(args.steps, 3)
      pf_full, wall_full = timed_flow_leg(pre_t, post_t, False, n_leg)
      full = leg_figures(pf_full, n_leg)
      full['flow_ms_per_pair'] = round(wall_full, 3)
      roof['unpruned'] = full
      for k in ('achieved', 'frac', 'avg_launch_ms', 'launches', 'patches_per_launch',
                'issued_frac'):
        if k in full:
          roof[k] = full[k]
      roof['frac_is'] = 'un-pruned leg of the same run (every tile computed)'
      # the other synthetic pair, both ways
      other = 'exact' if args.pair == 'warped' else 'warped'
      o_pre, o_post = synth_pair(size, args.seed + rank,
                                 warp=WARP if other == 'warped' else None)
      o_pre_t = torch.from_numpy(o_pre).to(dev)
      o_post_t = torch.from_numpy(o_post).to(dev)
      pf_p, wall_p = timed_flow_leg(o_pre_t, o_post_t, True, n_leg)
      pf_f, wall_f = timed_flow_leg(o_pre_t, o_post_t, False, n_leg)
      op, of = leg_figures(pf_p, n_leg), leg_figures(pf_f, n_leg)
      op['flow_ms_per_pair'] = round(wall_p, 3)
      of['flow_ms_per_pair'] = round(wall_f, 3)
      roof['other_pair'] = {'pair': other, 'pruned': op, 'unpruned': of}
      del o_pre_t, o_post_t
    else:
      # no un-pruned leg in this run: the line can only state the algorithmic figure
      for k in ('achieved', 'frac', 'avg_launch_ms', 'launches', 'patches_per_launch',
                'issued_frac'):
        if k in timed:
          roof[k] = timed[k]
      roof['frac_is'] = ('pruned launch of the timed region (algorithmic; run without '
                         '--no-legs at --gpus 1 for the un-pruned hardware leg)')

  ms_ms, ms_n = prof.kernel_ms[1], prof.launches[1]
  mesh_obj = {
      'value': node_updates_s, 'unit': 'node-updates/s',
      'nodes': mesh_nodes, 'iterations_per_step': mesh_steps_done // max(args.steps, 1),
      'ms_per_step': t_mesh / args.steps * 1e3,
  }
  if ms_n:
    # Timed launches are either one persistent kernel per chunk (num_iters
    # steps each) or one integrate kernel per step.
    steps_per_launch = mesh_steps_done / ms_n
    us_per_step = ms_ms * 1e3 / mesh_steps_done
    gbs = mesh_nodes * 56.0 / (us_per_step * 1e-6) / 1e9
    persistent = steps_per_launch > 1.5
    mesh_obj['roofline'] = {
        'kernel': 'mesh_persist2d_spec_kernel<16>' if persistent else 'integrate_kernel<2>',
        'bound': 'hbm', 'achieved': round(gbs, 2), 'peak': PEAK_HBM_GBS,
        'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 5),
        'kernel_us_per_step': round(us_per_step, 3), 'launches': int(ms_n),
        'steps_per_launch': round(steps_per_launch, 1),
        'bytes_per_node_update': 56, 'traffic': None,
        'note': 'the 1.3 MB state never leaves the chip (registers + LDS in the '
                'persistent kernel); the step is bound by the per-step '
                'inter-workgroup exchange latency, not by HBM: see steps_per_s',
    }
    mesh_obj['steps_per_s'] = mesh_steps_done / t_mesh

  # HBM traffic of the dominant kernel: PMC counters cannot be read from inside
  # the run, so the figure comes from the separate rocprofv3 --pmc passes of THIS
  # bench (tools/measure/profile_round6.sh -> profiles/r06_pmc_traffic.json), and
  # only when that file was measured on the library that is loaded now
  # (.build_sha); otherwise null.
  if roof and uses_mfma and size == 8192:
    try:
      pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r06_pmc_traffic.json')))
      meta = pmc.get('_meta', {})
      sha = build_sha()
      if sha and meta.get('git_sha') == sha and meta.get('pair', 'exact') == args.pair:
        for name, v in pmc.items():
          if 'xcorr_mfma_kernel<10, 11,' in name:
            meta_ppl = float(meta.get('patches_per_launch', n_patches))
            ppl = roof['pruned']['patches_per_launch']
            roof['traffic'] = int(v['hbm_bytes_per_launch'] / meta_ppl * ppl)
          if 'mesh_persist2d' in name and mesh_obj.get('roofline'):
            # HBM-side bytes of ONE launch of the persistent kernel (all its steps)
            mesh_obj['roofline']['traffic'] = int(v['hbm_bytes_per_launch'])
            mesh_obj['roofline']['traffic_note'] = (
                'per launch of %d FIRE steps (fabric traffic of the inter-workgroup exchange: '
                'the state itself stays on chip); counters: profiles/r06_pmc_traffic.json'
                % args.mesh_iters)
        if roof.get('traffic') is not None:
            roof['traffic_source'] = {
                'file': 'profiles/r06_pmc_traffic.json', 'measured_on_git_sha': sha,
                'launch': 'pruned (production) launch',
                'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of '
                          'this bench; FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, per '
                          'launch of %d patches' % round(ppl)}
      else:
        roof['traffic_note'] = ('null: profiles/r06_pmc_traffic.json was measured on %s / '
                                'pair %s, this library is %s / pair %s'
                                % (meta.get('git_sha'), meta.get('pair'), sha, args.pair))
    except (OSError, ValueError):
      pass

  # Steady state: the timed region above is < 1 s on a power-limited kernel, so
  # the same step is repeated back to back for >= --sustain seconds (all ranks).
  lap('roofline_legs')
  sustained = None
  if args.sustain > 0:
    barrier()
    s0 = time.perf_counter()
    n_sus = 0
    while True:
      flow_step()
      mesh_step(prev_t)
      n_sus += 1
      if n_sus % 8 == 0 or n_sus < 8:
        torch.cuda.synchronize(dev)
        # every rank leaves after the same count: rank 0's clock decides
        flag = torch.tensor([1.0 if time.perf_counter() - s0 >= args.sustain else 0.0],
                            device=dev if backend == 'nccl' else 'cpu')
        if world > 1:
          dist.broadcast(flag, src=0)
        if flag.item() > 0:
          break
    barrier()
    s_el = time.perf_counter() - s0
    sustained = {'seconds': round(s_el, 2), 'steps': n_sus,
                 'ms_per_step': round(s_el / n_sus * 1e3, 3),
                 'mpix_s': round(world * pix * n_sus / s_el / 1e6, 1),
                 'note': 'flow + mesh steps back to back (includes the mesh leg)'}

  lap('sustain')
  aux = None
  if world == 1 and not args.no_legs:
    aux = aux_legs(dev, args.seed)
    lap('aux_legs')

  cfg_legs = None
  if world == 1 and not args.no_legs:
    cfg_legs = config_legs(dev, pre, args.seed)
    lap('config_legs')

  sharded = None
  if args.mesh_sharded > 0:
    sharded = mesh_sharded_leg(args.mesh_sharded, dev, rank, world)

  out = {
      'metric': 'patch-xcorr Mpix/s (+ mesh node-updates/s) on 8192^2 tiles',
      'value': mpix_s, 'unit': 'Mpix/s', 'n_gpus': world,
      'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': elapsed / args.steps * 1e3,
      'flow_ms_per_step': t_flow / args.steps * 1e3,
      'mesh_ms_per_step': t_mesh / args.steps * 1e3,
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'u8' if uses_mfma else 'f32', 'data': 'synthetic',
      'config': {
          'workload': f'single {size}x{size} EM tile pair per GPU '
                      + ('(smooth 6 px / 2048 px deformation + shift + noise, SURVEY 8d '
                         'realistic pair)' if args.pair == 'warped' else
                         '(rigid integer shift + noise)')
                      + f', patch=160 step=40 batch=1024 flow + {args.mesh_iters}-iter FIRE '
                      f'mesh relax [2,1,{n_grid + 2 * pad},{n_grid + 2 * pad}] '
                      '(BASELINE configs[1])',
          'pair': args.pair,
          'patches_per_pair': n_patches, 'xcorr_method':
              'int8 MFMA' if uses_mfma else 'direct f32',
      },
      'patches_per_s': world * n_patches * args.steps / t_flow,
      'mesh': mesh_obj, 'roofline': roof,
      'build_sha': build_sha(),
      'wall_s': wall,
  }
  if sustained:
    out['sustained'] = sustained
    out['sustained_ms_per_step'] = sustained['ms_per_step']
  if sharded:
    out['mesh_sharded'] = sharded
  if aux:
    out['aux_rooflines'] = aux
  if cfg_legs:
    out['configs_legs'] = cfg_legs

  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    out['cpu_baseline'] = cpu_baseline(size, args.seed + rank, args.pair)
    lap('cpu_baseline')

  # N > 1: the legs that really communicate (the timed default workload above
  # is N independent tile pairs, no data-path collective).  They run LAST and
  # under a watchdog: the headline numbers above are complete, and a leg that
  # fails or hangs (they are the least-exercised code: single-GPU boxes cannot
  # run them over RCCL) costs its own entry, never the line.
  if (world > 1 or forced) and not args.no_multi_gpu_legs:
    import threading
    multi = {
        'backend': 'rccl' if backend == 'nccl' else backend,
        # the value of an all-reduce of ones over the process group
        'rccl_ranks' if backend == 'nccl' else 'ranks_seen': ranks_seen,
        'devices': 'one GPU per rank' if not os.environ.get('SFM_BENCH_ONE_DEVICE')
                   else 'all ranks share cuda:0 (smoke test)',
    }
    out['multi_gpu'] = multi
    state = {'leg': None}

    def bail():
      multi[state['leg'] or 'legs'] = {
          'error': f'no result within {args.multi_gpu_timeout:.0f} s (leg abandoned)'}
      if rank == 0:
        print(json.dumps(out), flush=True)
      os._exit(0)

    for name, fn in (
        ('section_chain', lambda: section_chain_leg(dev, rank, world, backend, args.seed,
                                                    args.mesh_iters, pre_t, post_t)),
        # (configs[4] is 16 z chunks: two per rank on an 8-GPU node, one per rank below that)
        ('volumetric_chunks', lambda: volumetric_leg(dev, rank, world, backend, args.seed,
                                                     chunks_per_rank=2 if world >= 8 else 1)),
        ('mesh_sharded', lambda: mesh_sharded_leg(
            2 if forced else 1, dev, rank, world,
            iters=min(200, max(args.mesh_iters, 20)), loopback=forced))):
      state['leg'] = name
      dog = threading.Timer(args.multi_gpu_timeout, bail)
      dog.daemon = True
      dog.start()
      try:
        multi[name] = fn()
      except Exception as e:   # pylint: disable=broad-except
        multi[name] = {'error': f'{type(e).__name__}: {e}'[:400]}
      finally:
        dog.cancel()
      lap('multi_gpu.' + name)
      # a rank that failed must not leave the others inside a collective of the
      # NEXT leg: agree on going on (an all-reduce with its own watchdog)
      dog = threading.Timer(args.multi_gpu_timeout, bail)
      dog.daemon = True
      dog.start()
      flag = torch.tensor([0.0 if 'error' in multi[name] else 1.0],
                          device=dev if backend == 'nccl' else 'cpu')
      dist.all_reduce(flag, op=dist.ReduceOp.MIN)
      dog.cancel()
      if flag.item() < 1.0:
        multi[name].setdefault('error', 'failed on another rank')
        break
  wall['total'] = round(sum(wall.values()), 3)
  if rank == 0:
    print(json.dumps(out), flush=True)
  if world > 1 or forced:
    try:
      dist.destroy_process_group()
    except Exception:   # pylint: disable=broad-except
      pass


AUX_TRAFFIC_FILE = 'r06_pmc_traffic_aux.json'


def aux_legs(dev, seed, only=None):
  """HBM roofline lines of the other kernels of the path, wall-clock timed with
  the inputs resident (a few hundred ms each): the FFT form of the correlation
  (float images: flow_field.py:374-441 with method 3; 3-D patches) and the
  volumetric / large in-plane mesh steps (mesh.py:192-279, 383-513).
  `achieved` = ALGORITHMIC bytes / time: each input patch pixel read once and
  the peak statistics written (FFT form); x, v, a read and written and prev
  read once per node update (mesh: 14 floats in-plane, 21 volumetric).
  `traffic` = HBM bytes of ONE call of the leg (all its launches) from separate
  rocprofv3 --pmc passes of `bench.py --aux-leg LEG` (tools/measure/pmc_aux.sh ->
  profiles/r06_pmc_traffic_aux.json), taken only when that file was measured on
  the library that is loaded now; `only`: run that one leg (the counted process)."""
  import torch
  from sofima_amd import _abi, flow_field, mesh
  out = []
  measured = {}
  try:
    pmc = json.load(open(os.path.join(ROOT, 'profiles', AUX_TRAFFIC_FILE)))
    if pmc.get('_meta', {}).get('git_sha') == build_sha():
      measured = pmc
  except (OSError, ValueError):
    pass

  def timed(fn, reps):
    fn()
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    for _ in range(reps):
      fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t) / reps

  def line(name, kernel, nbytes, sec, calls, **extra):
    gbs = nbytes / sec / 1e9
    o = {'leg': name, 'kernel': kernel, 'bound': 'hbm', 'achieved': round(gbs, 1),
         'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 4),
         'ms': round(sec * 1e3, 3), 'traffic': None, 'calls_in_counted_process': calls}
    m = measured.get(name)
    if m:
      o['traffic'] = int(m['hbm_bytes_per_call'])
      o['traffic_over_algorithmic'] = round(m['hbm_bytes_per_call'] / nbytes, 2)
      o['traffic_gbs'] = round(m['hbm_bytes_per_call'] / sec / 1e9, 1)
      o['traffic_source'] = 'profiles/' + AUX_TRAFFIC_FILE
    o.update(extra)
    out.append(o)

  rng = np.random.default_rng(seed)
  if only in (None, 'xcorr_fft_2d'):
    # FFT form, in-plane: float32 2048^2 pair, patch 160 step 40 -> 48 x 48 patches
    n2 = 2048
    a = torch.from_numpy(rng.random((n2, n2), dtype=np.float32)).to(dev)
    b = torch.roll(a, (2, -3), (0, 1)).contiguous()
    calc = flow_field.JAXMaskedXCorrWithStatsCalculator(method=_abi.XCORR_FFT)
    sec = timed(lambda: calc.flow_field(a, b, PATCH, STEP, batch_size=BATCH), 3)
    np2 = ((n2 - PATCH) // STEP + 1) ** 2
    line('xcorr_fft_2d', 'fft_pencil / fft_xfwd / fft_xinv (sfm_fft_own.hip)',
         np2 * (2 * PATCH * PATCH * 4 + 16), sec, 4, patches=np2,
         workload='float32 2048^2 pair, patch 160 step 40, FFT form')
  if only in (None, 'xcorr_fft_3d'):
    # FFT form, volumetric: 160^3 pair, patch 80 step 40 -> 3^3 patches
    v = torch.from_numpy(rng.random((160, 160, 160), dtype=np.float32)).to(dev)
    w = torch.roll(v, (1, -2, 2), (0, 1, 2)).contiguous()
    calc3 = flow_field.JAXMaskedXCorrWithStatsCalculator()
    sec = timed(lambda: calc3.flow_field(v, w, (80, 80, 80), (40, 40, 40), batch_size=64), 3)
    line('xcorr_fft_3d', 'fft3 kernels (sfm_fft_own.hip)', 27 * (2 * 80 ** 3 * 4 + 20), sec, 4,
         patches=27, workload='float32 160^3 pair, patch 80^3 step 40 (BASELINE configs[4] patch size)')
  if only in (None, 'xcorr_search_window'):
    # search-window call of EstimateMissingFlow (processor/flow.py:577,792-803): pre patches
    # of 160 + 2 x 40 against 160 post patches on the matrix cores (kModeGeneral, wide
    # variant) -- an MFMA-bound line; the FFT form of the same call is timed beside it
    n2 = 4096
    from tests.util import em_texture
    base = em_texture(rng, (n2 + 16, n2 + 16))
    a8 = torch.from_numpy(np.ascontiguousarray(base[8:8 + n2, 8:8 + n2])).to(dev)
    b8 = torch.from_numpy(np.ascontiguousarray(base[10:10 + n2, 5:5 + n2])).to(dev)
    pw = PATCH + 2 * 40
    calc_m = flow_field.JAXMaskedXCorrWithStatsCalculator()
    calc_f = flow_field.JAXMaskedXCorrWithStatsCalculator(method=_abi.XCORR_FFT)
    sec = timed(lambda: calc_m.flow_field(a8, b8, pw, STEP, batch_size=BATCH,
                                          post_patch_size=PATCH), 3)
    sec_f = timed(lambda: calc_f.flow_field(a8, b8, pw, STEP, batch_size=BATCH,
                                            post_patch_size=PATCH), 2)
    npw = ((n2 - PATCH) // STEP + 1) ** 2
    ops = 2.0 * pw * pw * PATCH * PATCH * npw
    out.append({'leg': 'xcorr_search_window', 'kernel': 'xcorr_mfma_kernel<15,11,general> (+ mfma_prep_wide_kernel)',
                'bound': 'mfma', 'achieved': round(ops / sec / 1e12, 1), 'peak': PEAK_I8_TOPS,
                'unit': 'TFLOP/s', 'frac': round(ops / sec / 1e12 / PEAK_I8_TOPS, 4),
                'ms': round(sec * 1e3, 3), 'traffic': None, 'calls_in_counted_process': 4,
                'patches': npw, 'us_per_patch': round(sec / npw * 1e6, 3),
                'fft_form_us_per_patch': round(sec_f / npw * 1e6, 3),
                'speedup_over_fft_form': round(sec_f / sec, 2),
                'workload': 'uint8 4096^2 pair, pre patch %d post patch %d step 40 (whole flow_field() '
                            'call: prep + correlation + peaks; no pruning in this mode)' % (pw, PATCH)})
  # mesh steps: FIRE, 200 iterations of one chunk
  iters = 200
  for name, shape, force, stride, fl in (
      ('mesh_3d', (3, 4, 100, 100, 100), mesh.elastic_mesh_3d, (40, 40, 40), 21),
      ('mesh_2d_large', (2, 4, 2048, 2048), None, (40, 40), 14),
      ('mesh_2d_montage_size', (2, 64, 204, 204), None, (40, 40), 14)):
    if only not in (None, name):
      continue
    prev = torch.from_numpy((rng.standard_normal(shape) * 3).astype(np.float32)).to(dev)
    x0 = torch.zeros_like(prev)
    cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=stride,
                                 num_iters=iters, max_iters=iters, stop_v_max=1e-9,
                                 dt_max=1000, start_cap=0.1, final_cap=10)
    kw = {'mesh_force': force} if force is not None else {}
    sec = timed(lambda: mesh.relax_mesh(x0, prev, cfg, **kw), 2)
    nodes = int(np.prod(shape[1:]))
    line(name, 'integrate_march3d_kernel<512> + advance_kernel<3>' if force is not None
         else 'integrate_shared2d_kernel',
         nodes * iters * fl * 4, sec, 3, nodes=nodes, us_per_step=round(sec / iters * 1e6, 2),
         node_updates_per_s=round(nodes * iters / sec, 0), state=list(shape),
         bytes_per_node_update=fl * 4)
    del prev, x0
  return out


def config_legs(dev, canvas, seed):
  """Driver-timed legs of BASELINE configs[2] and configs[4] at their workload
  shapes (bounded samples; wall clock with the inputs resident or uploaded as
  the drop-in functions do it):
    configs[2]  8 x 8 montage of 4096^2 tiles: stitch_elastic.compute_flow_map
                (patch 120, step 20, batch 256) on the 4 strip pairs of a 2 x 2
                grid of 4096^2 tiles cut from the headline canvas, and 200 FIRE
                steps of the [2, 64, 204, 204] montage mesh with the native
                target-mesh prev_fn and drift removal;
    configs[4]  volumetric: flow_field with 80^3 patches, step 40 on the overlap
                strip of two 512^3 tiles (512 x 512 x 120 voxels) and 200 FIRE
                steps of the [3, 64, 12, 12, 12] montage mesh (elastic_mesh_3d,
                volumetric target mesh, per-column drift removal)."""
  import torch
  from scipy import ndimage
  from sofima_amd import flow_field, mesh, stitch_elastic
  from tests.util import synth_montage
  out = []

  def timed(fn, reps):
    fn()
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    for _ in range(reps):
      r = fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t) / reps, r

  # -- configs[2]: flow leg ------------------------------------------------------
  T, OV, G = 4096, 400, 3
  if canvas.shape[0] >= 2 * T - OV:
    # a G x G grid of tiles from the headline canvas continued by reflection
    # (continuous content in every overlap; 12 strip pairs keep the loop's
    # run-ahead of MAX_PAIRS_IN_FLIGHT busy like a real montage does)
    side = G * T - (G - 1) * OV
    big = np.pad(canvas, ((0, max(0, side - canvas.shape[0])), (0, max(0, side - canvas.shape[1]))),
                 mode='reflect')
    tiles = {(x, y): np.ascontiguousarray(big[y * (T - OV):y * (T - OV) + T,
                                              x * (T - OV):x * (T - OV) + T])
             for y in range(G) for x in range(G)}
    del big
    cx = np.zeros((2, G, G)); cx[0] = -OV          # coarse offsets of the (x+1, y) tiles
    cy = np.zeros((2, G, G)); cy[1] = -OV
    sec, (fl, _) = timed(lambda: stitch_elastic.compute_flow_map(tiles, cx, 0), 2)
    sec_y, (fl_y, _) = timed(lambda: stitch_elastic.compute_flow_map(tiles, cy, 1), 2)
    n_pairs = len(fl) + len(fl_y)
    per_pair = (sec + sec_y) / n_pairs
    shp = next(iter(fl.values())).shape
    patches = int((shp[1] - 5) * (shp[2] - 5))
    out.append({
        'config': 'configs[2] flow', 'workload':
            f'compute_flow_map on {n_pairs} overlap strips ({T} x {OV}) of a {G} x {G} grid of '
            f'{T}^2 tiles, patch 120 step 20 batch 256 (host tiles: uploads included)',
        'ms_per_strip_pair': round(per_pair * 1e3, 3), 'patches_per_pair': patches,
        'mpix_s': round(T * OV / per_pair / 1e6, 1),
        'montage_8x8_estimate_ms': round(per_pair * 112 * 1e3, 1),
        'tops_algorithmic': round(2.0 * 120 ** 4 * patches / per_pair / 1e12, 1)})
  # -- configs[2]: mesh leg ------------------------------------------------------
  rng = np.random.default_rng(seed + 2)
  iters = 200
  for name, ms_shape, ov, stride, force, label in (
      ('configs[2] mesh', (204, 204), 20, (20.0, 20.0), None, '[2, 64, 204, 204]'),
      ('configs[4] mesh', (12, 12, 12), 3, (40.0, 40.0, 40.0), mesh.elastic_mesh_3d,
       '[3, 64, 12, 12, 12]')):
    nb, fx, fy, x0 = synth_montage(rng, 8, 8, ms_shape, ov, amp=4.0)
    cfg = mesh.IntegrationConfig(
        dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=stride, num_iters=iters, max_iters=iters,
        stop_v_max=1e-9, dt_max=100, prefer_orig_order=force is None, start_cap=0.1,
        final_cap=10.0, remove_drift=True)
    fn = stitch_elastic.TargetMeshFn(nb, fx, fy, stride)
    x_t = torch.from_numpy(x0).to(dev)
    kw = {'mesh_force': force} if force is not None else {}
    sec, _ = timed(lambda: mesh.relax_mesh(x_t, None, cfg, prev_fn=fn, **kw), 2)
    nodes = int(np.prod(x0.shape[1:]))
    out.append({
        'config': name, 'workload':
            f'{label} montage mesh (8 x 8 tiles), native target-mesh prev_fn, remove_drift, '
            f'{iters} FIRE steps',
        'us_per_step': round(sec / iters * 1e6, 2),
        'node_updates_per_s': round(nodes * iters / sec, 0), 'nodes': nodes})
  # -- configs[4]: flow leg ------------------------------------------------------
  vol = ndimage.gaussian_filter(
      rng.standard_normal((512 + 8, 512 + 8, 120 + 8), dtype=np.float32), 1.5)
  vol = ((vol - vol.min()) / (vol.max() - vol.min()) * 255).astype(np.uint8)
  a = torch.from_numpy(np.ascontiguousarray(vol[4:516, 4:516, 4:124])).to(dev)
  b = torch.from_numpy(np.ascontiguousarray(vol[6:518, 1:513, 7:127])).to(dev)
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  sec, f = timed(lambda: calc.flow_field(a, b, (80, 80, 80), 40, batch_size=64), 2)
  patches = int(np.prod(f.shape[1:]))
  out.append({
      'config': 'configs[4] flow', 'workload':
          '512 x 512 x 120 overlap strip of two 512^3 tiles, 80^3 patches step 40 batch 64 '
          '(FFT form, resident volumes)',
      'ms_per_strip_pair': round(sec * 1e3, 3), 'patches_per_pair': patches,
      'ms_per_patch': round(sec / patches * 1e3, 4),
      'mvox_s': round(512 * 512 * 120 / sec / 1e6, 1)})

  # -- masked correlation (mask_only_for_patch_selection=False, the reference's
  # default, flow_field.py:91-156) on the headline geometry: 5 % of both images
  # masked in discs of radius 130 px.  A patch takes 1 exact int8 product pass
  # if it holds no masked pixel, 4 if one side does, 8 if both do; the roofline
  # figure prices those passes (2 P^4 op each) against the dense int8 peak.
  if canvas.shape[0] >= 8192:
    from sofima_amd import flow_field as ff
    side = 8192
    pre8 = np.ascontiguousarray(canvas[:side, :side])
    post8 = np.ascontiguousarray(np.roll(pre8, (3, -5), (0, 1)))
    mrng = np.random.default_rng(seed + 7)
    yy, xx = np.mgrid[-130:131, -130:131]
    disc = yy ** 2 + xx ** 2 <= 130 ** 2
    masks = []
    for _ in range(2):
      m = np.zeros((side, side), bool)
      for _ in range(int(0.05 * side * side / (np.pi * 130 ** 2))):
        y, x = mrng.integers(130, side - 131, 2)
        m[y - 130:y + 131, x - 130:x + 131] |= disc
      masks.append(m)
    ca = ff._masked_counts(masks[0], (PATCH, PATCH), (STEP, STEP)) > 0
    cb = ff._masked_counts(masks[1], (PATCH, PATCH), (STEP, STEP)) > 0
    passes = np.where(ca & cb, 8, np.where(ca | cb, 4, 1))
    ta, tb = (torch.from_numpy(v).to(dev) for v in (pre8, post8))
    ma, mb = (torch.from_numpy(v).to(dev) for v in masks)
    calc2 = flow_field.JAXMaskedXCorrWithStatsCalculator()
    sec, f = timed(lambda: calc2.flow_field(ta, tb, PATCH, STEP, pre_mask=ma, post_mask=mb,
                                            batch_size=BATCH, max_masked=1.01,
                                            mask_only_for_patch_selection=False), 2)
    top = float(passes.sum()) * 2.0 * PATCH ** 4 / 1e12
    out.append({
        'config': 'configs[1] masked flow', 'workload':
            '8192^2 pair, masks on both images (5 % of the area in discs of radius 130), '
            'mask_only_for_patch_selection=False, patch 160 step 40 batch 1024',
        'ms_per_pair': round(sec * 1e3, 2), 'patches_per_pair': int(passes.size),
        'patches_with_masked_pixels': round(float((ca | cb).mean()), 4),
        'matrix_passes_per_patch': round(float(passes.mean()), 3),
        'mpix_s': round(side * side / sec / 1e6, 1),
        'roofline': {'bound': 'mfma', 'op_dtype': 'int8', 'peak': PEAK_I8_TOPS, 'unit': 'TFLOP/s',
                     'achieved': round(top / sec, 1), 'frac': round(top / sec / PEAK_I8_TOPS, 4),
                     'note': 'whole path (prep, tables, product passes, Padfield assembly, peak '
                             'search) against the matrix passes the patches need'}})
  return out


def section_chain_leg(dev, rank, world, backend, seed, iters, pre_t=None, post_t=None,
                      sections_per_rank=8):
  """BASELINE configs[3] end to end: a z-stack of 8 x world sections of 8192^2,
  one block of 8 consecutive sections per rank (em_alignment notebook cells 11-26,
  38-48).  Per section on its rank: flow_field against the previous section
  (patch 160, step 40, batch 1024; the field stays in HBM) -> clean_flow ->
  [2, 1, 205, 205] mesh target -> compose_maps_fast with the previous solved
  section -> relax_mesh (FIRE, `iters` iterations).  Then the last solved mesh
  of every block (336 KB) crosses ranks in ONE all-gather (RCCL, GPU to GPU, on
  an nccl group) and the small cross-block relaxation runs on every rank.
  Sections are synthetic: the headline pair's second image rolled by a
  section-dependent integer shift against the first (every section pair has the
  statistics of the headline pair; nine resident images per rank).  Times are
  the max over ranks; the hand-off includes waiting for the slowest rank,
  `handoff_only_us` is the collective alone."""
  import torch
  import torch.distributed as dist
  from sofima_amd import dist as sdist, flow_field, flow_utils, mesh
  with_flow = pre_t is not None
  size = int(pre_t.shape[0]) if with_flow else 8192
  n_grid, pad = (size - (PATCH - STEP)) // STEP, PATCH // 2 // STEP
  n = sections_per_rank * world
  first = rank * sections_per_rank
  cfg = mesh.IntegrationConfig(
      dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(STEP, STEP), num_iters=iters,
      max_iters=iters, stop_v_max=0.005, dt_max=1000, start_cap=0.01, final_cap=10,
      prefer_orig_order=True)
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  full = n_grid + 2 * pad

  def section_images():
    """The rank's 9 sections: image z = post rolled by (z, -z) pixels for odd z,
    pre rolled likewise for even z (adjacent sections differ like pre / post)."""
    imgs = []
    for z in range(first, first + sections_per_rank + 1):
      base = post_t if z & 1 else pre_t
      imgs.append(torch.roll(base, shifts=(z % 5, -(z % 3)), dims=(0, 1)).contiguous())
    return imgs

  def block_flows(imgs):
    """Cleaned, NaN-padded flow of the rank's block: [2, n, 205, 205] with the
    other ranks' sections left NaN (align_sections_blocked only reads its own)."""
    flow = np.full((2, n, full, full), np.nan, np.float32)
    for i in range(sections_per_rank):
      f = calc.flow_field(imgs[i], imgs[i + 1], PATCH, STEP, batch_size=BATCH,
                          device_output=True)
      clean = flow_utils.clean_flow(f.tensor[:, None], min_peak_ratio=1.4,
                                    min_peak_sharpness=1.4, max_magnitude=80,
                                    max_deviation=20)
      flow[:, first + i, pad:-pad, pad:-pad] = np.asarray(clean)[:, 0]
    return flow

  def synthetic_flows():
    from scipy import ndimage
    flow = np.full((2, n, full, full), np.nan, np.float32)
    drift = np.zeros((2, n_grid, n_grid), np.float32)
    for z in range(n):
      rng = np.random.default_rng(seed + 2 + z)          # seeds 1004 + z
      drift += ndimage.gaussian_filter(rng.standard_normal(drift.shape), (0, 30, 30)) * 8
      local = ndimage.gaussian_filter(rng.standard_normal(drift.shape), (0, 12, 12)) * 25
      flow[:, z, pad:-pad, pad:-pad] = 0.25 * drift + local
    return flow

  def barrier():
    torch.cuda.synchronize(dev)
    dist.barrier()

  imgs = section_images() if with_flow else None

  def run(tm):
    t0 = time.perf_counter()
    flow = block_flows(imgs) if with_flow else synthetic_flows()
    torch.cuda.synchronize(dev)
    t_flow = time.perf_counter() - t0
    blocks, last, xblk = sdist.align_sections_blocked(flow, cfg, float(STEP), n_blocks=world,
                                                      timing=tm)
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0, t_flow, last, xblk

  run({})                                              # warm-up
  reps = 2
  tot = np.zeros(5)
  for _ in range(reps):
    tm = {}
    barrier()
    total, t_flow, last, xblk = run(tm)
    tot += [total, t_flow if with_flow else 0.0, tm['solve_s'], tm['handoff_s'], tm['xblk_s']]
  tot /= reps
  # the collective alone, data ready on every rank
  mine = [torch.from_numpy(last[:, b:b + 1].copy()).to(dev)
          for b in sdist.shard_units(world, rank, world)]
  sdist.gather_boundaries(mine, world)
  barrier()
  t0 = time.perf_counter()
  for _ in range(20):
    sdist.gather_boundaries(mine, world)
  torch.cuda.synchronize(dev)
  only = (time.perf_counter() - t0) / 20
  t = torch.tensor(list(tot) + [only], dtype=torch.float64,
                   device=dev if backend == 'nccl' else 'cpu')
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  total, t_flow, solve, hand, xb, only = [float(v) for v in t.cpu()]
  nodes = full ** 2
  out = {
      'workload': f'{n} sections of {size}^2, {sections_per_rank} per rank: '
                  + ('flow_field (patch 160 step 40) + clean_flow + ' if with_flow else
                     '(synthetic flow fields) ')
                  + f'compose_maps_fast + {iters}-iteration FIRE relaxation of the '
                  f'[2,1,{full},{full}] mesh per section, blocks chained by the last-section mesh '
                  '(BASELINE configs[3])',
      'sections': n, 'blocks': world, 'ms_total': round(total * 1e3, 3),
      'ms_flow': round(t_flow * 1e3, 3),
      'ms_block_solve': round(solve * 1e3, 3), 'ms_handoff': round(hand * 1e3, 3),
      'ms_cross_block': round(xb * 1e3, 3), 'handoff_only_us': round(only * 1e6, 1),
      'handoff_bytes_per_block': int(2 * nodes * 4),
      'handoff': 'one all-gather of the blocks\' last meshes '
                 + ('(RCCL, device tensors)' if backend == 'nccl' else f'({backend}, host tensors)'),
      'sections_per_s': n / total,
      'node_updates_per_s': (n + world) * nodes * iters / total,
      'finite_fraction': float(np.isfinite(xblk).mean()),
  }
  if with_flow:
    out['mpix_s'] = round(n * float(size) * size / total / 1e6, 1)
  return out


def volumetric_leg(dev, rank, world, backend, seed, chunks_per_rank=1):
  """BASELINE configs[4] across the ranks: the 2 x 2 x 16 grid of 512^3 tiles is
  16 z chunks of one 2 x 2 montage each, and the chunks are independent units
  (no data-path collective: `dist.map_units` semantics, weak scaling).  Per
  chunk on its rank: the four tile pairs' volumetric flow (80^3 patches, step
  40, on a 512 x 512 x 120 overlap strip; FFT form) and 200 FIRE steps of the
  chunk's [3, 4, 12, 12, 12] montage mesh (elastic_mesh_3d, volumetric target
  mesh, per-column drift removal); the small flow fields are gathered to every
  rank at the end (one all-gather).  Times are the max over ranks."""
  import torch
  import torch.distributed as dist
  from scipy import ndimage
  from sofima_amd import flow_field, mesh, stitch_elastic
  from tests.util import synth_montage
  rng = np.random.default_rng(seed + 40 + rank)
  vol = ndimage.gaussian_filter(
      rng.standard_normal((512 + 8, 512 + 8, 120 + 8), dtype=np.float32), 1.5)
  vol = ((vol - vol.min()) / (vol.max() - vol.min()) * 255).astype(np.uint8)
  a = torch.from_numpy(np.ascontiguousarray(vol[4:516, 4:516, 4:124])).to(dev)
  b = torch.from_numpy(np.ascontiguousarray(vol[6:518, 1:513, 7:127])).to(dev)
  del vol
  calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
  iters = 200
  nb, fx, fy, x0 = synth_montage(rng, 2, 2, (12, 12, 12), 3, amp=4.0)
  stride = (40.0, 40.0, 40.0)
  cfg = mesh.IntegrationConfig(
      dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=stride, num_iters=iters, max_iters=iters,
      stop_v_max=1e-9, dt_max=100, start_cap=0.1, final_cap=10.0, remove_drift=True)
  fn = stitch_elastic.TargetMeshFn(nb, fx, fy, stride)
  x_t = torch.from_numpy(x0).to(dev)

  def chunk():
    flows = [calc.flow_field(a, b, (80, 80, 80), 40, batch_size=64, device_output=True)
             for _ in range(4)]
    mesh.relax_mesh(x_t, None, cfg, mesh_force=mesh.elastic_mesh_3d, prev_fn=fn)
    return torch.stack([f.tensor for f in flows])

  chunk()
  torch.cuda.synchronize(dev)
  dist.barrier()
  t0 = time.perf_counter()
  mine = [chunk() for _ in range(chunks_per_rank)]
  torch.cuda.synchronize(dev)
  t_work = time.perf_counter() - t0
  send = torch.stack(mine)
  if backend != 'nccl':
    send = send.cpu()
  parts = [torch.empty_like(send) for _ in range(world)]
  dist.all_gather(parts, send)
  torch.cuda.synchronize(dev)
  total = time.perf_counter() - t0
  t = torch.tensor([total, t_work], dtype=torch.float64,
                   device=dev if backend == 'nccl' else 'cpu')
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  total, t_work = [float(v) for v in t.cpu()]
  pairs = 4 * chunks_per_rank * world
  patches = int(np.prod(mine[0].shape[2:]))
  return {
      'workload': f'{chunks_per_rank * world} z chunks (2 x 2 tiles of 512^3 each), '
                  f'{chunks_per_rank} per rank: 4 volumetric tile-pair flows (80^3 patches, step 40, '
                  f'512 x 512 x 120 overlap) + {iters} FIRE steps of the [3,4,12,12,12] montage mesh '
                  'per chunk; flow fields gathered at the end (BASELINE configs[4])',
      'tile_pairs': pairs, 'patches_per_pair': patches,
      'ms_total': round(total * 1e3, 3), 'ms_work': round(t_work * 1e3, 3),
      'tile_pairs_per_s': pairs / total,
      'mvox_s': round(pairs * 512.0 * 512 * 120 / total / 1e6, 1),
      'gathered_bytes_per_rank': int(send.numel() * 4),
      'finite_fraction': float(torch.isfinite(torch.stack(parts)[..., :3, :, :, :]).float().mean()),
  }


def mesh_sharded_leg(bands_per_rank, dev, rank, world, iters=200, loopback=False):
  """ONE [2, 64, 204, 204] mesh (the configs[2] montage size) split into
  world x bands_per_rank bands of rows; the whole chunk of steps runs inside
  sfm_mesh_relax_banded (halo rows between ranks through sfm_comm_* = RCCL,
  between local bands by device copies), next to the un-split relaxation."""
  import torch
  from sofima_amd import dist as sdist, mesh
  rng = np.random.default_rng(7)
  shape = (2, 64, 204, 204)
  prev = rng.standard_normal(shape).astype(np.float32) * 2
  cfg = mesh.IntegrationConfig(
      dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(20.0, 20.0), num_iters=iters,
      max_iters=iters, stop_v_max=1e-9, dt_max=1000, start_cap=0.01, final_cap=10,
      prefer_orig_order=True)
  x0 = np.zeros(shape, np.float32)
  out = {'mesh': list(shape), 'iters': iters, 'bands_per_rank': bands_per_rank,
         'ranks': world}
  import torch.distributed as dist
  if world > 1:
    out['transport'] = ('RCCL send / recv + all-gather (sfm_comm_*)'
                        if dist.get_backend() == 'nccl' else
                        'host-staged callbacks over ' + dist.get_backend())
  mesh.relax_mesh(x0, prev, cfg)
  torch.cuda.synchronize(dev)
  t0 = time.perf_counter()
  mesh.relax_mesh(x0, prev, cfg)
  torch.cuda.synchronize(dev)
  out['unsplit_us_per_step'] = round((time.perf_counter() - t0) / iters * 1e6, 2)
  sdist.relax_mesh_banded(x0, prev, cfg, bands_per_rank=bands_per_rank,
                          loopback=loopback)  # warm-up
  torch.cuda.synchronize(dev)
  tm = {}
  sdist.relax_mesh_banded(x0, prev, cfg, bands_per_rank=bands_per_rank, timing=tm,
                          loopback=loopback)
  if loopback:
    out['transport'] = 'RCCL self send / recv between the local bands (loop-back)'
  spent = tm['banded_chunk_s']
  if world > 1:                      # max over ranks, like every other time here
    t = torch.tensor([spent], dtype=torch.float64,
                     device=dev if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    spent = tm['banded_chunk_s'] = float(t.item())
  out['banded_us_per_step'] = round(tm['banded_chunk_s'] / iters * 1e6, 2)
  out['banded_over_unsplit'] = round(out['banded_us_per_step'] / out['unsplit_us_per_step'], 3)
  out['node_updates_per_s'] = float(np.prod(shape[1:])) * iters / tm['banded_chunk_s']
  out['note'] = ('one mesh, %d band(s) per rank x %d rank(s); chunk of %d steps in ONE '
                 'sfm_mesh_relax_banded call' % (bands_per_rank, world, iters))
  return out


def spawn_ranks(n):
  """Re-executes this script as n ranks under torch.distributed.run."""
  import socket
  import subprocess
  sock = socket.socket()
  sock.bind(('127.0.0.1', 0))
  port = sock.getsockname()[1]
  sock.close()
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
         '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.call(cmd, env=env)


def cpu_baseline(size, seed, pair='warped'):
  """Runs the CPU leg in a child process (no HIP context there, so it can fork a
  worker pool) and returns its JSON object."""
  import subprocess
  cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only',
         '--size', str(size), '--seed', str(seed), '--pair', pair]
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
  if out.returncode != 0:
    return {'error': out.stderr[-400:]}
  return json.loads(out.stdout.strip().splitlines()[-1])


_CPU = {}


def _cpu_flow_batch(b):
  from oracle import flow_oracle
  st = _CPU['plan']
  lo = b * _CPU['batch']
  rows = slice(lo, lo + _CPU['batch'])
  flow_oracle.batched_xcorr_peaks(_CPU['pre'], _CPU['post'], None, None, (PATCH, PATCH),
                                  st[0][rows], None, 2, 0.5, 5, (PATCH, PATCH),
                                  st[1][rows], workers=1)
  return _CPU['batch']


def _cpu_mesh_replica(seed):
  import types
  from oracle import mesh_oracle
  rng = np.random.default_rng(seed)
  prev = rng.standard_normal((2, 1, 205, 205)).astype(np.float32)
  cfg = types.SimpleNamespace(**_CPU['cfg'])
  t0 = time.perf_counter()
  mesh_oracle.relax_mesh(np.zeros_like(prev), prev, cfg)
  return time.perf_counter() - t0


def cpu_baseline_child(size, seed, pair='warped'):
  """The CPU oracle (a NumPy / SciPy port of the reference algorithm) on ALL host
  cores: one process per core (fork), each correlating whole reference batches
  of the same pair (FFT form + peak statistics, single-threaded FFTs), then one
  independent 205^2 mesh relaxation per core."""
  import multiprocessing as mp
  cores = os.cpu_count() or 1
  pre, post = synth_pair(size, seed, warp=WARP if pair == 'warped' else None)
  n_grid = (size - (PATCH - STEP)) // STEP
  # reference batches small enough that every host core gets at least one
  batch = 256
  while batch > 32 and (n_grid * n_grid) // batch < cores:
    batch //= 2
  yy, xx = np.mgrid[:n_grid, :n_grid]
  starts = np.stack([yy.ravel(), xx.ravel()], axis=1).astype(np.int64) * STEP
  n_batches = starts.shape[0] // batch
  # bounded sample: at most 3 batches per core, whole batches of the real grid
  n_b = int(min(n_batches, 3 * cores))
  _CPU.update(pre=pre, post=post, plan=(starts, starts), batch=batch)
  iters = 300
  _CPU['cfg'] = dict(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(STEP, STEP),
                     num_iters=iters, max_iters=iters, stop_v_max=0.005, fire=True,
                     f_alpha=0.99, f_inc=1.1, f_dec=0.5, alpha=0.1, n_min=5, dt_max=1000,
                     start_cap=0.01, final_cap=10, cap_scale=1.1, cap_upscale_every=100,
                     prefer_orig_order=True, remove_drift=False)
  os.environ['OMP_NUM_THREADS'] = '1'
  os.environ['OPENBLAS_NUM_THREADS'] = '1'
  workers = min(cores, n_b)
  n_b = n_b // workers * workers      # whole rounds: every worker the same load
  with mp.get_context('fork').Pool(workers) as pool:
    pool.map(_cpu_flow_batch, range(workers))          # warm-up (imports, FFT plans)
    t0 = time.perf_counter()
    done = sum(pool.map(_cpu_flow_batch, range(n_b), chunksize=1))
    t = time.perf_counter() - t0
  with mp.get_context('fork').Pool(cores) as pool:
    t0 = time.perf_counter()
    pool.map(_cpu_mesh_replica, range(cores), chunksize=1)
    tm = time.perf_counter() - t0
  print(json.dumps({
      'value': done * STEP * STEP / 1e6 / t, 'unit': 'Mpix/s', 'cores': cores,
      'kind': 'port',
      'sample': f'{done} patches ({n_b} reference batches of {batch}) of the same '
                f'{size}^2 pair, one process per batch on {workers} of {cores} cores, '
                f'FFT form (scipy.fft) + peak statistics; {t:.1f} s',
      'patches_per_s': done / t,
      'mesh': {'value': cores * 205 * 205 * iters / tm, 'unit': 'node-updates/s',
               'cores': cores,
               'sample': f'{cores} independent [2,1,205,205] relaxations of {iters} FIRE '
                         f'steps, one per core (NumPy); {tm:.1f} s'},
  }))


if __name__ == '__main__':
  main()
