#!/usr/bin/env python3
"""Rehearsal of `bench.py --gpus N` (default 8) on a ONE-GPU box.

The driver's scaling run launches `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...` on an 8-GPU node under a wall-clock limit.  No such node is
available to the builder, so this script runs the very same command line with the
ranks sharing cuda:0 and talking over gloo (SFM_BENCH_BACKEND=gloo,
SFM_BENCH_ONE_DEVICE=1: the mechanism of tests/test_gpu_dist.py::
test_bench_runs_as_two_ranks), at the FULL default problem sizes (64 synthetic 8192^2
sections, 16 volumetric chunks... everything the N = 8 line carries), and records

  * the one JSON line of rank 0 (n_gpus, multi_gpu.* without `error`),
  * the wall time of the whole command and bench.py's own per-phase wall clock,

into profiles/r06_bench_8ranks_one_gpu.json.  The throughput numbers in that file are
those of 8 ranks time-slicing one GPU -- a correctness / wall-time rehearsal, not a
scaling measurement.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--ranks', type=int, default=8)
  ap.add_argument('--steps', type=int, default=3)
  ap.add_argument('--warmup', type=int, default=1)
  ap.add_argument('--size', type=int, default=8192)
  ap.add_argument('--limit', type=float, default=900.0, help='seconds the run may take')
  ap.add_argument('--out', default=os.path.join(ROOT, 'profiles',
                                                'r06_bench_8ranks_one_gpu.json'))
  args = ap.parse_args()
  sock = socket.socket()
  sock.bind(('127.0.0.1', 0))
  port = sock.getsockname()[1]
  sock.close()
  env = dict(os.environ, SFM_BENCH_BACKEND='gloo', SFM_BENCH_ONE_DEVICE='1',
             HSA_ENABLE_IPC_MODE_LEGACY='0')
  # the driver's own launch line
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         f'--nproc-per-node={args.ranks}', '--master-addr', '127.0.0.1',
         '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
         '--gpus', str(args.ranks), '--steps', str(args.steps), '--warmup', str(args.warmup),
         '--size', str(args.size)]
  t0 = time.time()
  try:
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT,
                         timeout=args.limit)
    rc, stdout, stderr = out.returncode, out.stdout, out.stderr
  except subprocess.TimeoutExpired as e:
    rc, stdout, stderr = -9, (e.stdout or b'').decode(errors='replace'), \
        (e.stderr or b'').decode(errors='replace')
  wall = time.time() - t0
  lines = [l for l in stdout.splitlines() if l.startswith('{')]
  rec = {
      'command': ' '.join(cmd[1:]).replace(ROOT + '/', ''),
      'env': {'SFM_BENCH_BACKEND': 'gloo', 'SFM_BENCH_ONE_DEVICE': '1'},
      'what': f'{args.ranks} ranks sharing ONE MI355X over gloo: rehearsal of the '
              'scaling run (wall time and code paths, not a scaling measurement)',
      'returncode': rc, 'wall_s': round(wall, 1), 'limit_s': args.limit,
      'json_lines': len(lines),
  }
  ok = rc == 0 and len(lines) == 1
  if lines:
    line = json.loads(lines[-1])
    rec['line'] = line
    mg = line.get('multi_gpu', {})
    errs = {k: v['error'] for k, v in mg.items() if isinstance(v, dict) and 'error' in v}
    rec['multi_gpu_errors'] = errs
    ok = ok and line.get('n_gpus') == args.ranks and not errs and all(
        k in mg for k in ('section_chain', 'volumetric_chunks', 'mesh_sharded'))
  else:
    rec['stderr_tail'] = stderr[-3000:]
  rec['ok'] = bool(ok)
  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  with open(args.out, 'w') as f:
    json.dump(rec, f, indent=1)
  print(json.dumps({k: rec[k] for k in ('ok', 'returncode', 'wall_s', 'json_lines')}))
  if lines:
    print(json.dumps({'wall_s': rec['line'].get('wall_s'),
                      'value': rec['line'].get('value'),
                      'multi_gpu_errors': rec.get('multi_gpu_errors')}))
  else:
    print(stderr[-3000:])
  sys.exit(0 if ok else 1)


if __name__ == '__main__':
  main()
