#!/usr/bin/env python3
"""HBM bytes of ONE call of a bench leg from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE) of a process that ran the leg `calls` times.

usage: pmc_leg_summary.py <leg> <calls> <fetch.csv> <write.csv> <out.json> <git sha>

Units and corrections as tools/pmc_summary.py (MI355X_MICROARCH.md, HBM section):
KiB counters; FETCH_SIZE x 2 on gfx950 for wide streaming reads.  Every dispatch
of the process is counted (the leg's own launches dominate by orders of magnitude;
the per-kernel split is kept).  The file accumulates one entry per leg."""
import collections
import csv
import json
import os
import sys


def total(path, counter):
  agg = collections.defaultdict(lambda: [0, 0.0])
  for r in csv.DictReader(open(path)):
    if r['Counter_Name'] == counter:
      a = agg[r['Kernel_Name']]
      a[0] += 1
      a[1] += float(r['Counter_Value']) * 1024
  return agg


def main(leg, calls, fetch_csv, write_csv, out, sha):
  calls = int(calls)
  f, w = total(fetch_csv, 'FETCH_SIZE'), total(write_csv, 'WRITE_SIZE')
  per = {}
  for k in set(f) | set(w):
    per[k] = {'launches_per_call': round(max(f.get(k, [0])[0], w.get(k, [0])[0]) / calls, 2),
              'fetch_bytes_x2': round(2 * f.get(k, [0, 0.0])[1] / calls),
              'write_bytes': round(w.get(k, [0, 0.0])[1] / calls)}
    per[k]['hbm_bytes'] = per[k]['fetch_bytes_x2'] + per[k]['write_bytes']
  top = sorted(per.items(), key=lambda kv: -kv[1]['hbm_bytes'])
  res = json.load(open(out)) if os.path.exists(out) else {}
  if res.get('_meta', {}).get('git_sha') != sha:
    res = {'_meta': {'git_sha': sha, 'method': __doc__.split('\n\n')[2]}}
  res[leg] = {'calls_counted': calls,
              'hbm_bytes_per_call': sum(v['hbm_bytes'] for v in per.values()),
              'kernels': {k[:90]: v for k, v in top[:8]}}
  json.dump(res, open(out, 'w'), indent=1)
  print(leg, res[leg]['hbm_bytes_per_call'], [(k[:50], v['hbm_bytes']) for k, v in top[:4]])


if __name__ == '__main__':
  main(*sys.argv[1:7])
