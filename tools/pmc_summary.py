#!/usr/bin/env python3
"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) per kernel.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> out.json [git sha]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
counter values are KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide
coalesced streaming reads, so `fetch_bytes_x2` is given next to the raw value
(the factor is calibrated for 16 B/lane streams only; WRITE_SIZE needs none).
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
  agg = collections.defaultdict(lambda: [0, 0.0])
  for r in csv.DictReader(open(path)):
    if r['Counter_Name'] != counter:
      continue
    k = r['Kernel_Name']
    agg[k][0] += 1
    agg[k][1] += float(r['Counter_Value'])
  return agg


def main(fetch_csv, write_csv, out, build='unknown', patches_per_launch=None):
  f = per_kernel(fetch_csv, 'FETCH_SIZE')
  w = per_kernel(write_csv, 'WRITE_SIZE')
  res = {}
  for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[1] + w.get(k, [0, 0])[1])):
    nf, vf = f.get(k, [0, 0.0])
    nw, vw = w.get(k, [0, 0.0])
    fetch = vf / nf * 1024 if nf else 0.0
    write = vw / nw * 1024 if nw else 0.0
    res[k] = {
        'launches': max(nf, nw),
        'fetch_bytes_raw': round(fetch), 'fetch_bytes_x2': round(2 * fetch),
        'write_bytes': round(write),
        'hbm_bytes_per_launch': round(2 * fetch + write),
    }
  # the build the counters belong to (bench.py reports it next to the traffic)
  res['_meta'] = {'git_sha': build}
  if patches_per_launch:
    res['_meta']['patches_per_launch'] = float(patches_per_launch)
  json.dump(res, open(out, 'w'), indent=1)
  for k, v in [kv for kv in res.items() if kv[0] != '_meta'][:6]:
    print(k[:70], v)


if __name__ == '__main__':
  main(*sys.argv[1:6])
