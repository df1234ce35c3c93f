#!/usr/bin/env python3
"""Timeline of the LAST flow + mesh step of a bench.py kernel trace (rocprofv3 rocpd
database): start offset, duration and the idle gap in front of every dispatch, from
the step's first kernel to the persistent mesh kernel."""
import sqlite3
import sys


def main(path):
  db = sqlite3.connect(path)
  rows = db.execute('select name, start, end from kernels order by start').fetchall()
  last = max(i for i, r in enumerate(rows) if 'mesh_persist' in r[0])
  first = max(i for i, r in enumerate(rows[:last]) if 'mesh_persist' in r[0]) + 1 if any(
      'mesh_persist' in r[0] for r in rows[:last]) else 0
  # skip the tail of the previous step (commit / stats kernels)
  while first < last and ('persist_commit' in rows[first][0] or 'stats_kernel' in rows[first][0]):
    first += 1
  t0 = rows[first][1]
  prev_end = t0
  busy = 0
  for name, s, e in rows[first:last + 1]:
    short = name.replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    print(f'{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {short}')
    busy += e - s
    prev_end = max(prev_end, e)
  print(f'span {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, idle {(prev_end - t0 - busy) / 1e3:.1f} us')


if __name__ == '__main__':
  main(sys.argv[1])
