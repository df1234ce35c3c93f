#!/usr/bin/env python3
"""Prints a per-kernel summary (calls, total/avg/min/max ns, %) from a rocprofv3
rocpd SQLite database; used to write the summaries under profiles/."""
import sqlite3
import sys


def main(path):
  db = sqlite3.connect(path)
  cur = db.cursor()
  cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
  rows = cur.execute('select name, start, end from kernels').fetchall()
  agg = {}
  for name, s, e in rows:
    d = e - s
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
  total = sum(a[1] for a in agg.values()) or 1
  print('| kernel | calls | total_ms | avg_us | min_us | max_us | % |')
  print('|---|---|---|---|---|---|---|')
  for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    short = name if len(name) < 90 else name[:87] + '...'
    print(f'| {short} | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.2f} | '
          f'{a[2] / 1e3:.2f} | {a[3] / 1e3:.2f} | {100 * a[1] / total:.2f} |')


if __name__ == '__main__':
  main(sys.argv[1])
