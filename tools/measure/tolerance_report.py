"""Summarises an SFM_TOL_REPORT file (tests/conftest.py): per call site of
np.testing.assert_allclose the tolerance it states and how much of it the worst
comparison of the run used (1.0 = at the limit).

  SFM_TOL_REPORT=gpurun_out/tol.jsonl python -m pytest tests -m gpu -q
  python tools/measure/tolerance_report.py gpurun_out/tol.jsonl
"""
import collections
import json
import sys

rows = collections.OrderedDict()
for line in open(sys.argv[1]):
  r = json.loads(line)
  k = (r['site'], r['rtol'], r['atol'])
  o = rows.setdefault(k, dict(r, count=0))
  o['count'] += 1
  for f in ('used', 'max_abs', 'max_rel'):
    o[f] = max(o[f], r[f])
  o['scale'] = max(o['scale'], r['scale'])
print('%-28s %9s %9s %8s %10s %10s %10s %5s' % ('site', 'rtol', 'atol', 'used', 'max_abs',
                                              'max_rel', 'scale', 'n'))
for (site, rtol, atol), o in sorted(rows.items(), key=lambda kv: -kv[0][1]):
  print('%-28s %9.2g %9.2g %8.3f %10.3g %10.3g %10.3g %5d' % (
      site, rtol, atol, o['used'], o['max_abs'], o['max_rel'], o['scale'], o['count']))
