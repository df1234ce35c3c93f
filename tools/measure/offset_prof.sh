# Kernel trace of stitch_rigid._estimate_offset on a 4096 x 300 overlap: per-kernel
# summary + the timeline of ONE call (start offsets, durations, gaps).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/offprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/t -o m -- python $R/tools/measure/offset_time.py > $O/log.txt 2>&1
DB=$(find $O/t -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $DB 2>&1 | head -24 | cut -c1-160 > $O/summary.md
python - "$DB" > $O/timeline.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, start, end from kernels order by start').fetchall()
# last third of the launches = the last of the three calls
n = len(rows) // 3
last = rows[-n:]
t0 = last[0][1]
prev_end = t0
busy = 0
for name, s, e in last:
  print('%8.1f us  +gap %6.1f  dur %7.1f  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name[:70]))
  prev_end = e
  busy += e - s
print('launches %d, span %.1f us, busy %.1f us' % (n, (last[-1][2] - t0) / 1e3, busy / 1e3))
PY
tail -3 $O/log.txt; cat $O/summary.md; tail -70 $O/timeline.txt
find $O -name '*.db' -delete
