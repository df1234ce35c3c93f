# LDS counters of the correlation kernel (production launch): bank conflicts against active cycles
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_lds; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-legs --sustain 0 --no-multi-gpu-legs"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $O/a -o a -- $B --steps 1 --warmup 1 --mesh-iters 10 > $O/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --output-format csv -d $O/b -o b -- $B --steps 1 --warmup 1 --mesh-iters 10 > $O/b.log 2>&1
python - <<PY
import csv, collections, glob
for d in ('a', 'b'):
  fs = glob.glob('$O/%s/**/*counter_collection.csv' % d, recursive=True)
  if not fs:
    print(d, 'no csv'); print(open('$O/%s.log' % d).read()[-600:]); continue
  agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
  for r in csv.DictReader(open(fs[0])):
    a = agg[r['Kernel_Name'][:50]][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
  for k, v in agg.items():
    if 'xcorr_mfma' in k or 'prep_same' in k:
      print(k, {c: (n, s / n) for c, (n, s) in v.items()})
PY
find $O -name '*.csv' -size +2M -delete
