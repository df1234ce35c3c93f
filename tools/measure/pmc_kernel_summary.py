"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files; argv[1] = kernel name substring."""
import csv, sys, collections
pat = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[2:]:
  for r in csv.DictReader(open(path)):
    k = r['Kernel_Name']
    if pat not in k: continue
    a = agg[k[:70]][r['Counter_Name']]
    a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in agg.items():
  print(k)
  for c, (n, v) in sorted(d.items()):
    print('   %-28s %16.1f per launch (%d)' % (c, v / n, n))
