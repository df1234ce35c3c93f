import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[1:]:
  for r in csv.DictReader(open(path)):
    k = r['Kernel_Name'][:60]
    if 'masked_phase' not in k and 'masked_tables' not in k: continue
    a = agg[k][r['Counter_Name']]
    a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in agg.items():
  print(k)
  for c, (n, v) in sorted(d.items()):
    print('   %-28s %14.1f per launch (%d)' % (c, v / n, n))
