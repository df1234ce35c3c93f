import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for path in sys.argv[1:]:
  for r in csv.DictReader(open(path)):
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
  if 'shared2d' not in k and 'integrate' not in k: continue
  print(k)
  for c, v in sorted(d.items()):
    n = cnt[(k, c)]
    print('   %-28s %.4g per launch (%d launches)' % (c, v / n, n))
