import csv, sys, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:48]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
    cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        n = cnt[(k, c)]
        print('   %-28s total %.4g  per-dispatch %.4g  (n=%d)' % (c, v, v / n, n))
