import csv, glob, collections, sys
for d in sorted(glob.glob('gpurun_out/bb_sq*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:24] + ' grid' + r.get('Grid_Size', '')
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, v in acc.items():
            if 'bb' in k: print(k, {a: round(b / 1e6, 2) for a, b in v.items()})
