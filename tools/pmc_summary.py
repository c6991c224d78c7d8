"""Aggregate rocprofv3 counter_collection csv files per kernel: python tools/pmc_summary.py gpurun_out/pmc_*"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][-40:]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            n[(k, r['Counter_Name'])] += 1
        print(f)
        for k in acc:
            print('  ', k, {c: (round(v / n[(k, c)], 1), n[(k, c)]) for c, v in acc[k].items()})
