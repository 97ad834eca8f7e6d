"""Per-kernel means of the counters in rocprofv3 counter_collection CSVs -- developer tool.
usage: pmc_report.py <dir> [kernel substring ...]"""
import csv, glob, sys, collections
d = sys.argv[1]
subs = sys.argv[2:] or ['conv3d']
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        for s in subs:
            if s in name:
                short = name[name.index(s):].split('(')[0]
                acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print('   %-32s n=%-3d mean=%.4g' % (c, len(v), sum(v) / len(v)))
