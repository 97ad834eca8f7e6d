"""Summarise a rocprofv3 --kernel-trace --stats output directory (developer tool)."""
import csv, glob, re, sys, collections
d = sys.argv[1]
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = glob.glob(d + '/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms %.2f -> %.3f ms/frame' % (tot / 1e6, tot / 1e6 / nframes))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    n = r['Name'].replace('(anonymous namespace)::', '')
    m = re.match(r'(void )?([\w:]+(<[^(]*>)?)\(', n)
    short = m.group(2) if m and 'at::native' not in n else n[:70]
    print('%6.2f%% %8.2f ms calls %4s avg %8.1f us  %s' % (float(r['Percentage']), int(r['TotalDurationNs']) / 1e6,
                                                            r['Calls'], float(r['AverageNs']) / 1e3, short[:80]))
