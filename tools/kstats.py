"""Per-step kernel statistics from a rocprofv3 --kernel-trace --stats run (developer tool):
python tools/kstats.py <dir with *kernel_stats.csv> <steps profiled> [top]"""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = sorted(glob.glob(d + '/**/*kernel_stats.csv', recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('%s: %.2f ms of kernel time per step over %d steps' % (f, tot / steps / 1e6, steps))
for r in rows[:top]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print('%-70s %6.1f /step  avg %8.1f us  %7.3f ms/step' % (n[:70], float(r['Calls']) / steps, float(r['AverageNs']) / 1e3,
                                                              float(r['TotalDurationNs']) / steps / 1e6))
