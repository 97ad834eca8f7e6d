"""Per-kernel summary of a rocprofv3 rocpd database (developer tool): name, calls, total / average duration.
usage: rocpd_summary.py results.db [first_dispatch_fraction_to_skip] [top_n]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cur = db.cursor()
cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
rows = list(cur.execute('select name, start, end from kernels order by start'))
t0, t1 = rows[0][1], rows[-1][2]
cut = t0 + skip * (t1 - t0)
agg = {}
for name, s, e in rows:
    if s < cut:
        continue
    n = name.replace('(anonymous namespace)::', '')
    m = re.match(r'(void )?([\w:]+(<[^(]*>)?)\(', n)
    short = m.group(2) if m and 'at::native' not in n else n[:60]
    a = agg.setdefault(short, [0, 0])
    a[0] += 1
    a[1] += e - s
tot = sum(a[1] for a in agg.values())
print('kernels after the cut: total %.2f ms, wall %.2f ms' % (tot / 1e6, (t1 - cut) / 1e6))
print('%-7s %10s %6s %10s  %s' % ('share', 'total ms', 'calls', 'avg us', 'kernel'))
for k, (c, d) in sorted(agg.items(), key=lambda t: -t[1][1])[:top]:
    print('%6.2f%% %10.3f %6d %10.1f  %s' % (100.0 * d / tot, d / 1e6, c, d / c / 1e3, k[:90]))
