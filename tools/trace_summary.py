"""Per-kernel and per-queue summary of a rocprofv3 --kernel-trace CSV (developer tool).
usage: trace_summary.py kernel_trace.csv [fraction_of_the_run_to_skip] [top_n]"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0')))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows.sort(key=lambda t: t[1])
t0, t1 = rows[0][1], max(r[2] for r in rows)
cut = t0 + skip * (t1 - t0)
agg, queues = {}, {}
busy_iv = []
for name, s, e, q in rows:
    if s < cut:
        continue
    n = name.replace('(anonymous namespace)::', '')
    m = re.match(r'(void )?([\w:]+(<[^(]*>)?)\(', n)
    short = m.group(2) if m and 'at::native' not in n else n[:70]
    a = agg.setdefault(short, [0, 0])
    a[0] += 1
    a[1] += e - s
    queues[q] = queues.get(q, 0) + e - s
    busy_iv.append((s, e))
busy_iv.sort()
busy, cs, ce = 0, None, None
for s, e in busy_iv:
    if cs is None or s > ce:
        if cs is not None:
            busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += (ce - cs) if cs is not None else 0
tot = sum(a[1] for a in agg.values())
print('after the cut: kernel time %.2f ms, wall %.2f ms, GPU busy (union) %.2f ms, dispatches %d' % (tot / 1e6, (t1 - cut) / 1e6, busy / 1e6, len(busy_iv)))
print('per queue (ms):', {q: round(v / 1e6, 2) for q, v in queues.items()})
print('%-7s %10s %6s %10s  %s' % ('share', 'total ms', 'calls', 'avg us', 'kernel'))
for k, (c, d) in sorted(agg.items(), key=lambda t: -t[1][1])[:top]:
    print('%6.2f%% %10.3f %6d %10.1f  %s' % (100.0 * d / tot, d / 1e6, c, d / c / 1e3, k[:100]))
