"""Timeline of one steady-state step from a rocprofv3 --kernel-trace CSV: busy time per HIP queue, their union, the gaps of
the main queue and (with -v) every kernel longer than 60 us or off the main queue (developer tool).
usage: python tools/step_timeline.py <kernel_trace.csv> [-v] [step index from the end, default 5]"""
import collections
import csv
import re
import sys

path = sys.argv[1]
verbose = '-v' in sys.argv
nums = [a for a in sys.argv[2:] if a.lstrip('-').isdigit()]
back = int(nums[0]) if nums else 5
rows = list(csv.DictReader(open(path)))
for r in rows:
    r['s'], r['e'], r['q'] = int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id']
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    r['n'] = re.sub(r'\(.*', '', n)[:44]
rows.sort(key=lambda r: r['s'])
marks = [r['s'] for r in rows if r['n'].startswith('feature_sample_rows')]      # first kernel of a step's forward
a, b = marks[-back - 1], marks[-back]
step = [r for r in rows if a <= r['s'] < b]
print('step %.3f ms, %d kernels' % ((b - a) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r['q']].append(r)
for q, l in sorted(byq.items()):
    print('queue %s: %4d kernels, busy %.3f ms' % (q, len(l), sum(r['e'] - r['s'] for r in l) / 1e6))
ev = sorted((r['s'], r['e']) for r in step)
u, (cs, ce) = 0, ev[0]
for s, e in ev[1:]:
    if s > ce:
        u += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
print('union busy %.3f ms' % ((u + ce - cs) / 1e6))
mq = max(byq, key=lambda q: len(byq[q]))
l = byq[mq]
gaps = sorted(((y['s'] - x['e'], x['n'], y['n'], (x['e'] - a) / 1e6) for x, y in zip(l, l[1:]) if y['s'] - x['e'] > 10000), reverse=True)
print('main queue %s: gaps > 10 us: %d, %.3f ms; all gaps %.3f ms' % (mq, len(gaps), sum(g[0] for g in gaps) / 1e6,
      sum(max(0, y['s'] - x['e']) for x, y in zip(l, l[1:])) / 1e6))
for g in gaps[:12]:
    print('   %7.1f us at %7.3f ms  %s -> %s' % (g[0] / 1e3, g[3], g[1], g[2]))
if verbose:
    for r in step:
        if r['e'] - r['s'] > 60000 or r['q'] != mq:
            wgs = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_X']))
            print('q%s %8.3f +%7.3f %s  wg %d' % (r['q'], (r['s'] - a) / 1e6, (r['e'] - r['s']) / 1e6, r['n'], wgs))
