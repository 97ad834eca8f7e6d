"""Per-queue timeline of one steady-state step from a rocprofv3 --kernel-trace run (developer tool).
usage: python tools/timeline.py <dir with *kernel_trace.csv> [marker kernel substring, default FusedOptimizer] [detail queue]
Steps are cut at the marker kernel (the optimizer step); the last full step is analysed: per queue the busy time, the idle gaps,
and -- for the queue given -- every kernel in order with its gap to the predecessor."""
import csv, glob, re, sys
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else 'FusedOptimizer'
detail = int(sys.argv[3]) if len(sys.argv) > 3 else None
f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[0]
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
marks = [r['s'] for r in rows if marker in r['Kernel_Name']]
if len(marks) < 3:
    sys.exit('fewer than 3 steps found')
t0, t1 = marks[-3], marks[-2]
step = [r for r in rows if t0 <= r['s'] < t1]
print('step span %.3f ms, %d kernels' % ((t1 - t0) / 1e6, len(step)))
queues = sorted(set(r['Queue_Id'] for r in step), key=int)


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', n)[:60]


for q in queues:
    ks = [r for r in step if r['Queue_Id'] == q]
    busy = sum(r['e'] - r['s'] for r in ks)
    gaps = [b['s'] - a['e'] for a, b in zip(ks, ks[1:])]
    idle = sum(g for g in gaps if g > 0)
    print('queue %s: %4d kernels, busy %.3f ms, idle between kernels %.3f ms (gaps > 20 us: %d, total %.3f ms), first %.3f last %.3f' % (
        q, len(ks), busy / 1e6, idle / 1e6, sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6,
        (ks[0]['s'] - t0) / 1e6, (ks[-1]['e'] - t0) / 1e6))
if detail is not None:
    ks = [r for r in step if int(r['Queue_Id']) == detail]
    prev = None
    for r in ks:
        gap = (r['s'] - prev) / 1e3 if prev is not None else 0.0
        print('%9.3f  +%7.1f us gap  %8.1f us  %s' % ((r['s'] - t0) / 1e6, gap, (r['e'] - r['s']) / 1e3, short(r['Kernel_Name'])))
        prev = r['e']
