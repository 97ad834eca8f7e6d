"""Stability check: many steps of bench.py's step functions in one process -- step time at the start and at the end, device
memory (allocated / reserved) and host RSS growth (developer tool).  usage: python tools/soak.py [hot|full] [steps]"""
import json
import os
import resource
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else 'hot'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
# two bench runs in one process would need bench internals; run the benchmark twice with different lengths instead and
# compare per-step time and peak memory reported by a small wrapper
code = r'''
import json, os, resource, sys, time
sys.argv = ['bench.py', '--mode', '%s', '--timed-only', '--steps', '%d', '--warmup', '5']
sys.path.insert(0, %r)
import torch, bench, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(json.dumps({'steps': d['steps'], 'ms_per_step': d['ms_per_step'], 'value': d['value'],
                  'max_allocated_MB': torch.cuda.max_memory_allocated() / 2**20, 'reserved_MB': torch.cuda.memory_reserved() / 2**20,
                  'rss_MB': resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024}))
'''
out = []
for n in (20, steps):
    r = subprocess.run([sys.executable, '-c', code % (mode, n, REPO)], capture_output=True, text=True, cwd=REPO)
    if r.returncode != 0:
        print(r.stderr[-2000:])
        sys.exit(1)
    out.append(json.loads(r.stdout.strip().splitlines()[-1]))
    print(out[-1], flush=True)
a, b = out
print('ms_per_step %.3f -> %.3f; max allocated %.0f -> %.0f MB; reserved %.0f -> %.0f MB; host RSS %.0f -> %.0f MB'
      % (a['ms_per_step'], b['ms_per_step'], a['max_allocated_MB'], b['max_allocated_MB'], a['reserved_MB'], b['reserved_MB'],
         a['rss_MB'], b['rss_MB']))
