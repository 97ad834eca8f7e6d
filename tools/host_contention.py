"""Host cost of the training step when several ranks share one host: N concurrent `bench.py --timed-only` processes (each
a fresh process, as torchrun would start them) on the one leased GPU.  The GPU is shared N ways, so ms_per_step is
meaningless here; what is measured is the per-process host time to enqueue a step (bench.py's host probe) under
contention for the host's threads.  At most 6 processes may use the GPU at once on this pool.  Developer tool."""
import json, os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mode = sys.argv[2] if len(sys.argv) > 2 else 'hot'
threads_per_proc = int(sys.argv[3]) if len(sys.argv) > 3 else 0       # e.g. 2 = the share of a rank when 8 ranks run on a 16-thread host
assert n <= 6
if threads_per_proc:
    cpus = sorted(os.sched_getaffinity(0))[:n * threads_per_proc]
    os.sched_setaffinity(0, cpus)                                        # inherited by the benchmark processes
t0 = time.time()
procs = [subprocess.Popen([sys.executable, os.path.join(REPO, 'bench.py'), '--timed-only', '--steps', '6', '--warmup', '2', '--mode', mode],
                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=REPO) for _ in range(n)]
res = []
for p in procs:
    out = p.communicate(timeout=900)[0].decode()
    line = [l for l in out.splitlines() if l.startswith('{')]
    res.append(json.loads(line[0]) if line else None)
ok = [r for r in res if r]
summary = {'processes': n, 'mode': mode, 'host_threads': len(os.sched_getaffinity(0)), 'wall_s': time.time() - t0,
           'host_enqueue_ms_per_step': [r['host_enqueue_ms_per_step'] for r in ok],
           'ms_per_step_gpu_shared': [r['ms_per_step'] for r in ok]}
print(json.dumps(summary))
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
json.dump(summary, open(os.path.join(REPO, 'gpurun_out', 'host_contention_%s_%d.json' % (mode, n)), 'w'), indent=1)
