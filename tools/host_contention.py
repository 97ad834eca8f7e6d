"""Host cost of the training step when several ranks share one host: N concurrent `bench.py --timed-only` processes (each
a fresh process, as torchrun would start them) on the one leased GPU.  The GPU is shared N ways, so ms_per_step is
meaningless here; what is measured is the per-process host time to enqueue a step (bench.py's host probe) under
contention for the host's threads.  At most 6 processes may use the GPU at once on this pool.  Developer tool."""
import json, os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mode = sys.argv[2] if len(sys.argv) > 2 else 'hot'
threads_per_proc = int(sys.argv[3]) if len(sys.argv) > 3 else 0       # e.g. 2 = the share of a rank when 8 ranks run on a 16-thread host
# points per frame: with a small cloud (e.g. 1500) a step is a few hundred microseconds of GPU work, so that six processes on the ONE
# card do not back their launch queues up into the probe -- the host probe then measures the host side under contention, which is
# what a rank with a GPU of its own would see (the Python / ctypes / launch cost of a step does not depend on the tensor sizes)
points = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
assert n <= 6
all_cpus = sorted(os.sched_getaffinity(0))


def pin(k):
    def f():
        if threads_per_proc:                                             # process k gets ITS OWN threads_per_proc host threads
            os.sched_setaffinity(0, all_cpus[k * threads_per_proc:(k + 1) * threads_per_proc])
    return f


t0 = time.time()
procs = [subprocess.Popen([sys.executable, os.path.join(REPO, 'bench.py'), '--timed-only', '--steps', '12', '--warmup', '4', '--mode', mode,
                           '--points', str(points)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=REPO, preexec_fn=pin(k),
                          env=dict(os.environ, MVX_CPU_THREADS=str(threads_per_proc or 64), OMP_NUM_THREADS=str(threads_per_proc or 16)))
         for k in range(n)]
res = []
for p in procs:
    out = p.communicate(timeout=900)[0].decode()
    line = [l for l in out.splitlines() if l.startswith('{')]
    res.append(json.loads(line[0]) if line else None)
ok = [r for r in res if r]
summary = {'processes': n, 'mode': mode, 'points': points, 'threads_per_process': threads_per_proc, 'host_threads': len(all_cpus), 'wall_s': time.time() - t0,
           'host_enqueue_ms_per_step': [r['host_enqueue_ms_per_step'] for r in ok],
           'ms_per_step_gpu_shared': [r['ms_per_step'] for r in ok]}
print(json.dumps(summary))
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
json.dump(summary, open(os.path.join(REPO, 'gpurun_out', 'host_contention_%s_%d_%dthr_%dpts.json' % (mode, n, threads_per_proc, points)), 'w'), indent=1)
