#!/usr/bin/env python3
"""Headline benchmark: KITTI-shaped frames/s through the MVXNet hot path
(voxelize + fusion + VFE + dense 3-D conv, forward + backward) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one batch of `--frames` synthetic ring frames PER GPU (weak scaling): batched GPU
voxelizer, then per frame fusion sampling + fusion MLP + VFE stack + scatter + CML forward and
the full backward (dL/d(middle) is a fixed resident tensor standing for RPN + loss), one flat
gradient all-reduce over RCCL, one AdamW step.  Inputs are resident in HBM before the timed
region.  Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, chip-level parameters (matrix f32, dense)
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (never the 2:1-sparsity figure)


def host_projection(pts, calib):
    """lidar2Img(uncheck=True) + swap to (row, col) in f32 (train.py:31-33), host numpy: input
    preparation outside the timed region."""
    p = np.ones((4, pts.shape[0]), np.float32)
    p[:3] = pts[:, :3].T
    m = (calib['R0_rect'].astype(np.float32) @ calib['Tr_velo_to_cam'].astype(np.float32))
    img = calib['P2'].astype(np.float32) @ (m @ p)
    uv = (img[:2] / img[2]).T
    return uv[:, ::-1].astype(np.float32)


def make_batch(frame_ids, dev, points_per_frame):
    from modules.data import Synthetic as S
    from modules.pipeline import FrameBatch
    cap = points_per_frame
    B = len(frame_ids)
    pts6 = np.zeros((B, cap, 6), np.float32)
    perms = np.zeros((B, cap), np.int32)
    n = np.zeros((B,), np.int32)
    fpn = []
    for k, fid in enumerate(frame_ids):
        pc = S.synth_ring(fid, cap)
        m = min(cap, pc.shape[0])
        pts6[k, :m, :4] = pc[:m]
        pts6[k, :m, 4:] = host_projection(pc[:m], S.KITTI_CALIB)
        perms[k, :m] = S.synth_perm(fid, m)
        n[k] = m
        g = torch.Generator(device='cpu').manual_seed(3000 + fid)
        fpn.append([torch.randn((1, 256, h, w), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
                    for h, w in ((104, 336), (52, 168), (26, 84))])
    return FrameBatch(torch.from_numpy(pts6).to(dev), torch.from_numpy(perms).to(dev),
                      torch.from_numpy(n).to(dev), fpn)


def host_threads():
    """Host threads this process may really use: affinity mask, then the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:
            quota, period = fh.read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get('MVX_CPU_THREADS', '64'))))


def cpu_baseline(points_per_frame, n_frames=3):
    """The CPU oracle (torch-CPU / oneDNN + the plain-C voxelizer) on a bounded sample of the same
    workload (n_frames ring frames, forward + backward, dense as the reference computes it), all host
    threads the cgroup allows.  A reported baseline, not the target."""
    import ctypes
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import mvx_oracle as O
    from modules.data import Synthetic as S
    torch.set_num_threads(host_threads())
    lib = ctypes.CDLL(os.path.join(REPO, 'oracle', 'liboracle_c.so'))
    lib.oracle_group9.restype = ctypes.c_int64
    P = {k: v.requires_grad_(True) for k, v in O.make_params(7).items()}
    G = torch.ones((1, 128, O.VOXELSHAPE[0], O.VOXELSHAPE[1]))
    rng = np.asarray(O.VELORANGE, np.float64)
    size = np.asarray(O.voxelsize(), np.float64)
    frames = []
    for fid in range(n_frames):                       # input preparation is not timed (resident on the GPU side too)
        pc = S.synth_ring(fid, points_per_frame)
        pcd = np.ascontiguousarray(np.concatenate([pc, host_projection(pc, S.KITTI_CALIB)], 1), np.float32)
        frames.append((pcd, S.synth_perm(fid, pcd.shape[0]), [torch.from_numpy(f) for f in S.synth_fpn(fid)]))
    nv = []
    t0 = time.perf_counter()
    for pcd, perm, feats in frames:
        Pn = pcd.shape[0]
        voxel = np.empty((Pn, 35, 9), np.float64)
        uidx = np.empty((Pn, 3), np.float64)
        cnt = np.empty(Pn, np.int64)
        V = lib.oracle_group9(pcd.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(6), perm.ctypes.data_as(ctypes.c_void_p),
                              ctypes.c_int64(Pn), rng.ctypes.data_as(ctypes.c_void_p), size.ctypes.data_as(ctypes.c_void_p),
                              ctypes.c_int32(35), voxel.ctypes.data_as(ctypes.c_void_p),
                              uidx.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
        vox = torch.from_numpy(voxel[:V].astype(np.float32))
        idx = torch.from_numpy(np.concatenate([np.zeros((V, 1)), uidx[:V]], 1).astype(np.int64))
        v23 = O.mvx_point_features(vox, feats, torch.tensor([370.0, 1224.0]), P)
        mid = O.voxelnet_middle(v23, idx, O.strip_prefix(P, 'backbone.'))
        mid.backward(G)
        nv.append(int(V))
    dt = time.perf_counter() - t0
    return {'value': n_frames / dt, 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d ring frames (%d pts, V=%s): C voxelizer + torch-CPU fusion/VFE/CML fwd+bwd, %.1f s'
                      % (n_frames, points_per_frame, nv, dt)}


def isolated_conv_roofline(dev, math):
    """The dominant kernel's launches (conv2/conv3 forward + dgrad) on an otherwise idle GPU: inside the
    step they share the CUs with the side-stream weight-gradient kernels and the other lane's frame, so
    the live figure above includes that sharing; this one prices the kernel alone."""
    from modules import _hip
    import modules.config as cfg
    H, W = cfg.voxelshape[0], cfg.voxelshape[1]
    split = math == 'bf16x3'
    tot_ms, tot_fl = 0.0, 0.0
    for cin, cout, din, sd, pd in ((64, 64, 5, 1, 0), (64, 64, 3, 2, 1)):
        dout = _hip.conv_out_depth(din, sd, pd)
        x = torch.randn((din, H, W, cin), device=dev)
        w = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.04
        b = torch.zeros(cout, device=dev)
        dz = torch.randn((dout, H, W, cout), device=dev)
        wf, wd = _hip.conv3d_pack(w, False, split=split), _hip.conv3d_pack(w, True, split=split)
        for fn, fl in ((lambda: _hip.conv3d_forward(x, wf, b, cout, sd, pd, split=split), _hip.conv_flops(dout, din, H, W, cin, cout, sd, pd)),
                       (lambda: _hip.conv3d_dgrad(dz, wd, din, cin, sd, pd, split=split), _hip.conv_flops(din, dout, H, W, cout, cin, sd, pd, True))):
            fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                fn()
            e.record()
            torch.cuda.synchronize()
            tot_ms += s.elapsed_time(e) / 5
            tot_fl += fl
    mult, peak = (3.0, BF16_MFMA_PEAK_TFLOPS) if split else (1.0, FP32_MFMA_PEAK_TFLOPS)
    ach = mult * tot_fl / (tot_ms * 1e-3) / 1e12
    return {'achieved': ach, 'frac': ach / peak, 'avg_launch_ms': tot_ms / 4}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--frames', type=int, default=4, help='frames per GPU per step')
    ap.add_argument('--points', type=int, default=20000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--convmath', choices=['bf16x3', 'f32'], default=None, help='override config.yml convmath')
    ap.add_argument('--no-alt', action='store_true', help='skip the extra run in the other convolution arithmetic')
    ap.add_argument('--timed-only', action='store_true',
                    help='only warm-up + the timed steps (no alt / unshared / isolated / CPU passes): what profiles/ is made from')
    args = ap.parse_args()
    if args.timed_only:
        args.no_alt = args.no_cpu_baseline = True

    from modules import parallel
    rank, world, local = parallel.init_from_env(os.environ.get('MVX_DIST_BACKEND'))
    assert world == args.gpus or world == 1, 'launch with torchrun --nproc-per-node = --gpus'
    dev = torch.device('cuda', local % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)

    import modules.config as cfg
    from modules import _hip
    from modules import pipeline as _pl
    train_step_frames = _pl.train_step_frame_set if _pl.BATCHED else _pl.train_step_frames
    from MVXNet import MVXNet
    if args.convmath:
        cfg.config['convmath'] = args.convmath
    main_math = cfg.config.get('convmath', 'f32')

    torch.manual_seed(0)
    model = MVXNet().to(dev)
    hot = [p for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
    bucket = parallel.GradBucket(hot)
    opt = torch.optim.AdamW(hot, lr=1e-3, eps=cfg.eps)
    frame_ids = [rank + world * j for j in range(args.frames)]
    batch = make_batch(frame_ids, dev, args.points)
    g = torch.Generator(device='cpu').manual_seed(77)
    grad_mid = (torch.randn((1, 128, cfg.voxelshape[0], cfg.voxelshape[1]), generator=g) * 1e-3).to(dev)
    imsize = [float(v) for v in cfg.imsize]            # host list: no device read-back inside the step
    frames_total = args.frames * world

    from modules import pipeline as pl_mod
    pending = [None]
    step_events = []
    debug_sleep_ms = float(os.environ.get('MVX_DEBUG_HOST_DELAY_MS', '0'))
    pipelined = os.environ.get('MVX_PIPELINE_INPUT', '2') == '1'      # own preparation stream: measured slower
    pipe_mid = os.environ.get('MVX_PIPELINE_INPUT', '2') == '2'       # default: next batch prepared mid-step on the main stream
    mid_ready = [None]

    def step():
        # input double-buffering: this step consumes the batch that was voxelized during the previous one and voxelizes
        # the next (same resident synthetic batch every step; every step still voxelizes exactly once, inside the
        # timed region; MVX_PIPELINE_INPUT=0 voxelizes at the start of the step instead)
        ready = None
        tt = [time.perf_counter()]
        if pipelined:
            # bounded run-ahead: never more than one step of launches in flight (a full HIP queue blocks the host
            # inside launches for milliseconds at a time)
            if len(step_events) >= 2:
                pl_mod._spin(step_events.pop(0))
            if pending[0] is not None:
                ready = pl_mod.prepare_end(pending[0], model.head)
                tt.append(time.perf_counter())
            pending[0] = pl_mod.prepare_begin(batch)
        tt.append(time.perf_counter())
        bucket.zero()
        if debug_sleep_ms > 0:                       # host-slack probe: busy-wait on the host before enqueuing the frames
            t_end = time.perf_counter() + debug_sleep_ms * 1e-3
            while time.perf_counter() < t_end:
                pass
        if pipe_mid:
            nv, statuses, mid_ready[0] = train_step_frames(model, batch, grad_mid, imsize, ready=mid_ready[0], prepare_next=batch)
        else:
            nv, statuses = train_step_frames(model, batch, grad_mid, imsize, ready=ready)
        tt.append(time.perf_counter())
        if pipelined:
            pl_mod.prepare_mid(pending[0], model.head)
        tt.append(time.perf_counter())
        bucket.all_reduce_mean(frames_total)
        opt.step()
        tt.append(time.perf_counter())
        if os.environ.get('MVX_DEBUG_TIMES'):
            sys.stderr.write('step phases ms: ' + ' '.join('%.2f' % ((b - a) * 1e3) for a, b in zip(tt, tt[1:])) + '\n')
        pending_status.extend(statuses)
        if pipelined:
            ev = torch.cuda.Event()
            ev.record()
            step_events.append(ev)
        return nv

    pending_status = []

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    host_ms = []
    exec_stages = []        # per timed_run: executed gather stages counted by the background-aware forward kernel

    def timed_run(warmup, steps):
        nv = None
        _hip.KERNEL_TIMERS = {}                           # the warm-up also fills the pool of timing events
        for _ in range(warmup):
            nv = step()
        fence()
        _hip.recycle_timing_events(_hip.KERNEL_TIMERS)
        gc.collect()
        _hip.KERNEL_TIMERS = {}
        if _hip.EXEC_STAGES is not None:
            _hip.EXEC_STAGES.zero_()
        t0 = time.perf_counter()
        for _ in range(steps):
            nv = step()
        host_dt = time.perf_counter() - t0                 # host time to ENQUEUE the steps
        fence()
        dt_ = time.perf_counter() - t0
        host_ms.append(host_dt / steps * 1e3)
        tm, _hip.KERNEL_TIMERS = _hip.KERNEL_TIMERS, None
        exec_stages.append(int(_hip.EXEC_STAGES.item()) if _hip.EXEC_STAGES is not None else 0)
        if world > 1:
            t = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t)
        return nv, dt_, tm

    def conv_roofline(tm, math, run=0):
        name = 'conv3d_gather_split' if math == 'bf16x3' else 'conv3d_gather_pw'
        ev = tm.get('conv3d_gather_split' if math == 'bf16x3' else 'conv3d_gather', [])
        ms = sum(s.elapsed_time(e) for s, e, _ in ev)
        fl = sum(f for _, _, f in ev)
        bg = tm.get('conv3d_gather_bg', []) if math != 'bf16x3' else []
        dense_fl = fl
        if bg:
            # forward launches with the background rewrite: EXECUTED FLOPs from the kernel's own stage counter
            # (the timer's own figure is the dense-equivalent work of the launch)
            ms += sum(s.elapsed_time(e) for s, e, _ in bg)
            fl += float(exec_stages[run]) * _hip.STAGE_FLOP
            dense_fl += sum(f for _, _, f in bg)
            ev = ev + bg
        tiles = tm.get('conv3d_gather_tiles', []) if math != 'bf16x3' else []
        if tiles:
            # tile-restricted dgrad launches of the same kernel: their executed stages are in the same counter
            ms += sum(s.elapsed_time(e) for s, e, _ in tiles)
            dense_fl += sum(f for _, _, f in tiles)
            ev = ev + tiles
        ms_all, dense_all = ms, dense_fl
        if math == 'bf16x3':
            peak, mult, note = BF16_MFMA_PEAK_TFLOPS, 3.0, ('executed bf16 MFMA FLOPs = 3 x algorithmic '
                                                             '(hi*hi + hi*lo + lo*hi per product)')
            if cfg.config.get('convbackground', True):
                note += ('; the background-aware launches are priced at their DENSE FLOPs (the bf16x3 kernels carry no '
                         'stage counter): an upper bound on the executed rate')
        else:
            peak, mult, note = FP32_MFMA_PEAK_TFLOPS, 1.0, 'exact f32 MFMA, executed = algorithmic FLOPs'
        ach = mult * fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
                'kernel': name + ' (every launch of the step: conv2/conv3 forward + dgrad)', 'launches': len(ev),
                'avg_launch_ms': ms / max(1, len(ev)), 'flop_per_launch': fl / max(1, len(ev)),
                'fp32_equivalent_tflops': fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                # what a dense evaluation of the same launches (incl. the tile-restricted dgrads) would have to sustain
                'dense_equivalent_tflops': dense_all / (ms_all * 1e-3) / 1e12 if ms_all > 0 else 0.0,
                'note': note + ('; forward launches skip voxel-free tiles (convbackground): FLOPs are the EXECUTED ones, '
                                'counted by the kernel' if bg else '')}

    nvox, dt, timers = timed_run(args.warmup, args.steps)

    def host_probe(n=3):
        """Host time to enqueue one step when the launch queues are empty (over `n` steps from an idle GPU, no sync inside):
        what the Python side costs.  Inside the timed region the host runs ahead of the GPU until the HIP queues are full,
        so the time it spends there mostly measures the GPU."""
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        d = (time.perf_counter() - t0) / n * 1e3
        fence()
        return d
    host_probe_ms = host_probe()
    bad = int(torch.stack([t.reshape(()) for t in pending_status]).max()) if pending_status else 0
    assert bad == 0, 'a kernel reported a data-dependent error (status %d)' % bad
    del pending_status[:]
    sparse_quads = int(_hip.SPARSE_QUADS) if _hip.SPARSE_QUADS is not None else 0
    alt = None
    if not args.no_alt:
        alt_math = 'bf16x3' if main_math == 'f32' else 'f32'
        if os.environ.get('MVX_BENCH_ALT_SAME'):
            alt_math = main_math
        cfg.config['convmath'] = alt_math
        _, dt_alt, tm_alt = timed_run(max(2, args.warmup), args.steps)     # allocator re-settles after the switch
        cfg.config['convmath'] = main_math
        alt = {'convmath': alt_math, 'value': frames_total * args.steps / dt_alt, 'unit': 'frames/s',
               'ms_per_step': dt_alt / args.steps * 1e3, 'host_enqueue_ms_per_step': host_ms[-1],
               'roofline': conv_roofline(tm_alt, alt_math, run=1)}

    unshared = None
    if world == 1 and not args.timed_only:
        # The same steps with every kernel alone on the GPU (one lane, weight gradients on the main stream): what the
        # dominant kernel does when it does not share the CUs with the other lane's frame and the side-stream wgrads.
        import modules.pipeline as pl
        old = (pl.LANES, pl.ASYNC_WGRAD)
        pl.LANES, pl.ASYNC_WGRAD = 1, False
        try:
            _, dt_u, tm_u = timed_run(1, 2)
        finally:
            pl.LANES, pl.ASYNC_WGRAD = old
        r = conv_roofline(tm_u, main_math, run=len(exec_stages) - 1)
        unshared = {'achieved': r['achieved'], 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms'],
                    'frames_per_s': frames_total * 2 / dt_u,
                    'note': 'same step, MVX_LANES=1 and weight gradients on the main stream: no kernel shares the GPU'}

    if rank == 0:
        roof = conv_roofline(timers, main_math)
        roof['note'] += ('; measured while the other lane and the side-stream weight-gradient kernels share the CUs '
                         '(see unshared / isolated)')
        if unshared is not None:
            roof['unshared'] = unshared
        if not args.timed_only:
            roof['isolated'] = isolated_conv_roofline(dev, main_math)
        tpath = os.path.join(REPO, 'profiles', 'traffic.json')
        if os.path.exists(tpath) and main_math == 'f32':
            with open(tpath) as fh:
                roof['traffic'] = json.load(fh).get('conv3d_gather_pw_hbm_bytes_per_launch')
        out = {
            'metric': 'KITTI frames/sec (voxelize+VFE+fusion+3Dconv fwd+bwd)',
            'value': frames_total * args.steps / dt,
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'host_enqueue_ms_per_step': host_probe_ms,
            'host_in_timed_region_ms_per_step': host_ms[0],
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32' if main_math == 'f32' else 'f32 (bf16x3 split MFMA, f32 accumulate)',
            'data': 'synthetic',
            'config': {'workload': 'S2 ring frames, %d pts, grid 10x352x400, T=35, %d frames/GPU/step, '
                                   'MVXNet middle (fusion on) fwd+bwd + AdamW, convmath=%s' % (args.points, args.frames, main_math),
                       'frames_per_gpu': args.frames, 'voxels_per_frame': nvox, 'parallelism': 'dp%d' % world},
            'roofline': roof,
        }
        if alt is not None:
            out['alt_modes'] = [alt]
        other = {}
        for name in ('conv3d_gather_sparse_input', 'conv3d_wgrad', 'conv3d_wgrad_bg', 'conv3d_dgrad_sites', 'conv3d_wgrad_sites'):
            evs = timers.get(name, [])
            if evs:
                tms = sum(s.elapsed_time(e) for s, e, _ in evs)
                tfl = sum(f for _, _, f in evs)
                if name == 'conv3d_gather_sparse_input':
                    tfl = sparse_quads * 8 * 4096.0          # executed MFMAs only (exact-zero blocks skipped)
                other[name] = {'launches': len(evs), 'avg_launch_ms': tms / len(evs)}
                if tfl > 0:
                    other[name].update({'executed_tflops': tfl / (tms * 1e-3) / 1e12 if tms > 0 else 0.0,
                                        'executed_gflop_per_launch': tfl / len(evs) / 1e9})
        out['other_kernels'] = other
        stages = {}
        for name, evs in timers.items():
            if name.startswith('hbm:') and evs:
                tms = sum(s.elapsed_time(e) for s, e, _ in evs)
                tb = sum(b for _, _, b in evs)
                stages[name[4:]] = {'launches': len(evs), 'avg_ms': tms / len(evs),
                                    'algorithmic_GBps': tb / (tms * 1e-3) / 1e9 if tms > 0 else 0.0,
                                    'frac_of_8TBps': tb / (tms * 1e-3) / 8e12 if tms > 0 else 0.0}
        out['hbm_stages'] = stages
        if world == 1 and not args.no_cpu_baseline:
            del model, batch
            torch.cuda.empty_cache()
            out['cpu_baseline'] = cpu_baseline(args.points)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
