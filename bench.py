#!/usr/bin/env python3
"""Headline benchmark: KITTI-shaped frames/s through the MVXNet hot path
(crop + projection + voxelize + fusion + VFE + dense 3-D conv, forward + backward) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step (--mode hot, the headline) = one batch of `--frames` synthetic frames PER GPU (weak scaling):
  raw 120,000-point clouds resident in HBM -> crop + cropToSight + lidar2Img (one compaction pass) -> batched
  voxelizer -> ONE launch per layer for all frames (modules/frames.py): fusion sampling + fusion MLP + VFE stack +
  reindex + CML forward and the full backward (dL/d(middle) is a fixed resident tensor standing for RPN + loss) ->
  one flat gradient all-reduce over RCCL -> one AdamW step.
Other modes (BASELINE.json configs): --mode fusion (config 4: FPN feature sampling + fusion MLP + VFE, 2 frames),
--mode vfe (config 2: voxelize + VFE stack forward/backward, 16 frames, voxel
indices asserted bit-exact against the C oracle inside the run), --mode dropin (the nn.Module API: MVXNet.forward +
VoxelLoss + autograd + AdamW, one frame at a time like train.py:110-164: compact rows, RPN as one HIP node), --mode full (the WHOLE model
of train.py:110-164 for B frames per step on this library's kernels: classifyAnchors, frame sets through fusion / VFE /
CML, the RPN on the same gather kernels, VoxelLoss, whole backward).
--workload S1|S2: uniform worst case (V ~ 19.9 k voxels per frame) or the KITTI-like ring model (V ~ 5 k, default).
Prints ONE JSON line on rank 0.
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
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s
RAW_POINTS = 120000                # raw (un-cropped) clouds of ~120 k points


def host_projection(pts, calib):
    """lidar2Img(uncheck=True) + swap to (row, col) in f32 (train.py:31-33), host numpy (CPU baseline inputs)."""
    p = np.ones((4, pts.shape[0]), np.float32)
    p[:3] = pts[:, :3].T
    m = (calib['R0_rect'].astype(np.float32) @ calib['Tr_velo_to_cam'].astype(np.float32))
    img = calib['P2'].astype(np.float32) @ (m @ p)
    uv = (img[:2] / img[2]).T
    return uv[:, ::-1].astype(np.float32)


_FRAME_CACHE = {}


def frame_points(workload, fid, P):
    """The P points of a synthetic frame that survive crop + cropToSight, in cloud order (cached: the runs of one process
    share their frames)."""
    from modules.data import Synthetic as S
    key = (workload, fid, P)
    if key not in _FRAME_CACHE:
        _FRAME_CACHE[key] = S.synth_ring(fid, P) if workload == 'S2' else S.synth_uniform_in_sight(fid, P)
    return _FRAME_CACHE[key]


def make_batch(frame_ids, dev, P, workload='S2', raw_points=RAW_POINTS):
    """Resident inputs of one step: raw clouds (kept points interleaved with points that fail the crops), shuffle
    permutations of the kept points, FPN maps."""
    from modules.data import Synthetic as S
    from modules.pipeline import FrameBatch
    B = len(frame_ids)
    raw = np.zeros((B, raw_points, 4), np.float32)
    perms = np.zeros((B, P), np.int32)
    fpn = []
    for k, fid in enumerate(frame_ids):
        pc = frame_points(workload, fid, P)
        assert pc.shape[0] == P, 'synthetic frame %d has %d points' % (fid, pc.shape[0])
        raw[k] = S.synth_raw_around(pc, fid, raw_points)
        perms[k] = S.synth_perm(fid, P)
        g = torch.Generator(device='cpu').manual_seed(3000 + fid)
        fpn.append([torch.randn((1, 256, h, w), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
                    for h, w in ((104, 336), (52, 168), (26, 84))])
    return FrameBatch(None, torch.from_numpy(perms).to(dev), None, fpn, raw=torch.from_numpy(raw).to(dev),
                      calib=S.KITTI_CALIB, cap_points=P)


def synthetic_gt():
    """Eight 'Car' boxes (x y z l w h r, LiDAR frame) inside the range: the targets of --mode full / dropin."""
    gg = np.random.default_rng(11)
    n = 8
    gt = np.stack([gg.uniform(8, 60, n), gg.uniform(-30, 30, n), gg.uniform(-1.8, -0.6, n), gg.uniform(3.4, 4.4, n),
                   gg.uniform(1.5, 1.8, n), gg.uniform(1.4, 1.7, n), gg.choice([0.0, np.pi / 2], n) + gg.normal(0, 0.05, n)], 1)
    return torch.tensor(gt, dtype=torch.float32)


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


def cpu_baseline(P, workload, budget_s=75.0, with_rpn=False):
    """The CPU oracle (plain-C voxelizer + torch-CPU / oneDNN fusion, VFE, CML forward + backward, dense as the reference
    computes it) on a BOUNDED sample of the same workload: warm-up frames, then up to 8 timed frames within the time budget
    (at least 3); medians per stage.  All host threads the cgroup allows, and 8 threads (the size of
    the build container).  A reported baseline, not the target.  ``with_rpn`` (--mode full): the C
    classifyAnchors + torch-CPU RPN + VoxelLoss forward and backward in place of the fixed dL/d(BEV map)."""
    import ctypes
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import mvx_oracle as O
    from modules.data import Synthetic as S
    lib = ctypes.CDLL(os.path.join(REPO, 'oracle', 'liboracle_c.so'))
    lib.oracle_group9.restype = ctypes.c_int64
    G = torch.ones((1, 128, O.VOXELSHAPE[0], O.VOXELSHAPE[1]))
    rng = np.asarray(O.VELORANGE, np.float64)
    size = np.asarray(O.voxelsize(), np.float64)
    if with_rpn:
        Prpn = {k: v.requires_grad_(True) for k, v in O.rpn_params(np.load(os.path.join(REPO, 'tests', 'golden', 'rpn_shapes.npz'))).items()}
        anchors = O.create_anchors(O.VOXELSHAPE[0] // 2, O.VOXELSHAPE[1] // 2)
        abev = O.bbox3d2bev(anchors.reshape(anchors.shape[:2] + (-1, 7)))
        gt = synthetic_gt()
        gbev = O.bbox3d2bev(gt)

    def rpn_loss(mid):
        """dL/d(mid) of clsLoss + regLoss through the RPN (train.py:131-161)."""
        leaf_m = mid.detach().requires_grad_(True)
        pi, ni, gi = O.classify_anchors(gbev, gt[:, [0, 1]], abev, O.VELORANGE, 0.45, 0.6)
        score, reg = O.rpn(leaf_m, Prpn)
        cls, rl = O.voxel_loss(pi, ni, gi, gt, score[0].permute(1, 2, 0), reg[0].permute(1, 2, 0), anchors, 2)
        (cls if rl is None else cls + rl).backward()
        return leaf_m.grad

    def one_frame(fid, P_):
        pc = frame_points(workload, fid, P)
        pcd = np.ascontiguousarray(np.concatenate([pc, host_projection(pc, S.KITTI_CALIB)], 1), np.float32)
        perm = S.synth_perm(fid, pcd.shape[0])
        feats = [torch.from_numpy(f) for f in S.synth_fpn(fid)]
        Pn = pcd.shape[0]
        t0 = time.perf_counter()
        voxel = np.empty((Pn, 35, 9), np.float64)
        uidx = np.empty((Pn, 3), np.float64)
        cnt = np.empty(Pn, np.int64)
        V = lib.oracle_group9(pcd.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(6), perm.ctypes.data_as(ctypes.c_void_p),
                              ctypes.c_int64(Pn), rng.ctypes.data_as(ctypes.c_void_p), size.ctypes.data_as(ctypes.c_void_p),
                              ctypes.c_int32(35), voxel.ctypes.data_as(ctypes.c_void_p),
                              uidx.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
        vox = torch.from_numpy(voxel[:V].astype(np.float32))
        idx = torch.from_numpy(np.concatenate([np.zeros((V, 1)), uidx[:V]], 1).astype(np.int64))
        t1 = time.perf_counter()
        v23 = O.mvx_point_features(vox, feats, torch.tensor([370.0, 1224.0]), P_)
        feat = O.voxel_features(v23, O.strip_prefix(P_, 'backbone.'))
        t2 = time.perf_counter()
        leaf = feat.detach().requires_grad_(True)
        mid = O.cml(O.reindex(leaf, idx), O.strip_prefix(P_, 'backbone.')).reshape(1, -1, O.VOXELSHAPE[0], O.VOXELSHAPE[1])
        tr0 = time.perf_counter()
        g_mid = rpn_loss(mid) if with_rpn else G
        tr1 = time.perf_counter()
        mid.backward(g_mid)
        t3 = time.perf_counter()
        feat.backward(leaf.grad)
        t4 = time.perf_counter()
        return int(V), (t1 - t0, (t2 - t1) + (t4 - t3), (t3 - t2) - (tr1 - tr0), t4 - t0, tr1 - tr0)

    out = {}
    all_thr = host_threads()
    # 2 warm-up + 8 timed frames with all host threads (about a minute of CPU work on the GPU box), a shorter
    # run with 8 threads for comparison with the build container
    for label, n, warm, timed in (('all', all_thr, 2, 8), ('n8', min(8, all_thr), 1, 3)):
        if label == 'n8' and n == all_thr:
            out['n8'] = out['all']
            continue
        torch.set_num_threads(n)
        P_ = {k: v.requires_grad_(True) for k, v in O.make_params(7).items()}
        for wu in range(warm):
            one_frame(wu, P_)                           # warm-up (oneDNN primitive creation, page faults)
        t_begin, times, nv, fid = time.perf_counter(), [], [], warm
        while len(times) < 3 or (len(times) < timed and time.perf_counter() - t_begin < budget_s):
            v, t = one_frame(fid, P_)
            times.append(t)
            nv.append(v)
            fid += 1
        med = np.median(np.asarray(times), axis=0)
        out[label] = {'threads': n, 'frames_timed': len(times), 'voxels': nv,
                      'median_s': {'voxelize': float(med[0]), 'fusion_vfe_fwd_bwd': float(med[1]),
                                   'scatter_cml_fwd_bwd': float(med[2]), 'frame': float(med[3]),
                                   'anchors_rpn_loss_fwd_bwd': float(med[4])},
                      'frames_per_s': float(1.0 / med[3])}
    a = out['all']
    return {'value': a['frames_per_s'], 'unit': 'frames/s', 'cores': a['threads'], 'kind': 'port',
            'sample': '%s frames, %d pts: 2 warm-up + %d timed frames on %d threads (median %.2f s/frame: C voxelizer %.3f, '
                      'torch-CPU fusion+VFE fwd+bwd %.2f, reindex+CML fwd+bwd %.2f, C classifyAnchors + torch-CPU RPN + VoxelLoss '
                      'fwd+bwd %.2f%s); and %d timed frames on %d threads'
                      % (workload, P, a['frames_timed'], a['threads'], a['median_s']['frame'], a['median_s']['voxelize'],
                         a['median_s']['fusion_vfe_fwd_bwd'], a['median_s']['scatter_cml_fwd_bwd'],
                         a['median_s']['anchors_rpn_loss_fwd_bwd'], '' if with_rpn else ' (not part of this mode)',
                         out['n8']['frames_timed'], out['n8']['threads']),
            'runs': out}


def _oracle_group(pcd6, perm):
    import ctypes
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import mvx_oracle as O
    lib = ctypes.CDLL(os.path.join(REPO, 'oracle', 'liboracle_c.so'))
    lib.oracle_group9.restype = ctypes.c_int64
    rng = np.asarray(O.VELORANGE, np.float64)
    size = np.asarray(O.voxelsize(), np.float64)
    Pn = pcd6.shape[0]
    voxel = np.empty((Pn, 35, 9), np.float64)
    uidx = np.empty((Pn, 3), np.float64)
    cnt = np.empty(Pn, np.int64)
    V = lib.oracle_group9(pcd6.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(6), perm.ctypes.data_as(ctypes.c_void_p),
                          ctypes.c_int64(Pn), rng.ctypes.data_as(ctypes.c_void_p), size.ctypes.data_as(ctypes.c_void_p),
                          ctypes.c_int32(35), voxel.ctypes.data_as(ctypes.c_void_p), uidx.ctypes.data_as(ctypes.c_void_p),
                          cnt.ctypes.data_as(ctypes.c_void_p))
    return voxel[:V], uidx[:V]


def voxel_index_check(batch, workload, frame_ids, P):
    """CPU-baseline leg of --mode vfe: GPU crop + projection + voxelizer of every frame of the batch against the C oracle:
    the cropped clouds, the voxel indices (order included) and the voxel payload must be bit-identical."""
    import modules.config as cfg
    from modules import _hip
    from modules.data import Synthetic as S
    points6, n_points = batch.prepared()
    res = _hip.voxelize(points6, batch.perms, n_points, cfg.velorange[0:3], cfg.voxelsize, 35, 9)
    n = n_points.tolist()
    for k, fid in enumerate(frame_ids):
        pc = frame_points(workload, fid, P)
        assert n[k] == P and np.array_equal(points6[k, :P, :4].cpu().numpy(), pc), 'GPU crop differs from the generator'
        ref6 = np.ascontiguousarray(np.concatenate([pc, host_projection(pc, S.KITTI_CALIB)], 1), np.float32)
        got6 = points6[k, :P].cpu().numpy()
        assert np.abs(got6[:, 4:] - ref6[:, 4:]).max() < 1e-2, 'GPU projection differs'
        rv, ri = _oracle_group(np.ascontiguousarray(got6), S.synth_perm(fid, P))
        V = int(res.n_voxels[k])
        assert V == rv.shape[0], 'voxel count differs from the oracle'
        assert np.array_equal(res.coords[k, :V, 1:].cpu().numpy(), ri.astype(np.int64)), 'voxel indices differ from the oracle'
        assert np.array_equal(res.voxels[k, :V].cpu().numpy(), rv.astype(np.float32)), 'voxel payload differs from the oracle'
    return 'ok, %d frames' % len(frame_ids)


def cpu_baseline_vfe(P, workload, budget_s=20.0):
    """CPU leg of --mode vfe: C voxelizer + torch-CPU SVFE + FCN + max forward/backward (dense (V,35,23) rows as the
    reference computes them), all host threads."""
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import mvx_oracle as O
    from modules.data import Synthetic as S
    n = host_threads()
    torch.set_num_threads(n)
    P_ = {k: v.requires_grad_(True) for k, v in O.strip_prefix(O.make_params(7), 'backbone.').items()}
    times = []
    t_begin = time.perf_counter()
    fid = 0
    while len(times) < 4 or (time.perf_counter() - t_begin < budget_s and len(times) < 17):
        pc = frame_points(workload, fid, P)
        pcd = np.ascontiguousarray(np.concatenate([pc, host_projection(pc, S.KITTI_CALIB)], 1), np.float32)
        t0 = time.perf_counter()
        rv, _ = _oracle_group(pcd, S.synth_perm(fid, P))
        vox = torch.from_numpy(rv.astype(np.float32))
        pad = (vox[..., :3] == 0).all(-1)
        vox[pad] = 0
        x = torch.cat([vox[..., :7], torch.randn(vox.shape[0], 35, 16)], dim=-1)
        feat = O.voxel_features(x, P_)
        feat.backward(torch.ones_like(feat) * 1e-3)
        times.append(time.perf_counter() - t0)
        fid += 1
    med = float(np.median(times[1:]))
    return {'value': 1.0 / med, 'unit': 'frames/s', 'cores': n, 'kind': 'port',
            'sample': '%s frames, %d pts: 1 warm-up + %d timed frames, C voxelizer + torch-CPU SVFE+FCN+max fwd+bwd on dense '
                      '(V,35,23) rows, median %.3f s/frame' % (workload, P, len(times) - 1, med)}


ALT_STEPS = 20            # timed steps of every alternative mode of the default run
MATH_PIECES = {'f32': 0, 'bf16x3': 2, 'bf16x6': 3, 'fp16x3': 4}   # split code of modules/_hip.py (4 = two fp16 pieces)
MATH_MFMAS = {'f32': 1.0, 'bf16x3': 3.0, 'bf16x6': 6.0, 'fp16x3': 3.0}   # 16-bit MFMAs per product (executed matrix FLOPs = this x the algorithmic ones)


def isolated_conv_roofline(dev, math):
    """The dominant kernel's launches (conv2/conv3 forward + dgrad, dense) on an otherwise idle GPU."""
    from modules import _hip
    import modules.config as cfg
    H, W = cfg.voxelshape[0], cfg.voxelshape[1]
    split = MATH_PIECES[math]
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
    mult, peak = (MATH_MFMAS[math], BF16_MFMA_PEAK_TFLOPS) if split else (1.0, FP32_MFMA_PEAK_TFLOPS)
    ach = mult * tot_fl / (tot_ms * 1e-3) / 1e12
    return {'achieved': ach, 'frac': ach / peak, 'avg_launch_ms': tot_ms / 4,
            'note': 'the same kernel, one frame, dense (no tile skipping), alone on the GPU'}


LINE_LIMIT = 12288        # the ONE line rank 0 prints stays below this (the driver keeps a bounded tail of stdout and parses it)


def write_detail(out):
    """Everything the run measured (per-stage GB/s, every kernel class of every alternative mode, the CPU baseline's per-run
    medians, the notes) goes to gpurun_out/bench_detail.json (MVX_BENCH_DETAIL overrides the path); the printed line names it."""
    path = os.environ.get('MVX_BENCH_DETAIL', os.path.join(REPO, 'gpurun_out', 'bench_detail.json'))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as fh:
            json.dump(out, fh, indent=1)
        return os.path.relpath(path, REPO)
    except OSError as e:                                   # a read-only checkout: the line still prints
        return 'not written (%s)' % e.__class__.__name__


def _r(x, n=4):
    return None if x is None else (float('%.*g' % (n, x)) if isinstance(x, float) else x)


def _roof(r):
    """The roofline object of the line: the contract's keys + the kernel it is about, without the prose."""
    keep = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'launches', 'avg_launch_ms', 'flop_per_launch',
            'algorithmic_tflops', 'dense_equivalent_tflops')
    o = {k: (r[k] if k in ('achieved', 'peak', 'frac') else _r(r[k], 5)) for k in keep if k in r}
    o['kernel'] = str(o.get('kernel', '')).split(' (')[0]
    if isinstance(r.get('isolated'), dict):
        o['isolated_frac'] = _r(r['isolated']['frac'])
    return o


def compact_line(out, detail_path):
    """The printed line: the bench contract's keys, `roofline` and `cpu_baseline` in full, and for every other mode measured in the
    same run only what identifies it and its result.  Everything else is in the detail file."""
    line = {k: out[k] for k in ('metric', 'value', 'unit', 'summary', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                'scaling', 'vs_baseline', 'dtype', 'data') if k in out}
    c = out['config']
    line['config'] = {k: c[k] for k in ('workload', 'mode', 'frames_per_gpu', 'voxels_per_frame', 'parallelism', 'voxel_indices_vs_oracle') if k in c}
    line['roofline'] = _roof(out['roofline'])
    if 'traffic_note' in out['roofline']:
        line['roofline']['traffic_note'] = out['roofline']['traffic_note'][:160]
    if 'cpu_baseline' in out:
        line['cpu_baseline'] = {k: out['cpu_baseline'][k] for k in ('value', 'unit', 'cores', 'kind', 'sample', 'voxel_indices_vs_oracle')
                                if k in out['cpu_baseline']}
    for k in ('host_enqueue_ms_per_step', 'ms_per_step_all_timers', 'timers', 'library_launches_per_step', 'allreduce_ms_per_call', 'last_losses', 'INVALID_diagnostic_knockout',
              'gradient_fingerprint'):
        if k in out:
            line[k] = out[k]
    line['other_kernels'] = {n: {'frac': _r(v.get('frac')), 'avg_launch_ms': _r(v['avg_launch_ms']), 'launches': v['launches'],
                                 'arithmetic': v.get('arithmetic')} for n, v in out.get('other_kernels', {}).items()}
    line['hbm_stages_frac_of_8TBps'] = {n: _r(v['frac_of_8TBps'], 3) for n, v in out.get('hbm_stages', {}).items()
                                        if v.get('frac_of_8TBps') is not None}
    alts = []
    for a in out.get('alt_modes', []):
        alts.append({'config': a.get('baseline_config', 'headline path'), 'mode': a.get('mode', out['config']['mode']),
                     'convmath': a['convmath'], 'workload': a['workload'], 'value': _r(a['value'], 5), 'unit': a['unit'],
                     'ms_per_step': _r(a['ms_per_step'], 5), 'steps': a.get('steps'), 'roofline_frac': _r(a['roofline']['frac']),
                     'roofline_kernel': str(a['roofline'].get('kernel', '')).split(' (')[0]})
    if alts:
        line['alt_modes'] = alts
    line['detail'] = detail_path
    if 'summary_tail' in out:
        line['summary_tail'] = out['summary_tail']
    return line


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mode', choices=['hot', 'vfe', 'fusion', 'dropin', 'full'], default='hot')
    ap.add_argument('--workload', choices=['S1', 'S2'], default='S2')
    ap.add_argument('--frames', type=int, default=None, help='frames per GPU per step (default 4; 16 in --mode vfe, 2 in --mode fusion)')
    ap.add_argument('--points', type=int, default=20000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--convmath', choices=['fp16x3', 'bf16x6', 'bf16x3', 'f32'], default=None, help='override config.yml convmath')
    ap.add_argument('--no-alt', action='store_true', help='skip the extra runs (other workload, bf16x3 arithmetic, the other BASELINE configs)')
    ap.add_argument('--frame-ids', default=None, help='comma-separated synthetic frame ids of THIS process (default: rank + world * j)')
    ap.add_argument('--timed-only', action='store_true',
                    help='only warm-up + the timed steps (no alt / isolated / CPU passes): what profiles/ is made from')
    args = ap.parse_args(argv)
    if args.timed_only:
        args.no_alt = args.no_cpu_baseline = True
    return args


def main():
    args = parse_args()
    from modules import parallel
    rank, world, local = parallel.init_from_env(os.environ.get('MVX_DIST_BACKEND'))
    assert world == args.gpus or world == 1, 'launch with torchrun --nproc-per-node = --gpus'
    dev = torch.device('cuda', local % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    out = run(args, rank, world, dev)
    if rank == 0:
        line = compact_line(out, write_detail(out))
        text = json.dumps(line)
        assert len(text) < LINE_LIMIT, 'bench line grew to %d bytes' % len(text)
        print(text)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run(args, rank, world, dev):
    """One benchmark run in this process: the timed region of ``args.mode`` plus (unless switched off) its alternative runs and
    the CPU baseline; returns the JSON object on rank 0 (None elsewhere).  Called again by itself for the other BASELINE
    configs that the default run appends under ``alt_modes``."""
    import modules.config as cfg
    from modules import _hip
    old_math = cfg.config.get('convmath', 'f32')
    try:
        return _run(args, rank, world, dev)
    finally:
        cfg.config['convmath'] = old_math
        _hip.KERNEL_TIMERS = None


def _run(args, rank, world, dev):
    from modules import parallel
    if args.frames is None:
        args.frames = {'vfe': 16, 'fusion': 2}.get(args.mode, 4)

    import modules.config as cfg
    from modules import _hip
    from modules import pipeline as pl
    from MVXNet import MVXNet
    if args.convmath:
        cfg.config['convmath'] = args.convmath
    main_math = cfg.config.get('convmath', 'f32')

    torch.manual_seed(0)
    model = MVXNet().to(dev)
    with_rpn = args.mode in ('dropin', 'full')
    hot = [p for k, p in model.named_parameters() if p.requires_grad and (with_rpn or '.rpn.' not in k)]
    # the weight gradient of the fusion MLP's first layer is the last kernel of a step: it goes out in the second, small part
    # of the exchange (modules/parallel.py)
    bucket = parallel.GradBucket(hot, late=[model.head.fusion.fcn1.fc.weight])
    bucket.timing = world > 1
    # train.py:64's optimizer; fused=True is the same update as ONE multi-tensor kernel instead of ~10 foreach launches with
    # Python between them (hot 348 -> 364, one-frame-at-a-time 88 -> 105 frames/s on the same box; MVX_FUSED_OPT=0 for the default)
    opt = torch.optim.AdamW(hot, lr=1e-3, eps=cfg.eps, fused=os.environ.get('MVX_FUSED_OPT', '1') == '1')
    frame_ids = [rank + world * j for j in range(args.frames)]
    if args.frame_ids:
        frame_ids = [int(v) for v in args.frame_ids.split(',')]
        assert len(frame_ids) == args.frames, '--frame-ids must name --frames ids'
    batch = make_batch(frame_ids, dev, args.points, args.workload)
    g = torch.Generator(device='cpu').manual_seed(77)
    grad_mid = (torch.randn((1, 128, cfg.voxelshape[0], cfg.voxelshape[1]), generator=g) * 1e-3).to(dev)
    imsize = [float(v) for v in cfg.imsize]            # host list: no device read-back inside the step
    frames_total = args.frames * world
    pending_status = []
    state = {'ready': None}

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------------------------------------ step bodies
    def step_hot(b=None):
        b = batch if b is None else b
        bucket.zero()
        fn = pl.train_step_frame_set if pl.BATCHED else pl.train_step_frames
        nv, statuses, state['ready'] = fn(model, b, grad_mid, imsize, ready=state['ready'], prepare_next=b)
        bucket.all_reduce_mean(frames_total)
        opt.step()
        pending_status.extend(statuses)
        return nv

    vfe_state = {}

    def step_vfe(b=None):
        """BASELINE config 2: voxelize + VFE stack (SVFE + FCN + max, voxelnet/Pipe.py:5-29, VoxelNet.py:27-33) forward and
        backward for all frames in one frame set; the 16 fused image channels and dL/d(voxel features) are resident random
        tensors standing for the branches that are not part of this configuration."""
        from modules import frames as fr
        bucket.zero()
        nv, statuses, state['ready'] = pl.train_step_rows_only(model, batch, vfe_state, ready=state['ready'], prepare_next=batch)
        bucket.all_reduce_mean(frames_total)
        opt.step()
        pending_status.extend(statuses)
        return nv

    def step_fusion(b=None):
        """BASELINE config 4: PointFusion on -- crop + projection (KITTI calibration), voxelize, FPN feature sampling + fusion
        MLP + VFE stack forward and backward, 2 frames per step; dL/d(voxel features) is a resident tensor."""
        bucket.zero()
        nv, statuses, state['ready'] = pl.train_step_rows_only(model, batch, vfe_state, ready=state['ready'], prepare_next=batch,
                                                               with_fusion=True, imsize=imsize)
        bucket.all_reduce_mean(frames_total)
        opt.step()
        pending_status.extend(statuses)
        return nv

    drop = {}

    def targets_setup():
        from modules import Calc
        from modules.data import Preprocessing as pre
        from modules.voxelnet import VoxelLoss
        if not drop:
            anchors = pre.createAnchors(cfg.voxelshape[0] // 2, cfg.voxelshape[1] // 2, cfg.velorange, cfg.carsize)
            drop['anchors'] = anchors.to(dev)
            drop['bevs'] = Calc.bbox3d2bev(anchors.reshape(anchors.shape[:2] + (-1, 7))).to(dev).contiguous()
            drop['gt'] = synthetic_gt()
            drop['gt_dev'] = drop['gt'].to(dev)
            drop['gt_bev'] = Calc.bbox3d2bev(drop['gt'])
            drop['crit'] = VoxelLoss()
            drop['imsize'] = torch.tensor(imsize, device=dev)
        return drop

    def step_dropin(b=None):
        """The reference's own interface, one frame at a time (train.py:110-164): preprocessing -> MVXNet.forward ->
        VoxelLoss -> backward -> AdamW.  The nn.Module API under autograd: MVXNet.forward evaluates fusion + VFE on compact rows,
        the first CML layer on the voxel rows, the RPN as one node on this library's kernels (modules/voxelnet/Pipe.py)."""
        from modules import Calc
        targets_setup()

        def prepare():
            """The CPU part of train.py's iteration (cputask, train.py:26-49) for the frames of a step: voxelization (with
            the compact-row maps) and target assignment, each with one host read for all frames."""
            fr_, st_ = pl.voxelize_batch(batch, with_maps=True)
            gt = drop['gt']
            tg_ = Calc.classifyAnchorsFrames([(drop['gt_bev'], gt[:, [0, 1]])] * len(fr_), drop['bevs'], cfg.velorange, 0.45, 0.6)
            return fr_, st_, tg_

        # ... done one step ahead on the preparation stream (the reference prepares its batches in worker processes while the
        # model runs, train.py:185-187): the three host reads then wait for a few small kernels instead of draining the
        # training stream at every step boundary; then the model one frame at a time
        main = torch.cuda.current_stream(dev)
        if drop.get('ready') is None:
            frames, st, tg = prepare()
        else:
            frames, st, tg, ev = drop['ready']
            main.wait_event(ev)
            st.record_stream(main)
            for v, idx in frames:
                v.record_stream(main)
                idx.record_stream(main)
                if getattr(v, '_mvx_fs', None) is not None:
                    v._mvx_fs.hand_over(main)
            for t3 in tg:
                for x in tuple(t3[0]) + tuple(t3[1]) + (t3[2],):
                    if isinstance(x, torch.Tensor) and x.is_cuda:
                        x.record_stream(main)
        statuses = [st]
        # the parameters' .grad are views of the flat bucket (GradBucket) and stay so (zero_grad(set_to_none=False)): let the
        # node add its gradients straight into them instead of handing 80 tensors back to the autograd engine
        old_sink, _hip.GRAD_SINK = _hip.GRAD_SINK, True
        try:
            for f, (voxels, idx) in enumerate(frames):
                opt.zero_grad(set_to_none=False)
                pi, ni, gi = tg[f]
                score, reg = model(voxels, batch.fpn_levels[f], idx, [None], drop['imsize'])
                score = score.squeeze(0).permute(1, 2, 0)
                reg = reg.squeeze(0).permute(1, 2, 0)
                cls_loss, reg_loss = drop['crit'](pi, ni, gi, drop['gt_dev'], score, reg, drop['anchors'], 2)
                loss = cls_loss if reg_loss is None else cls_loss + reg_loss
                loss.backward()
                opt.step()
        finally:
            _hip.GRAD_SINK = old_sink
        with torch.cuda.stream(pl._prep_stream(dev)):
            nxt = prepare()
            ev = torch.cuda.Event()
            ev.record()
        drop['ready'] = nxt + (ev,)
        from modules import whole
        pending_status.extend(statuses + whole.take_status(model))
        return [v.shape[1] for v, _ in frames]

    full = {}

    def full_targets():
        """train.py:44-46 for every frame of the batch: classifyAnchors on the GPU, all frames in one kernel pass and one host
        read of the list lengths (Calc.classifyAnchorsFrames)."""
        from modules import Calc
        d = targets_setup()
        res = Calc.classifyAnchorsFrames([(d['gt_bev'], d['gt'][:, [0, 1]])] * args.frames, d['bevs'], cfg.velorange, 0.45, 0.6)
        return [(pi, ni, gi, d['gt_dev']) for pi, ni, gi in res]

    def step_full(b=None):
        """The WHOLE model on this library's kernels (train.py:110-164 for B frames): crop + projection, voxelizer,
        classifyAnchors, fusion + VFE + CML as frame sets, RPN (modules/rpn_frames.py), VoxelLoss, everything back, one
        all-reduce, AdamW.  The losses of a step are read one step later (the host never waits for what it just enqueued)."""
        d = targets_setup()
        bucket.zero()
        if state['ready'] is None:
            ready, targets = None, full_targets()
        else:
            ready, targets = state['ready']
        out = pl.train_step_full(model, batch, targets, d['crit'], d['anchors'], imsize, ready=ready,
                                 prepare_next=(batch, full_targets), read=False)
        state['ready'] = out.pop('next')
        bucket.all_reduce_mean(frames_total)
        opt.step()
        if full.get('pending') is not None:
            t_r = time.perf_counter()
            full['last'] = pl.read_losses(full['pending'])['loss']          # waits for the PREVIOUS step's kernels
            full['read_s'] = full.get('read_s', 0.0) + time.perf_counter() - t_r
        full['pending'] = out
        return out['voxels']

    def gradient_fingerprint():
        """MVX_BENCH_FINGERPRINT=1 (tests of the N > 1 path): one hot step WITHOUT the optimizer, then two linear functionals of
        the exchanged gradient bucket (its sum and its product with a fixed probe vector) in float64.  Linear, so the bucket of a
        2-rank run must equal the mean of the buckets of its ranks' frames run alone (tests/test_bench_gpu.py)."""
        bucket.zero()
        fn = pl.train_step_frame_set if pl.BATCHED else pl.train_step_frames
        _, statuses = fn(model, batch, grad_mid, imsize)
        bucket.all_reduce_mean(frames_total)
        torch.cuda.synchronize()
        flat = bucket.flat.double()
        probe = torch.cos(torch.arange(flat.numel(), device=dev, dtype=torch.float64) * 0.37)
        return {'sum': float(flat.sum()), 'probe': float((flat * probe).sum()), 'norm': float(flat.norm()),
                'exchange': list(bucket.calls)[-1] if bucket.calls else None, 'frame_ids': frame_ids}

    fingerprint = gradient_fingerprint() if (os.environ.get('MVX_BENCH_FINGERPRINT') and args.mode == 'hot') else None
    step = {'hot': step_hot, 'vfe': step_vfe, 'fusion': step_fusion, 'dropin': step_dropin, 'full': step_full}[args.mode]
    host_ms, exec_stages, launches = [], [], []

    # Timers of the timed region: the HEADLINE run records HIP events only around the kernel its roofline is quoted on (two event
    # records per library call are ~180 markers per hot step and cost 0.5 ms of it: tools/soak_plain.py, the same loop without
    # any timer, runs 7.65 ms where the fully instrumented step takes 8.2); `other_kernels` and `hbm_stages` come from a second,
    # shorter pass with every timer on, whose own step time is reported beside them (`ms_per_step_all_timers`).
    ROOF_TIMERS = {'hot': ('conv3d_gather_bg', 'conv3d_gather_tiles', 'conv3d_gather'), 'fusion': ('hbm:feature_sample',),
                   'vfe': ('hbm:voxelize',)}
    ROOF_TIMERS['full'] = ROOF_TIMERS['dropin'] = ROOF_TIMERS['hot']

    def timed_run(warmup, steps, fn=None, names=None):
        fn = step if fn is None else fn
        nv = None
        _hip.TIMER_NAMES = frozenset(names) if names is not None else None
        _hip.KERNEL_TIMERS = {}                           # the warm-up also fills the pool of timing events
        for _ in range(warmup):
            nv = fn()
        fence()
        _hip.recycle_timing_events(_hip.KERNEL_TIMERS)
        gc.collect()
        _hip.KERNEL_TIMERS = {}
        if _hip.EXEC_STAGES is not None:
            _hip.EXEC_STAGES.zero_()
        l0 = _hip.X.lib.mvx_launch_count()
        prof = None
        if os.environ.get('MVX_BENCH_CPROFILE'):          # developer aid: where the host side of the timed steps goes
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t0 = time.perf_counter()
        for _ in range(steps):
            nv = fn()
        host_dt = time.perf_counter() - t0
        if prof is not None:
            import io
            import pstats
            prof.disable()
            buf = io.StringIO()
            pstats.Stats(prof, stream=buf).sort_stats('tottime').print_stats(60)
            with open(os.environ['MVX_BENCH_CPROFILE'], 'w') as fh:
                fh.write('INVALID as a timing: cProfile was on.  %d steps\n' % steps + buf.getvalue())
        fence()
        dt_ = time.perf_counter() - t0
        launches.append((_hip.X.lib.mvx_launch_count() - l0) / steps)
        host_ms.append(host_dt / steps * 1e3)
        tm, _hip.KERNEL_TIMERS = _hip.KERNEL_TIMERS, None
        _hip.TIMER_NAMES = None
        exec_stages.append(int(_hip.EXEC_STAGES.item()) if _hip.EXEC_STAGES is not None else 0)
        if world > 1:
            t = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t)
        return nv, dt_, tm

    def check_status():
        bad = 0
        if pending_status:
            for v in torch.stack([t.reshape(()) for t in pending_status]).tolist():       # ONE host read
                bad |= int(v)
        _hip.raise_on_status(bad)
        del pending_status[:]

    def conv_roofline(tm, run, math=None):
        """Dominant kernel conv3d_gather_pw (exact f32) / conv3d_gather_split (bf16x3): conv2/conv3 forward (background-aware)
        + tile-restricted dgrad launches.  achieved = EXECUTED matrix FLOPs (the kernel's own stage counter x 4.72 MFLOP;
        skipped tiles are not credited; x 3 MFMAs per product in the bf16x3 arithmetic) / sum of the launch durations from
        HIP events recorded on the launch stream inside the timed region, against the peak of the MFMA type that ran."""
        math = main_math if math is None else math
        ev = tm.get('conv3d_gather_bg', []) + tm.get('conv3d_gather_tiles', []) + tm.get('conv3d_gather', [])
        ms = sum(s.elapsed_time(e) for s, e, _ in ev)
        dense_fl = sum(_hip.timer_value(f) for _, _, f in ev)
        fl = float(exec_stages[run]) * _hip.STAGE_FLOP + sum(_hip.timer_value(f) for _, _, f in tm.get('conv3d_gather', []))
        alg = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        r = {'bound': 'mfma', 'achieved': alg, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': alg / FP32_MFMA_PEAK_TFLOPS,
             'traffic': None, 'kernel': 'conv3d_gather_pw (conv2 / conv3 forward + dgrad of all frames of the step: 4 launches per step)',
             'launches': len(ev), 'avg_launch_ms': ms / max(1, len(ev)), 'flop_per_launch': fl / max(1, len(ev)),
             'dense_equivalent_tflops': dense_fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
             'note': 'exact f32 MFMA (v_mfma_f32_32x32x2_f32); FLOPs are the EXECUTED ones counted by the kernel: tiles that hold '
                     'only the voxel-free background, depth taps whose source halo holds no voxel and idle image-border tiles are '
                     'written from constants and not credited, while the time of filling them (about 60 % of the tiles of a launch) '
                     'stays in the denominator; measured while the side-stream weight-gradient kernels share the CUs; `isolated` '
                     '= the same kernel on dense launches alone'}
        if math != 'f32':
            # every product is three (bf16x3) or six (bf16x6) bf16 MFMAs: executed matrix FLOPs = that multiple of the algorithmic ones
            mm = MATH_MFMAS[math]
            r.update(achieved=mm * alg, peak=BF16_MFMA_PEAK_TFLOPS, frac=mm * alg / BF16_MFMA_PEAK_TFLOPS, algorithmic_tflops=alg,
                     flop_per_launch=mm * r['flop_per_launch'],
                     kernel='conv3d_gather_splitT (conv2 / conv3 forward + dgrad of all frames of the step)',
                     note='%s split MFMA (v_mfma_f32_32x32x16_{bf16,f16}, f32 accumulate), EXECUTED stages counted by the kernel '
                          'x 4.72 MFLOP x %d MFMAs per product, against the dense bf16 / fp16 peak; skipped background tiles are not credited;'
                          ' algorithmic_tflops = the same work priced as one multiply-add per product' % (math, int(mm)))
        return r

    def hbm_stages(tm):
        out = {}
        for name, evs in tm.items():
            if name.startswith('hbm:') and evs:
                tms = sum(s.elapsed_time(e) for s, e, _ in evs)
                tb = sum(_hip.timer_value(b) for _, _, b in evs)       # data-dependent byte counts are evaluated here, after the run
                out[name[4:]] = {'launches': len(evs), 'avg_ms': tms / len(evs),
                                 'algorithmic_GBps': tb / (tms * 1e-3) / 1e9 if tms > 0 and tb > 0 else None,
                                 'frac_of_8TBps': tb / (tms * 1e-3) / 8e12 if tms > 0 and tb > 0 else None}
        return out

    # ------------------------------------------------------------------------------------------------ the timed run
    vfe_check = None
    if args.mode == 'vfe' and rank == 0 and not args.timed_only:
        # BASELINE config 2: the voxel indices of every frame of the run are checked bit-exact against the C oracle (part
        # of the CPU-baseline leg: the only place the benchmark touches oracle/)
        vfe_check = voxel_index_check(batch, args.workload, frame_ids, args.points)
    nvox, dt, timers = timed_run(args.warmup, args.steps, names=ROOF_TIMERS[args.mode])
    check_status()
    n_detail = min(args.steps, 10)
    _, dt_detail, timers_all = timed_run(2, n_detail)     # every timer on: other_kernels, hbm_stages
    check_status()

    def host_probe(n=3):
        """Host time to enqueue one step when the launch queues are empty (`n` steps from an idle GPU, no sync inside): what
        the Python side costs.  Inside the timed region the host runs ahead until the HIP queues are full, so the time it
        spends there mostly measures the GPU."""
        fence()
        full['read_s'] = 0.0
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        # --mode full reads the previous step's losses inside a step: that read waits for the GPU, not for Python, and is
        # taken out of the figure
        d = (time.perf_counter() - t0 - full.get('read_s', 0.0)) / n * 1e3
        fence()
        return d
    host_probe_ms = host_probe()
    check_status()

    alt = []
    if not args.no_alt and args.mode == 'hot':
        # (1) the other workload on the same code path
        other = 'S1' if args.workload == 'S2' else 'S2'
        b2 = make_batch(frame_ids, dev, args.points, other)
        state['ready'] = None
        n_alt = max(ALT_STEPS, args.steps) if args.steps >= 5 else args.steps      # alternative modes are timed over >= 20 steps
        nv2, dt2, tm2 = timed_run(3, n_alt, lambda: step_hot(b2), names=ROOF_TIMERS['hot'])
        check_status()
        alt.append({'workload': other, 'convmath': main_math, 'value': frames_total * n_alt / dt2, 'unit': 'frames/s', 'steps': n_alt,
                    'ms_per_step': dt2 / n_alt * 1e3, 'voxels_per_frame': nv2,
                    'roofline': conv_roofline(tm2, len(exec_stages) - 1), 'hbm_stages': hbm_stages(tm2)})
        del b2
        state['ready'] = None
        # (2) the same step in the other arithmetics (config.yml convmath: exact-f32 MFMA, bf16x6 = three bf16 pieces / six MFMAs per
        # product = fp32-grade, bf16x3 = two pieces / three MFMAs, ~2e-5 per product)
        for math in ('fp16x3', 'f32', 'bf16x6', 'bf16x3'):
            if math == main_math:
                continue
            cfg.config['convmath'] = math
            try:
                _, dt3, tm3 = timed_run(3, n_alt, names=ROOF_TIMERS['hot'])
            finally:
                cfg.config['convmath'] = main_math
                state['ready'] = None
            check_status()
            r3 = conv_roofline(tm3, len(exec_stages) - 1, math)
            alt.append({'workload': args.workload, 'convmath': math, 'value': frames_total * n_alt / dt3, 'steps': n_alt,
                        'unit': 'frames/s', 'ms_per_step': dt3 / n_alt * 1e3, 'roofline': r3,
                        'note': {'f32': 'every MFMA kernel on the exact-f32 matrix instruction (v_mfma_f32_32x32x2_f32)',
                                 'fp16x3': 'convolutions and wide row GEMMs, forward and both gradients, in two fp16 pieces per operand / three '
                                           'MFMAs per product (22 mantissa bits, fp32-grade), gradients scaled by their device-side amax',
                                 'bf16x6': 'convolutions and wide row GEMMs, forward and both gradients, in three-piece split arithmetic',
                                 'bf16x3': 'convolutions, row-GEMM gradients and the RPN GEMMs in two-piece split arithmetic (forward maps '
                                           'within 1e-5 of exact f32 on the CML; gradients carry 2e-5 per product)'}[math]})

    if not args.no_alt and args.mode == 'hot' and world == 1:
        # (3) the other BASELINE.json configs as short runs of their own modes, so that the driver's default invocation
        # carries a timed number and a roofline for each of them: config 2 (--mode vfe, 16 frames), config 4 (--mode fusion,
        # 2 frames), config 3 (--mode full, 4 frames: exact f32 and "bf16 MFMA conv" = convmath bf16x3)
        import copy
        for cfg_no, mode, math in ((2, 'vfe', main_math), (4, 'fusion', main_math), (3, 'full', main_math)) + tuple(
                (3, 'full', m) for m in ('fp16x3', 'f32', 'bf16x6', 'bf16x3') if m != main_math):
            a2 = copy.copy(args)
            a2.mode, a2.convmath, a2.frames = mode, math, None
            # short steps: enough of them to be out of the warm-up's shadow (allocator, clocks) at a few tenths of a second each
            a2.steps, a2.warmup = ((60, 15) if mode in ('vfe', 'fusion') else (ALT_STEPS, 4)) if args.steps >= 5 else (args.steps, 1)
            a2.no_alt = a2.no_cpu_baseline = True
            a2.timed_only = False
            state['ready'] = None
            o2 = run(a2, rank, world, dev)
            alt.append({'baseline_config': cfg_no, 'mode': mode, 'convmath': math, 'workload': args.workload, 'metric': o2['metric'],
                        'value': o2['value'], 'unit': o2['unit'], 'steps': o2['steps'], 'warmup': o2['warmup'],
                        'ms_per_step': o2['ms_per_step'], 'frames_per_gpu': o2['config']['frames_per_gpu'],
                        'host_enqueue_ms_per_step': o2['host_enqueue_ms_per_step'], 'roofline': o2['roofline'],
                        'other_kernels': o2.get('other_kernels', {}), 'config': o2['config']['workload']})
            gc.collect()
            torch.cuda.empty_cache()
        state['ready'] = None

    if rank == 0:
        dtype = {'f32': 'f32', 'bf16x6': 'f32 (bf16x6: three bf16 pieces = the f32 operand exactly, six MFMAs per product, f32 accumulate)',
                 'fp16x3': 'f32 storage / accumulate, 22-bit operands (fp16x3: two fp16 pieces, three MFMAs per product)',
                 'bf16x3': 'f32 storage / accumulate, 16-bit operands (bf16x3: two bf16 pieces, three MFMAs per product)'}[main_math]
        wl = {'S2': 'S2 ring frames (64-beam model, KITTI-like occupancy)', 'S1': 'S1 uniform frames (worst-case voxel count)'}[args.workload]
        if args.mode == 'hot':
            workload = ('%s, %d raw pts -> %d pts after crop, grid 10x352x400, T=35, %d frames/GPU/step: crop+cropToSight+lidar2Img, '
                        'voxelize, fusion sampling+MLP, VFE, reindex+CML fwd+bwd, all-reduce, AdamW; convmath=%s'
                        % (wl, RAW_POINTS, args.points, args.frames, main_math))
            roof = conv_roofline(timers, 0)
            if not args.timed_only:
                roof['isolated'] = isolated_conv_roofline(dev, main_math)
            # HBM bytes per launch from the PMC passes (profiles/traffic.json, written by tools/pmc_summary.py): printed only when
            # the file was measured on the SAME kernel sources and arithmetic that ran here (hash of the .hip files)
            tpath = os.path.join(REPO, 'profiles', 'traffic.json')
            if os.path.exists(tpath):
                import hashlib
                with open(tpath) as fh:
                    tj = json.load(fh)
                hsh = hashlib.sha256()
                for f in tj.get('source_files', []):
                    with open(os.path.join(REPO, 'mvxnet-makise_amd', 'csrc', f), 'rb') as fh:
                        hsh.update(fh.read())
                same_kernel = tj.get('convmath') == main_math
                if tj.get('source_sha16') == hsh.hexdigest()[:16] and same_kernel:
                    roof['traffic'] = tj['hbm_bytes_per_launch']
                    roof['traffic_note'] = 'PMC FETCH_SIZE x2 + WRITE_SIZE of %s, %s' % (tj['kernel'], tj.get('from', 'profiles/'))
                else:
                    roof['traffic_note'] = 'profiles/traffic.json was measured on other kernel sources / arithmetic: not printed'
            metric = 'KITTI frames/sec (voxelize+VFE+fusion+3Dconv fwd+bwd)'
        elif args.mode == 'fusion':
            workload = ('%s, %d raw pts -> %d pts, T=35, %d frames/GPU/step: crop+cropToSight+lidar2Img (KITTI 2011_09_26 calibration), '
                        'voxelize, FPN feature sampling (3 levels x 256 ch) + fusion MLP + VFE stack fwd+bwd, AdamW'
                        % (wl, RAW_POINTS, args.points, args.frames))
            ev = timers.get('hbm:feature_sample', [])
            ms = sum(s.elapsed_time(e) for s, e, _ in ev)
            tb = sum(_hip.timer_value(b) for _, _, b in ev)
            ach = tb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            roof = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBPS, 'traffic': None,
                    'kernel': 'feature_sample_rows (bilinear FPN sampling of all frames of the step)', 'launches': len(ev),
                    'avg_launch_ms': ms / max(1, len(ev)),
                    'note': 'algorithmic bytes = 12 taps x 1 KiB gathered + 3 KiB written per real row (SURVEY.md 8d); the gathers hit '
                            'L2 / MALL, so the figure can exceed what HBM alone would deliver'}
            metric = 'KITTI frames/sec (voxelize + FPN feature sample + fusion MLP + VFE fwd+bwd)'
        elif args.mode == 'vfe':
            workload = ('%s, %d raw pts -> %d pts, T=35, %d frames/GPU/step: crop+cropToSight+lidar2Img, voxelize, VFE stack '
                        '(SVFE + FCN + max) fwd+bwd, AdamW; voxel indices bit-exact vs the C oracle (checked in this run: %s)'
                        % (wl, RAW_POINTS, args.points, args.frames, vfe_check))
            ev = timers.get('hbm:voxelize', [])
            ms = sum(s.elapsed_time(e) for s, e, _ in ev)
            tb = sum(_hip.timer_value(b) for _, _, b in ev)
            ach = tb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            roof = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBPS, 'traffic': None,
                    'kernel': 'voxelizer (all its launches, %d frames per call)' % args.frames, 'launches': len(ev),
                    'avg_launch_ms': ms / max(1, len(ev)),
                    'note': 'algorithmic bytes = points + permutation read, voxel payload + indices written (SURVEY.md 8d)'}
            metric = 'KITTI frames/sec (voxelize+VFE fwd+bwd)'
        elif args.mode == 'full':
            workload = ('%s, %d raw pts -> %d pts after crop, grid 10x352x400, T=35, %d frames/GPU/step: crop+cropToSight+lidar2Img, '
                        'voxelize, classifyAnchors, fusion sampling+MLP, VFE, reindex+CML, RPN (all on this library\'s kernels, frame '
                        'sets), VoxelLoss, whole backward, all-reduce, AdamW; convmath=%s'
                        % (wl, RAW_POINTS, args.points, args.frames, main_math))
            roof = conv_roofline(timers, 0)
            roof['note'] += '; CML launches only (the RPN\'s 2-D convolutions run the same gather kernel and are listed under other_kernels)'
            metric = 'KITTI frames/sec (whole model fwd+bwd: voxelize+VFE+fusion+3Dconv+RPN+VoxelLoss)'
        else:
            workload = ('%s, %d pts, grid 10x352x400, T=35, %d frames/GPU/step one at a time through the nn.Module API: voxelize, '
                        'classifyAnchors, MVXNet.forward (compact rows, sparse first CML layer, RPN as one HIP autograd node), VoxelLoss, autograd backward, AdamW'
                        % (wl, args.points, args.frames))
            roof = conv_roofline(timers, 0)
            roof['note'] += '; drop-in path: one frame per launch set, frames one after the other under autograd'
            metric = 'KITTI frames/sec (MVXNet.forward + VoxelLoss + backward, nn.Module API)'
        out = {
            'metric': metric,
            'value': frames_total * args.steps / dt,
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'host_enqueue_ms_per_step': host_probe_ms,
            'host_in_timed_region_ms_per_step': host_ms[0],
            'library_launches_per_step': launches[0],
            'library_launches_per_frame': launches[0] / args.frames,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': dtype,
            'data': 'synthetic',
            'config': {'workload': workload, 'mode': args.mode, 'frames_per_gpu': args.frames, 'voxels_per_frame': nvox,
                       'parallelism': 'dp%d' % world, 'frame_sets': bool(pl.BATCHED and args.mode != 'dropin')},
            'roofline': roof,
            'hbm_stages': hbm_stages(timers_all),
            'ms_per_step_all_timers': dt_detail / n_detail * 1e3,
            'timers': 'value / ms_per_step / roofline: HIP events around the roofline kernel only; other_kernels / hbm_stages: a second '
                      'pass of %d steps with events around every library call (ms_per_step_all_timers)' % n_detail,
        }
        from modules import frames as _fr
        if _fr.KNOCKOUT:
            out['INVALID_diagnostic_knockout'] = sorted(_fr.KNOCKOUT)       # kernels were skipped: timing experiment, not a result
        # the other MFMA kernels of the step, each with EXECUTED FLOPs / event time inside the timed region (they share the
        # chip with the kernels of the other streams: side-stream weight gradients run beside main-stream convolutions)
        other = {}
        for name in ('conv3d_wgrad_bg', 'linear_fwd', 'linear_dgrad', 'linear_wgrad', 'rpn_conv', 'rpn_wgrad'):
            evs = timers_all.get(name, [])
            if evs:
                tms = sum(s.elapsed_time(e) for s, e, _ in evs)
                other[name] = {'launches': len(evs), 'avg_launch_ms': tms / len(evs)}
                fl = sum(_hip.timer_value(f) for _, _, f in evs)
                if fl > 0:
                    # algorithmic = one multiply-add per product; in a split arithmetic the matrix pipes execute 3 (bf16x3) or 6
                    # (bf16x6) bf16 MFMA products per algorithmic one.  The convolution kernels run split as a whole; of the row
                    # GEMMs the wide layers (n > 64: 99 % of the FLOPs) do, the narrow ones stay on the exact-f32 kernel -- their
                    # figure is therefore an upper bound by < 1 %
                    split_k = main_math != 'f32'
                    if main_math == 'bf16x3' and name == 'linear_fwd':
                        split_k = False                      # forward rows stay exact f32 in that mode (modules/_hip.py row_split)
                    alg = fl / (tms * 1e-3) / 1e12
                    mult, peak = (MATH_MFMAS[main_math], BF16_MFMA_PEAK_TFLOPS) if split_k else (1.0, FP32_MFMA_PEAK_TFLOPS)
                    other[name]['algorithmic_tflops'] = alg
                    other[name]['executed_tflops'] = mult * alg
                    other[name]['peak_tflops'] = peak
                    other[name]['frac'] = other[name]['executed_tflops'] / peak
                    other[name]['arithmetic'] = main_math if split_k else 'f32'
        out['other_kernels'] = other
        if world > 1 and bucket.times:
            # the gradient exchange's own duration (events on its streams): 'early' = everything but the step's last weight
            # gradient, on the communication stream while that kernel still runs; 'late' = the rest (modules/parallel.py)
            out['allreduce_ms_per_call'] = bucket.collective_ms()
        if alt:
            out['alt_modes'] = alt
        if fingerprint is not None:
            out['gradient_fingerprint'] = fingerprint
        if world == 1 and not args.no_cpu_baseline:
            del model, batch
            torch.cuda.empty_cache()
            if args.mode == 'vfe':
                out['cpu_baseline'] = cpu_baseline_vfe(args.points, args.workload)
            else:
                out['cpu_baseline'] = cpu_baseline(args.points, args.workload, with_rpn=args.mode == 'full')
            if vfe_check is not None:
                out['cpu_baseline']['voxel_indices_vs_oracle'] = vfe_check
        if vfe_check is not None:
            out['config']['voxel_indices_vs_oracle'] = vfe_check
        if args.mode == 'full' and full.get('last'):
            out['last_losses'] = full['last']
        # one compact record of every BASELINE config measured in this run (frames/s), FIRST and LAST in the line, so that a
        # truncated copy of the line keeps it whichever end survives
        summ = {'%s_%s_%s' % (args.mode, args.workload, main_math): round(out['value'], 1)}
        for a in alt:
            if 'baseline_config' in a:
                summ['cfg%d_%s_%s' % (a['baseline_config'], a['mode'], a['convmath'])] = round(a['value'], 1)
            else:
                summ['%s_%s_%s' % (args.mode, a['workload'], a['convmath'])] = round(a['value'], 1)
        if 'cpu_baseline' in out:
            summ['cpu_baseline_frames_per_s'] = round(out['cpu_baseline']['value'], 4)
        out = dict([('metric', out['metric']), ('value', out['value']), ('unit', out['unit']), ('summary', summ)] +
                   [(k, v) for k, v in out.items() if k not in ('metric', 'value', 'unit')] + [('summary_tail', summ)])
        return out
    return None


if __name__ == '__main__':
    main()
