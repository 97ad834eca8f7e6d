"""Frame sets: all frames of a step through ONE launch per layer.

The reference is strictly batch-1 (config.yml:18; modules/voxelnet/VoxelNet.py:19; MVXNet.py:23-24), so a batch of
B frames means B independent forwards with per-frame BatchNorm statistics and summed parameter gradients
(SURVEY.md 8e).  ``modules/tape.py`` runs them one after the other (about 170 launches per frame); this module runs the
same chain -- MVXNet.middle on compact rows: fusion sampling + fusion MLP (imhead/Pipe.py:23-104), the concat of
MVXNet.py:26, SVFE + FCN + max (voxelnet/Pipe.py:5-29, VoxelNet.py:27-33), reindex + CML (VoxelNet.py:16-36,
voxelnet/Pipe.py:31-43) -- ONCE for the whole frame set through the ``*_frames`` entry points of the C ABI: row
matrices hold the frames back to back, grids stack them along the depth axis, every per-frame quantity carries a
leading frame dimension.  Same kernels, same arithmetic per frame; parameter gradients are summed over the frames by
the reductions themselves.

Weight-gradient kernels go to the side stream (modules/_hip.py), everything else stays on the caller's stream.
"""
import ctypes
import os

import torch

import modules.config as cfg
from modules import _hip
from modules import Extension as X
from modules.layers.Blocks import conv_background_on, conv_split_math

R = _hip.STATS_REPLICAS


class FrameSet:
    """Voxels of F frames back to back + the compact-row bookkeeping every layer needs.

    voxels f32 (Vt, T, 9), coords i64 (Vt, 4); vox_off / real_off: host lists (F+1)."""

    def __init__(self, voxels, coords, vox_off, T):
        self.voxels, self.coords = voxels, coords
        self.vox_off = [int(v) for v in vox_off]
        self.F, self.T = len(vox_off) - 1, int(T)
        self.Vt = self.vox_off[-1]
        self.real_off = None
        self.desc = None
        self.sampled = None          # (FPN features of the real rows, status word) when sampled with the preparation (sample_rows)
        self.grid = None             # voxel index grid + activity masks / tile flags of the three CML layers (grid_activity), same

    # -- step 1 (enqueue only): dense-row -> compact-row map of all frames, real-row offsets stay on the device
    def enqueue_map(self):
        dev = self.voxels.device
        rows = self.Vt * self.T
        vox2d = self.voxels.view(rows, self.voxels.shape[-1])
        self.vox2d = vox2d
        self.row_map = torch.empty((rows,), dtype=torch.int32, device=dev)
        self.rows_sel = torch.empty((rows,), dtype=torch.int32, device=dev)
        self.n_real_dev = torch.empty((1,), dtype=torch.int32, device=dev)
        self.real_off_dev = torch.empty((self.F + 1,), dtype=torch.int32, device=dev)
        d0 = X.FramesDesc.make(self.vox_off, [0] * (self.F + 1), self.T)
        ws = _hip.workspace(X.lib.mvx_row_compact_workspace_bytes(rows), dev, 'compact')
        X.check(X.lib.mvx_row_compact_map_frames(X.ptr(vox2d), vox2d.shape[1], rows, X.ptr(self.row_map), X.ptr(self.rows_sel),
                                                 X.ptr(self.n_real_dev), X.ptr(ws), ws.numel(), d0.ref(),
                                                 X.ptr(self.real_off_dev), X.stream()), 'mvx_row_compact_map_frames')
        return self.real_off_dev

    # -- step 2 (after ONE host read of real_off_dev): the descriptor and the per-voxel offsets
    def finish_map(self, real_off):
        dev = self.voxels.device
        self.real_off = [int(v) for v in real_off]
        self.Rt = self.real_off[-1]
        self.desc = X.FramesDesc.make(self.vox_off, self.real_off, self.T)
        self.voff = torch.empty((self.Vt,), dtype=torch.int32, device=dev)
        self.vcnt = torch.empty((self.Vt,), dtype=torch.int32, device=dev)
        self.row_w = torch.empty((self.Rt + self.Vt,), dtype=torch.float32, device=dev)          # MVX_ROWS_VFE layout
        self.fusion_row_w = torch.empty((self.Rt + self.F,), dtype=torch.float32, device=dev)    # MVX_ROWS_FUSION layout
        X.check(X.lib.mvx_voxel_row_offsets_frames(X.ptr(self.row_map), self.Vt, self.T, self.Rt, X.ptr(self.voff),
                                                   X.ptr(self.vcnt), X.ptr(self.row_w), X.ptr(self.fusion_row_w),
                                                   self.desc.ref(), X.stream()), 'mvx_voxel_row_offsets_frames')
        return self


    def hand_over(self, stream, fenced=False):
        """The set was built on another stream (input preparation): keep its tensors alive for ``stream`` too.
        ``fenced`` (modules/pipeline.py): the consumer records an end-of-step event and the preparation stream waits for it before
        it allocates again, so the tensors only have to stay referenced until the step function returns -- no
        ``record_stream``, whose release depends on when the allocator next polls its events (the peak crept 4 % over 1,000
        identical steps and the pools held 2.6 x the live bytes, profiles/r05b_soak_hot.txt)."""
        kept = []
        for t in (self.voxels, self.coords, self.row_map, self.rows_sel, self.n_real_dev, self.real_off_dev, self.voff,
                  self.vcnt, self.row_w, self.fusion_row_w) + (tuple(self.sampled) if self.sampled is not None else ()) + \
                 (tuple(self.grid['tensors']) if self.grid is not None else ()):
            planes = getattr(t, '_mvx_planes', None)                 # the sampled features' planes of bf16 pieces (sample_rows)
            amax = _hip.amax_of(t)
            for u in (t, planes, amax):
                if u is None:
                    continue
                if fenced:
                    kept.append(u)
                else:
                    u.record_stream(stream)
        self._kept = kept                                            # sampled / grid are dropped mid-step: their tensors stay here


# ---------------------------------------------------------------------------------------------------------
# thin wrappers over the frame-set entry points (allocation + pointers only)
# ---------------------------------------------------------------------------------------------------------
def _stats(F, C, dev):
    buf, fz = _hip._acc_f64((F, R, 2, C), dev)
    return buf, fz


SAMPLE_PLANES = os.environ.get('MVX_SAMPLE_PLANES', '1') != '0'   # the FPN sampler writes the operand planes of its rows itself (A/B: 0)
TAPS_ON_SIDE = os.environ.get('MVX_TAPS_ON_SIDE', '1') != '0'   # the layers' tap sums in front of their weight gradient on the side stream (A/B: 0)
ROW_PITCH4 = os.environ.get('MVX_ROW_PITCH4', '1') != '0'   # the VFE input rows with a pitch that is a multiple of 4 floats (A/B: 0)
BG_READ_SET = os.environ.get('MVX_BG_READ_SET', '1') != '0'   # background tiles of x1 / x2 that the next layer does not read stay unwritten (A/B: 0)
BEV_FUSED = os.environ.get('MVX_BEV_FUSED', '1') != '0'   # conv3's BatchNorm apply writes the (F, C * D, H, W) map itself (A/B: 0)
TAP_SKIP = os.environ.get('MVX_TAP_SKIP', '1') != '0'     # conv2 / conv3 forward: skip depth taps with a background-only source halo
# DIAGNOSTIC ONLY (tools/knockout.sh): comma-separated kernel classes that are NOT launched, to measure what each class costs
# on the critical path of a step (step time with the class removed).  Results are garbage; bench.py marks such a run invalid.
KNOCKOUT = frozenset(k for k in os.environ.get('MVX_KNOCKOUT', '').split(',') if k)


def linear_bn(x, w, b, fs, kind, row_w, eps, tag='fusion', foreign=False):
    """rows -> (y = ReLU(x w^T + b), mean_inv (F,2,N)) with per-frame statistics formed inside the launch.  ``foreign``: x was not
    produced by this library (the sampled image features): _hip.foreign_split."""
    Rr, K = x.shape
    N = w.shape[0]
    w2 = w.reshape(N, -1)
    y = torch.empty((Rr, N), dtype=torch.float32, device=x.device)
    stats, fz = _stats(fs.F, N, x.device)
    counter = _hip._fin_slot(x.device, fz)              # arena slot when the statistics came from the (pre-zeroed) arena
    if counter is None:
        counter = torch.zeros((1,), dtype=torch.float64, device=x.device)
    mi = torch.empty((fs.F, 2, N), dtype=torch.float32, device=x.device)
    if 'lin_fwd' in KNOCKOUT:
        return y, mi
    sp, xfl = _hip.row_split(tag), 0                     # convmath bf16x3 / bf16x6 / fp16x3: the wide layers on the split-MFMA row GEMM
    xp = getattr(x, '_mvx_planes', None)                 # the producer of x wrote it as planes of bf16 pieces too (sample_rows)
    if xp is not None and _hip.PRECUT_FWD and _hip.precut_ok(sp, Rr, K, N):
        # the row GEMM on pre-cut operands (csrc/rowgemm_pre.hip): same products, same accumulation order -- bit-identical y
        _hip.linear_forward_pre(xp, _hip.weight_planes(w2, sp), b, y, stats, row_w, _hip.FLAG_RELU | fz | _hip.split_flags(sp, True),
                                counter, eps, mi, fs.desc, kind)
        return y, mi
    if foreign:
        sp, xfl = _hip.foreign_split(sp, x)
    if sp and int(sp) == 4:
        _hip.guard_fp16_weight(w2)                          # fp16x3: |w| < 255.9 (device-side status bit, checked once per step)
    code, sp = sp, _hip.split_flags(sp, True) | xfl
    with _hip._Timed('linear_fwd', 2.0 * Rr * K * N if _hip.KERNEL_TIMERS is not None else 0):
        _hip.bind_amax(code, x)                             # fp16x3: the range tag of a foreign input (the sampled image features)
        X.check(X.lib.mvx_linear_forward_bn_frames(_hip._vptr(x), _hip._ld(x), _hip._vptr(w2), _hip._ld(w2), 0, X.ptr(b),
                                                   _hip._vptr(y), _hip._ld(y), X.ptr(stats), X.ptr(row_w), Rr, K, N,
                                                   _hip.FLAG_RELU | fz | sp, X.ptr(counter), float(eps), X.ptr(mi), fs.desc.ref(),
                                                   kind, X.stream()), 'mvx_linear_forward_bn_frames')
    return y, mi


def bn_apply(y, mi, fs, kind):
    C = mi.shape[-1]
    rows = y.numel() // C
    if 'bn_apply_rows' in KNOCKOUT:
        return y
    out = torch.empty_like(y)
    with _hip._timed_bytes('bn_apply', 2 * y.numel() * 4):
        X.check(X.lib.mvx_bn_apply_frames(X.ptr(y), X.ptr(mi), X.ptr(out), rows, C, fs.desc.ref(), kind, X.stream()),
                'mvx_bn_apply_frames')
    return out


def bn_relu_backward(dyhat, y, mi, fs, kind, row_w, dbias_into, dz=None, planes=False):
    """dz (may alias dyhat); the bias gradient is ADDED to ``dbias_into`` (summed over the frames).  ``planes``: dz is
    written as three planes of bf16 pieces, int16 (3, rows, C), instead of f32 (a layer whose dz only feeds its own weight
    gradient on pre-cut operands: _hip.linear_wgrad_pre)."""
    C = mi.shape[-1]
    rows = y.numel() // C
    if planes:
        dzp = torch.empty((3, rows, C), dtype=torch.int16, device=y.device)
        if 'bn_bwd_rows' in KNOCKOUT:
            return dzp
        scratch, fz = _hip._acc_f64((X.lib.mvx_bn_backward_scratch_bytes_frames(C, fs.F) // 8,), y.device)
        with _hip._timed_bytes('bn_relu_backward', 5.5 * y.numel() * 4):
            X.check(X.lib.mvx_bn_relu_backward_planes_frames(X.ptr(dyhat), X.ptr(y), X.ptr(mi), 1.0, X.ptr(dzp), X.ptr(dbias_into),
                                                             X.ptr(scratch), X.ptr(row_w), rows, C, _hip.FLAG_ACCUMULATE | fz,
                                                             fs.desc.ref(), kind, None, X.stream()), 'mvx_bn_relu_backward_planes_frames')
        return dzp
    if dz is None:
        dz = torch.empty_like(y)
    if ('bn_bwd_rows' if kind != X.ROWS_GRID else 'bn_bwd_grid') in KNOCKOUT:
        return dz
    scratch, fz = _hip._acc_f64((X.lib.mvx_bn_backward_scratch_bytes_frames(C, fs.F) // 8,), y.device)
    amax = _hip.new_amax(y.device)
    with _hip._timed_bytes('bn_relu_backward', 5 * y.numel() * 4):
        X.check(X.lib.mvx_bn_relu_backward_frames(X.ptr(dyhat), X.ptr(y), X.ptr(mi), 1.0, X.ptr(dz), X.ptr(dbias_into),
                                                  X.ptr(scratch), X.ptr(row_w), rows, C, _hip.FLAG_ACCUMULATE | fz,
                                                  fs.desc.ref(), kind, X.ptr(amax), X.stream()), 'mvx_bn_relu_backward_frames')
    return _hip.tag_amax(dz, amax)                      # max |dz|: the range the fp16x3 kernels scale dz by


def bn_relu_backward_planes_parts(dyhat, y, mi, fs, kind, row_w, dbias_into, nparts):
    """Generator form of ``bn_relu_backward(..., planes=True)``: the apply pass is enqueued range by range; every iteration
    enqueues one part and yields ``(planes, row_lo, row_hi)`` -- the rows whose planes that part writes -- so the caller can
    put the weight gradient of those rows on the side stream before the next part is enqueued
    (mvx_bn_relu_backward_planes_part_frames).  The bias gradient is complete after the last part."""
    import ctypes
    C = mi.shape[-1]
    rows = y.numel() // C
    dzp = torch.empty((3, rows, C), dtype=torch.int16, device=y.device)
    if 'bn_bwd_rows' in KNOCKOUT:
        yield dzp, 0, rows
        return
    scratch, fz = _hip._acc_f64((X.lib.mvx_bn_backward_scratch_bytes_frames(C, fs.F) // 8,), y.device)
    rng = (ctypes.c_int64 * 2)()
    for part in range(nparts):
        with _hip._timed_bytes('bn_relu_backward', 5.5 * y.numel() * 4 / nparts):
            X.check(X.lib.mvx_bn_relu_backward_planes_part_frames(X.ptr(dyhat), X.ptr(y), X.ptr(mi), 1.0, X.ptr(dzp),
                                                                  X.ptr(dbias_into), X.ptr(scratch), X.ptr(row_w), rows, C,
                                                                  _hip.FLAG_ACCUMULATE | fz, fs.desc.ref(), kind, part, nparts, rng,
                                                                  X.stream()), 'mvx_bn_relu_backward_planes_part_frames')
        if rng[1] > rng[0]:
            yield dzp, int(rng[0]), int(rng[1])


# (The BatchNorm-backward reduction folded into the epilogue of the producing input-gradient kernel -- VERDICT r03 #1a, built and
# measured in round 4: hot 413.0 vs 418.0 frames/s, full 190.9 vs 190.8, its kernel variants spilling -- was removed in round 5.)


_GRAD_TARGETS = None        # id(parameter) -> buffer the gradient is ADDED into instead of .grad (a second lane of frame sets)


class grad_targets:
    """``with grad_targets(mapping)``: parameter gradients of the enclosed calls are added into ``mapping[id(p)]`` instead of
    ``p.grad`` (modules/pipeline.py: the second lane accumulates into its own flat buffer, so two lanes never write the
    same memory concurrently).  ``None`` = the parameters' own .grad."""

    def __init__(self, mapping):
        self.mapping = mapping

    def __enter__(self):
        global _GRAD_TARGETS
        self.old, _GRAD_TARGETS = _GRAD_TARGETS, self.mapping
        return self

    def __exit__(self, *exc):
        global _GRAD_TARGETS
        _GRAD_TARGETS = self.old
        return False


def _grad_of(p):
    if _GRAD_TARGETS is not None:
        return _GRAD_TARGETS[id(p)]
    if p.grad is None or not p.grad.is_contiguous():
        raise X.MvxHipError('the frame-set path adds gradients into existing contiguous .grad buffers (GradBucket)')
    return p.grad


# ---------------------------------------------------------------------------------------------------------
# the chain
# ---------------------------------------------------------------------------------------------------------
class _Saved:
    pass


def middle_forward(model, fs, fpn_levels, imsize, status_sink):
    """Forward of MVXNet.middle for the whole frame set.  ``fpn_levels``: per frame [f0, f1, f2] (1,C,H,W).
    Returns (mid (F,128,H,W), saved state for the backward)."""
    feat, S = rows_forward(model, fs, fpn_levels, imsize, status_sink)
    return cml_forward(model, fs, feat, S, status_sink), S


def sample_rows(head, fs, fpn_levels, imsize):
    """FPN features of the real rows of all frames (imhead/Pipe.py:23-82), each from its own frame's maps:
    ([real rows | one zero row per frame] x (levels * C), status word).  Depends on the inputs only, so the training
    pipeline runs it on the preparation stream for the NEXT step (pipeline.prepare_frame_set(..., sample=...)): 0.21 ms of
    HBM-bound work per 4-frame step that then runs beside MFMA-bound kernels instead of in front of the first GEMM."""
    dev = fs.voxels.device
    F, Rt = fs.F, fs.Rt
    levels = [f[0].permute(1, 2, 0).contiguous() for lv in fpn_levels for f in head.extractor(lv)]
    L = len(levels) // F
    C = levels[0].shape[2]
    ptrs = (ctypes.c_void_p * (F * L))(*[t.data_ptr() for t in levels])
    hw = (ctypes.c_int32 * (2 * L))(*[int(v) for t in levels[:L] for v in t.shape[:2]])
    compact = torch.empty((Rt + F, L * C), dtype=torch.float32, device=dev)
    compact[Rt:].zero_()                                   # the shared padded rows (Pipe.py:80)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    # fp16x3: the image features come from outside this library -- their range (max |value|) is formed by the sampler while it
    # writes them, so that the first fusion layer can scale them (forward: the coarse scale, _hip.foreign_split)
    amax = torch.zeros((1,), dtype=torch.float32, device=dev) if _hip.split_pieces() == 4 else None
    # the first fusion layer's weight gradient (and, opt-in, the layer itself) reads its input as planes of bf16 pieces
    # (csrc/rowgemm_pre.hip): the sampler writes them beside the f32 rows while the values are in registers
    w0 = head.fusion._layers()[0][0]
    planes = None
    want_planes = _hip.precut_ok(_hip.split_pieces(), Rt + F, L * C, w0.shape[0]) and _hip.split_pieces() == 3
    if want_planes and SAMPLE_PLANES:
        planes = torch.empty((3, Rt + F, L * C), dtype=torch.int16, device=dev)
        planes[:, Rt:].zero_()
    with _hip._timed_bytes('feature_sample', Rt * L * C * 4 * 5 + Rt * 9 * 4 + (Rt * L * C * 6 if planes is not None else 0)):
      if 'sample' not in KNOCKOUT:
        if planes is not None:
            # with the first fusion layer on planes too (_hip.PRECUT_FWD) nobody reads the f32 rows: they are not written (245 MB per
            # 4-frame step); `compact` stays as the rows' handle (shape, planes, range tag).  MVX_POISON_BG=1 (tests) fills it with
            # NaN instead, which any read would carry into the results
            K0, N0 = L * C, w0.shape[0]
            planes_only = bool(_hip.PRECUT_FWD and not KNOCKOUT and
                               _hip.precut_ok(_hip.row_split('fusion_%dx%d' % (N0, K0)), Rt + F, K0, N0) and
                               _hip.precut_ok(_hip.row_split('wgrad'), Rt + F, K0, N0))       # both readers take the planes
            if planes_only and os.environ.get('MVX_POISON_BG'):
                compact[:Rt].fill_(float('nan'))
            X.check(X.lib.mvx_feature_sample_rows_planes_frames(X.ptr(fs.vox2d), fs.vox2d.shape[1], X.ptr(fs.rows_sel), Rt, ptrs, hw, L,
                                                                C, float(imsize[0]), float(imsize[1]), float(cfg.eps),
                                                                None if planes_only else X.ptr(compact),
                                                                X.ptr(status), fs.desc.ref(), X.ptr(amax), X.ptr(planes), Rt + F,
                                                                X.stream()), 'mvx_feature_sample_rows_planes_frames')
        else:
            X.check(X.lib.mvx_feature_sample_rows_frames(X.ptr(fs.vox2d), fs.vox2d.shape[1], X.ptr(fs.rows_sel), Rt, ptrs, hw, L, C,
                                                         float(imsize[0]), float(imsize[1]), float(cfg.eps), X.ptr(compact),
                                                         X.ptr(status), fs.desc.ref(), X.ptr(amax), X.stream()),
                    'mvx_feature_sample_rows_frames')
    _hip.tag_amax(compact, amax)
    if planes is not None:
        compact._mvx_planes = planes
    elif want_planes:
        compact._mvx_planes = _hip.split_rows(compact, 3)           # the two-pass form (MVX_SAMPLE_PLANES=0)
    return compact, status


def rows_forward(model, fs, fpn_levels, imsize, status_sink, imfeat=None):
    """The row part of the chain: fusion sampling + fusion MLP -> concat -> SVFE -> FCN + max: voxel features
    (Vt,128) of all frames.  ``imfeat`` (Rt+F, 16), if given, stands for the fusion branch's output (VFE-only runs)."""
    head, bb = model.head, model.backbone
    dev = fs.voxels.device
    F, T, Rt, Vt = fs.F, fs.T, fs.Rt, fs.Vt
    _hip.require_plain_batchnorm()
    S = _Saved()
    S.fs = fs
    eps = cfg.eps
    S.fusion = []
    if imfeat is not None:
        x = imfeat
        return _vfe_forward(bb, fs, x, S, eps)
    # ---- fusion sampling (imhead/Pipe.py:23-82): real rows of all frames, each from its own frame's maps -- done with the
    # input preparation when the frame set was prepared a step ahead (sample_rows: no parameter is involved)
    if fs.sampled is not None:
        compact, status = fs.sampled
        fs.sampled = None
    else:
        compact, status = sample_rows(head, fs, fpn_levels, imsize)
    status_sink.append(status)
    if _hip.split_pieces() == 4:
        status_sink.append(_hip.fp16_weight_status(dev))    # fp16x3: the weight-range guard's word (modules/_hip.py guard_fp16_weight)
    # ---- fusion MLP (imhead/Pipe.py:84-104) on [real rows | one shared padded row per frame]
    x = compact
    for i, (w, b) in enumerate(head.fusion._layers()):
        y, mi = linear_bn(x, w, b, fs, X.ROWS_FUSION, fs.fusion_row_w, eps, 'fusion_%dx%d' % (w.shape[0], w[0].numel()),
                          foreign=(i == 0))
        S.fusion.append((x, w, b, y, mi))
        x = bn_apply(y, mi, fs, X.ROWS_FUSION)
    return _vfe_forward(bb, fs, x, S, eps)


def _vfe_forward(bb, fs, x, S, eps):
    dev = fs.voxels.device
    F, T, Rt, Vt = fs.F, fs.T, fs.Rt, fs.Vt
    # ---- concat with the 7 geometric channels (MVXNet.py:26): [real rows | one padded row per voxel]
    Fc = x.shape[1]
    # rows of 7 + Fc = 23 floats are written with a pitch of 24 (one zero column): the first VFE layer and its weight gradient then
    # read them with 16-byte loads (k = 24 against a weight padded with a zero column: the same sums) instead of falling to the
    # scalar-load forms of their kernels (config 2: 106 / 170 us for 11 / 50 MB)
    ldr = (7 + Fc + 3) & ~3 if ROW_PITCH4 else 7 + Fc
    rows23 = torch.empty((Rt + Vt, ldr), dtype=torch.float32, device=dev)
    X.check(X.lib.mvx_vfe_compact_input_pitch_frames(X.ptr(fs.vox2d), fs.vox2d.shape[1], X.ptr(fs.rows_sel), X.ptr(x), Fc, Rt, Vt,
                                                     X.ptr(rows23), ldr, fs.desc.ref(), X.stream()), 'mvx_vfe_compact_input_pitch_frames')
    S.fc = Fc
    # ---- SVFE (voxelnet/Pipe.py:5-29) and FCN + max (VoxelNet.py:27-33)
    S.vfe = []
    x = rows23
    for vfe in (bb.svfe.vfe1, bb.svfe.vfe2):
        w, b = vfe.fcn.fc.weight, vfe.fcn.fc.bias
        y, mi = linear_bn(x, _hip.padded_weight(w.reshape(w.shape[0], -1), x.shape[1]), b, fs, X.ROWS_VFE, fs.row_w, eps, 'vfe')
        Cn = w.shape[0]
        out = torch.empty((Rt + Vt, 2 * Cn), dtype=torch.float32, device=dev)
        am = torch.empty((Vt, Cn), dtype=torch.int32, device=dev)
        with _hip._timed_bytes('vfe_bn_max_concat', (Rt + Vt) * Cn * 4 * 3 + Vt * Cn * 4):      # read y, write [x | max] + argmax
            X.check(X.lib.mvx_vfe_bn_max_concat_frames(X.ptr(y), X.ptr(mi), X.ptr(out), X.ptr(am), Vt, T, Cn, X.ptr(fs.voff),
                                                       X.ptr(fs.vcnt), Rt, fs.desc.ref(), X.stream()), 'mvx_vfe_bn_max_concat_frames')
        S.vfe.append((x, w, b, y, mi, am))
        x = out
    w, b = bb.fcn.fc.weight, bb.fcn.fc.bias
    y, mi = linear_bn(x, w, b, fs, X.ROWS_VFE, fs.row_w, eps, 'vfe')
    feat = torch.empty((Vt, w.shape[0]), dtype=torch.float32, device=dev)
    am = torch.empty((Vt, w.shape[0]), dtype=torch.int32, device=dev)
    with _hip._timed_bytes('vfe_bn_segment_max', (Rt + Vt) * w.shape[0] * 4 + Vt * w.shape[0] * 8):   # read y, write max + argmax
        X.check(X.lib.mvx_bn_segment_max_frames(X.ptr(y), X.ptr(mi), X.ptr(feat), X.ptr(am), Vt, T, w.shape[0], X.ptr(fs.voff),
                                                X.ptr(fs.vcnt), Rt, fs.desc.ref(), X.stream()), 'mvx_bn_segment_max_frames')
    S.head = (x, w, b, y, mi, am)
    return feat, S


def grid_activity(model, fs):
    """Everything of the CML forward that depends on the voxel COORDINATES only (no parameter, no activation): the voxel index
    grid of the sparse first layer and the site-level activity of the three layers' outputs -- masks, halo / tile flags, the
    tile set of layer 2's restricted backward (csrc/activity.hip; VoxelNet.py:16-22, voxelnet/Pipe.py:36-42).  ~0.2 ms of small
    dependent launches in front of the first convolution: the training pipeline runs it with the input preparation of the NEXT
    step on the preparation stream (pipeline.prepare_frame_set(..., grid=model)), like the FPN sampling."""
    bb = model.backbone
    dev = fs.voxels.device
    F, Vt = fs.F, fs.Vt
    D0, H, W = cfg.voxelshape[2], cfg.voxelshape[0], cfg.voxelshape[1]
    ntl = _hip.n_tiles(H, W)
    idx_grid = torch.empty((X.lib.mvx_index_grid_bytes_frames(D0, H, W, F) // 4,), dtype=torch.int32, device=dev)
    st2 = torch.zeros((1,), dtype=torch.int32, device=dev)
    X.check(X.lib.mvx_index_grid_frames(X.ptr(fs.coords), Vt, D0, H, W, X.ptr(idx_grid), X.ptr(st2), fs.desc.ref(), X.stream()),
            'mvx_index_grid_frames')
    tensors = [idx_grid, st2]
    layers = []
    src, is_index, din = idx_grid, True, D0
    for li, m in enumerate((bb.cml.conv1, bb.cml.conv2, bb.cml.conv3)):
        sd, pd = m._sd, m._pd
        dout = _hip.conv_out_depth(din, sd, pd)
        mask = torch.empty((F * dout, H, W), dtype=torch.uint8, device=dev)
        hflag = torch.empty((F * dout, ntl), dtype=torch.int32, device=dev)
        tflag = torch.empty_like(hflag)
        X.check(X.lib.mvx_activity_dilate_frames(X.ptr(src), 1 if is_index else 0, din, dout, H, W, sd, pd, 0 if li == 0 else 1,
                                                 X.ptr(mask), X.ptr(hflag), X.ptr(tflag), F, X.stream()), 'mvx_activity_dilate_frames')
        layers.append((mask, hflag, tflag))
        tensors += [mask, hflag, tflag]
        src, is_index, din = mask, False, dout
    # the tiles layer 2's own restricted backward touches (from layer 1's and layer 2's tile flags)
    c2 = bb.cml.conv2
    d1 = _hip.conv_out_depth(D0, bb.cml.conv1._sd, bb.cml.conv1._pd)
    d2 = _hip.conv_out_depth(d1, c2._sd, c2._pd)
    bflag2 = torch.empty((F * d2, ntl), dtype=torch.int32, device=dev)
    X.check(X.lib.mvx_tile_dilate_flags_frames(X.ptr(layers[0][2]), X.ptr(layers[1][2]), d1, d2, H, W, c2._sd, c2._pd, X.ptr(bflag2), F,
                                               X.stream()), 'mvx_tile_dilate_flags_frames')
    tensors.append(bflag2)
    # which tiles of conv1's / conv2's normalised OUTPUT the next layer reads (its forward gather and weight gradient: the 3 x 3
    # tile neighbourhoods of the output tiles it computes): bn_apply_bg leaves the other background tiles unwritten
    reads = []
    din = d1
    for li, m in enumerate((bb.cml.conv2, bb.cml.conv3)):
        dout = _hip.conv_out_depth(din, m._sd, m._pd)
        R = torch.empty_like(layers[li][1])
        X.check(X.lib.mvx_tile_read_flags_frames(X.ptr(layers[li][1]), din, dout, H, W, m._sd, m._pd, X.ptr(R), F, X.stream()),
                'mvx_tile_read_flags_frames')
        reads.append(R)
        din = dout
    tensors += reads
    return dict(idx_grid=idx_grid, status=st2, layers=layers, bflag2=bflag2, reads=reads, tensors=tensors)


def cml_forward(model, fs, feat, S, status_sink, want_bev=True):
    """reindex + CML + the BEV reshape (VoxelNet.py:16-36) for the whole frame set: (Vt,128) -> (F,128,H,W).
    The channels-last CML output [F*D3][H][W][64] stays in ``S.x3`` (what modules/rpn_frames.py reads); with
    ``want_bev=False`` the (F,128,H,W) map is not materialised and None is returned."""
    bb = model.backbone
    dev = fs.voxels.device
    F, T, Rt, Vt = fs.F, fs.T, fs.Rt, fs.Vt
    eps = cfg.eps
    _hip.require_plain_batchnorm()
    # ---- reindex + conv1 through the voxel-GEMM factorisation (VoxelNet.py:16-22, voxelnet/Pipe.py:36)
    D0, H, W = cfg.voxelshape[2], cfg.voxelshape[0], cfg.voxelshape[1]
    c1, c2, c3 = bb.cml.conv1, bb.cml.conv2, bb.cml.conv3
    w1, b1 = c1.conv.weight, c1.conv.bias
    cout, cin = w1.shape[0], w1.shape[1]
    w_all = w1.permute(2, 3, 4, 0, 1).reshape(27 * cout, cin).contiguous()
    P, _ = _hip.linear_forward(feat, w_all, None, relu=False, want_stats=False, split=_hip.row_split('conv1'))
    ga = fs.grid if fs.grid is not None else grid_activity(model, fs)      # made with the input preparation when the set was prepared a step ahead
    fs.grid = None
    idx_grid = ga['idx_grid']
    status_sink.append(ga['status'])
    D1 = _hip.conv_out_depth(D0, c1._sd, c1._pd)
    # voxel-free tiles of y1 are never read (bn_apply_bg / the tile-restricted BatchNorm backward walk the flagged tiles only):
    # they are not written either (MVX_FLAG_NO_BG_FILL: 742 -> ~300 MB of stores per 4-frame step).  MVX_POISON_BG=1 (tests)
    # fills them with NaN instead, which any read would carry into the results
    if os.environ.get('MVX_POISON_BG'):
        y1 = torch.full((F * D1, H, W, cout), float('nan'), dtype=torch.float32, device=dev)
    else:
        y1 = torch.empty((F * D1, H, W, cout), dtype=torch.float32, device=dev)
    stats, fz = _stats(F, cout, dev)
    with _hip._timed_bytes('sparse_conv_output', y1.numel() * 4 + F * D0 * H * W * 4):
      if 'sparse_out' not in KNOCKOUT:
        # only the tiles of y1 that hold a site next to a voxel are built: the flags of this layer's output from grid_activity
        X.check(X.lib.mvx_sparse_conv_output_tiles_frames(X.ptr(P), X.ptr(idx_grid), X.ptr(b1), X.ptr(y1), X.ptr(stats), D0, D1, H, W,
                                                          cout, c1._sd, c1._pd, _hip.FLAG_RELU | _hip.FLAG_NO_BG_FILL | fz, F,
                                                          X.ptr(ga['layers'][0][2]), X.stream()),
                'mvx_sparse_conv_output_tiles_frames')
    mi1 = torch.empty((F, 2, cout), dtype=torch.float32, device=dev)
    X.check(X.lib.mvx_bn_finalize_frames(X.ptr(stats), float(D1 * H * W), float(eps), X.ptr(mi1), cout, F, X.stream()),
            'mvx_bn_finalize_frames')
    ntl = _hip.n_tiles(H, W)

    def dilate(li):
        return ga['layers'][li]

    def background(bg_pre, bias, mi, planes, C_):
        c_out = torch.empty((F * planes, C_), dtype=torch.float32, device=dev)
        y_bg = torch.empty_like(c_out)
        X.check(X.lib.mvx_bn_background_frames(X.ptr(bg_pre), X.ptr(bias), X.ptr(mi), planes, C_, _hip.FLAG_RELU, X.ptr(y_bg),
                                               X.ptr(c_out), F, X.stream()), 'mvx_bn_background_frames')
        return c_out, y_bg

    def bn_apply_bg(y, mi, c_bg, tflag, planes, read=None):
        """BatchNorm apply of a layer output with a background: tiles without a non-background site take the normalised
        constant without being read (bit-identical to bn_apply).  ``read``: tile flags of what the consuming layer reads
        (grid_activity): the background tiles outside that set are not written (MVX_POISON_BG=1, tests: they hold NaN)."""
        if 'bn_apply_cml' in KNOCKOUT:
            return y
        if read is not None and not BG_READ_SET:
            read = None
        out = torch.full_like(y, float('nan')) if (read is not None and os.environ.get('MVX_POISON_BG')) else torch.empty_like(y)
        Cn = y.shape[-1]
        # algorithmic bytes (timing runs only): flagged tiles are read and written, the others only written
        nbytes = (lambda: (tflag.ne(0).sum() + (torch.logical_or(tflag.ne(0), read.ne(0)).sum() if read is not None else tflag.numel()))
                  * (128 * Cn * 4)) if _hip.KERNEL_TIMERS is not None else 0
        with _hip._timed_bytes('bn_apply', nbytes):
            if read is not None:
                X.check(X.lib.mvx_bn_apply_tiles_read_frames(X.ptr(y), X.ptr(mi), X.ptr(c_bg), X.ptr(tflag), X.ptr(read), X.ptr(out),
                                                             planes, H, W, Cn, F, X.stream()), 'mvx_bn_apply_tiles_read_frames')
            else:
                X.check(X.lib.mvx_bn_apply_tiles_frames(X.ptr(y), X.ptr(mi), X.ptr(c_bg), X.ptr(tflag), X.ptr(out), planes, H, W, Cn,
                                                        F, X.stream()), 'mvx_bn_apply_tiles_frames')
        return out

    mask1, hflag1, tflag1 = dilate(0)
    cc1, ybg1 = background(None, b1, mi1, D1, cout)
    x1 = bn_apply_bg(y1, mi1, cc1, tflag1, D1, ga['reads'][0])
    S.conv1 = dict(feat=feat, w=w1, b=b1, w_all=w_all, y=y1, mi=mi1, c=cc1, ybg=ybg1, tflag=tflag1, D0=D0, D1=D1)

    # ---- conv2, conv3 on the MFMA gather kernel with the background rewrite (voxelnet/Pipe.py:37-42)
    split = conv_split_math()                # convmath: bf16x3 -> the split-MFMA forms of the three conv2 / conv3 kernels
    if not conv_background_on():
        raise X.MvxHipError('the frame-set path needs convbackground (the default configuration)')
    S.split = split
    S.convs = []
    x_in, din, c_in, mask_in, hflag_in, tflag_in = x1, D1, cc1, mask1, hflag1, tflag1
    bflag_in = tflag1                                     # tiles on which the gradient of x_in is produced / consumed
    for li, m in enumerate((c2, c3)):
        w, b = m.conv.weight, m.conv.bias
        co, ci = w.shape[0], w.shape[1]
        sd, pd = m._sd, m._pd
        dout = _hip.conv_out_depth(din, sd, pd)
        wpk = m._packer(False, split)
        # background constants of this layer: [planes][co] totals, followed by the per-depth-tap ones the exact-f32 gather
        # uses to skip the depth taps whose source halo holds no active site (TAP_SKIP)
        bg_all = torch.empty((F * dout * 13, co), dtype=torch.float32, device=dev)      # totals | 3 depth taps | 9 border classes
        bg_pre = bg_all[:F * dout]
        X.check(X.lib.mvx_conv3d_background_taps_frames(X.ptr(w), X.ptr(c_in), din, dout, ci, co, sd, pd, X.ptr(bg_all), F,
                                                        X.stream()), 'mvx_conv3d_background_taps_frames')
        mask_o, hflag_o, tflag_o = dilate(li + 1)
        y = torch.empty((F * dout, H, W, co), dtype=torch.float32, device=dev)
        stats, fz = _stats(F, co, dev)
        fin = _hip._fin_slot(dev, fz)
        if fin is None:
            fin = torch.zeros((1,), dtype=torch.float64, device=dev)
        mi = torch.empty((F, 2, co), dtype=torch.float32, device=dev)
        counter = None
        if _hip.KERNEL_TIMERS is not None:
            if _hip.EXEC_STAGES is None:
                _hip.EXEC_STAGES = torch.zeros((1,), dtype=torch.int64, device=dev)
            counter = _hip.EXEC_STAGES
        if split:
            with _hip._Timed('conv3d_gather_bg', F * _hip.conv_flops(dout, din, H, W, ci, co, sd, pd) if _hip.KERNEL_TIMERS is not None else 0):
                X.check(X.lib.mvx_conv3d_forward_bg_split_frames(X.ptr(x_in), X.ptr(wpk), X.ptr(b), X.ptr(y), X.ptr(stats), din, dout,
                                                                 H, W, ci, co, sd, pd,
                                                                 _hip.FLAG_RELU | fz | (_hip.FLAG_BG_TAPS if TAP_SKIP else 0) | _hip.split_flags(split),
                                                                 X.ptr(hflag_in), X.ptr(mask_o), X.ptr(bg_pre), 1, X.ptr(counter), F,
                                                                 X.stream()),
                        'mvx_conv3d_forward_bg_split_frames')
            X.check(X.lib.mvx_bn_finalize_frames(X.ptr(stats), float(dout * H * W), float(eps), X.ptr(mi), co, F, X.stream()),
                    'mvx_bn_finalize_frames')
        elif 'gather_fwd' not in KNOCKOUT:
          with _hip._Timed('conv3d_gather_bg', F * _hip.conv_flops(dout, din, H, W, ci, co, sd, pd) if _hip.KERNEL_TIMERS is not None else 0):
            X.check(X.lib.mvx_conv3d_forward_bg_frames(X.ptr(x_in), X.ptr(wpk), X.ptr(b), X.ptr(y), X.ptr(stats), din, dout, H, W,
                                                       ci, co, sd, pd, _hip.FLAG_RELU | fz | (_hip.FLAG_BG_TAPS if TAP_SKIP else 0),
                                                       X.ptr(hflag_in), X.ptr(mask_o),
                                                       X.ptr(bg_pre), 1, X.ptr(counter), X.ptr(fin), float(dout * H * W),
                                                       float(eps), X.ptr(mi), X.ptr(_hip._work_counter(dev)), F, X.stream()),
                    'mvx_conv3d_forward_bg_frames')
        c_o, ybg_o = background(bg_pre, b, mi, dout, co)         # the background of this layer's output
        if want_bev and li == 1 and co <= 64 and 'bn_apply_cml' not in KNOCKOUT and BEV_FUSED:
            # the last layer's normalised output goes straight into the reference's (F, C * D, H, W) layout
            # (mvx_bn_apply_tiles_bev_frames): no channels-last copy of it, no transposition pass
            mid = torch.empty((F, co * dout, H, W), dtype=torch.float32, device=dev)
            nbytes = (lambda: (tflag_o.ne(0).sum() + tflag_o.numel()) * (128 * co * 4)) if _hip.KERNEL_TIMERS is not None else 0
            with _hip._timed_bytes('bn_apply', nbytes):
                X.check(X.lib.mvx_bn_apply_tiles_bev_frames(X.ptr(y), X.ptr(mi), X.ptr(c_o), X.ptr(tflag_o), X.ptr(mid), dout, H, W, co,
                                                            F, X.stream()), 'mvx_bn_apply_tiles_bev_frames')
            x_out = None
        else:
            # conv2's output is read by conv3 only; conv3's (channels-last form: --mode full) by the RPN, everywhere
            x_out = bn_apply_bg(y, mi, c_o, tflag_o, dout, ga['reads'][1] if li == 0 else None)
        rec = dict(x=x_in, w=w, b=b, y=y, mi=mi, din=din, dout=dout, sd=sd, pd=pd, m=m, c_in=c_in, hflag_in=hflag_in,
                   bflag_in=bflag_in, split=split)
        if li == 0:
            # the tiles this layer's own restricted backward touches
            bflag_o = ga['bflag2']
            rec.update(c_out=c_o, ybg_out=ybg_o, bflag_out=bflag_o)
            c_in, bflag_in = c_o, bflag_o
        S.convs.append(rec)
        x_in, din, mask_in, hflag_in, tflag_in = x_out, dout, mask_o, hflag_o, tflag_o
    D3 = din
    S.D3, S.H, S.W, S.C3 = D3, H, W, co
    S.x3 = x_in                                # None when the last layer wrote the (F, C * D, H, W) map itself
    if not want_bev:
        return None
    if x_in is None:
        return mid
    mid = torch.empty((F, x_in.shape[-1] * D3, H, W), dtype=torch.float32, device=dev)
    with _hip._timed_bytes('cl_bev_transpose', 2 * x_in.numel() * 4):
        X.check(X.lib.mvx_cl_to_bev_frames(X.ptr(x_in), X.ptr(mid), D3, H, W, x_in.shape[-1], 0, F, X.stream()), 'mvx_cl_to_bev_frames')
    return mid


def _wgrad_bg_flops(rec, F):
    """EXECUTED FLOPs of a background-aware weight-gradient launch as a zero-argument callable (timing runs only; evaluated
    after the timed region): the kernel runs the (plane, tile) steps whose source halo holds a non-background site, every
    step = 8 x 16 sites x Cin x Cout x 9 taps.  Only the flag tensor is captured, not the activations."""
    din, dout, sd, pd = rec['din'], rec['dout'], rec['sd'], rec['pd']
    hflag = rec['hflag_in']
    co, ci = rec['w'].shape[0], rec['w'].shape[1]

    def count():
        hf = hflag.view(F, din, -1).ne(0)
        steps = 0
        for kd in range(3):
            for d in range(dout):
                ds = d * sd - pd + kd
                if 0 <= ds < din:
                    steps = steps + hf[:, ds].sum()
        return float(steps) * (2.0 * 128 * ci * co * 9)
    return count


def _wgrad_bg(rec, dz, tap_sums, F, H, W):
    if 'wgrad_bg' in KNOCKOUT:
        return
    x, w = rec['x'], rec['w']
    co, ci = w.shape[0], w.shape[1]
    dw = _grad_of(w)
    fl = _wgrad_bg_flops(rec, F) if _hip.KERNEL_TIMERS is not None else 0      # a callable, evaluated after the timed region
    nbytes = X.lib.mvx_conv3d_wgrad_bg_workspace_bytes_frames(rec['dout'], H, W, ci, co, F)
    # split arithmetic (bf16x3 / bf16x6): the same entry point, conv3d_wgrad4s (csrc/conv3d.hip) under MVX_FLAG_SPLIT[3]
    sp = _hip.split_flags(rec.get('split'), True)
    with _hip._SideStream(x, dz, tap_sums, rec['c_in'], rec['hflag_in']):
        ws = _hip.workspace(nbytes, x.device, 'wgrad_bg_side')
        with _hip._Timed('conv3d_wgrad_bg', fl):
            _hip.bind_amax(rec.get('split'), None, dz)
            X.check(X.lib.mvx_conv3d_wgrad_bg_frames(X.ptr(x), X.ptr(dz), X.ptr(dw), rec['din'], rec['dout'], H, W, ci, co,
                                                     rec['sd'], rec['pd'], _hip.FLAG_ACCUMULATE | sp, X.ptr(rec['hflag_in']),
                                                     X.ptr(rec['c_in']), X.ptr(tap_sums), X.ptr(ws), ws.numel(), F, X.stream()),
                    'mvx_conv3d_wgrad_bg_frames')


def _linear_wgrad_side(x, dz, w):
    """dW += dz^T x over the rows of ALL frames, on the side stream."""
    if 'lin_wgrad' in KNOCKOUT:
        return
    _hip.linear_wgrad(x, dz, accumulate_into=_grad_of(w).view(w.shape[0], -1))


def middle_backward(model, S, grad_mid):
    """Backward of the chain for the whole frame set.  ``grad_mid`` (F or 1, 128, H, W) = dL/d(mid); parameter gradients
    are ADDED into the existing .grad buffers (summed over the frames)."""
    old_async, _hip.ASYNC_WGRAD = _hip.ASYNC_WGRAD, True
    try:
        rows_backward(model, S, cml_backward(model, S, grad_mid))
    finally:
        _hip.ASYNC_WGRAD = old_async


_MUTATE = {}        # TESTS ONLY (tests/test_fullsize_gpu.py): name -> factor applied to a closed-form term of the restricted
                    # backward, to prove that the parity tests would notice a 1 % error in it.  Empty in every product run.


def _mut(name, t):
    return _hip.mutate(_MUTATE, name, t) if _MUTATE else t


def cml_backward(model, S, grad_mid, g_cl=None):
    """Backward of cml_forward: returns dL/d(voxel features) (Vt,128).  ``g_cl``: the gradient already in the
    channels-last layout of ``S.x3`` ([F*D3][H][W][64], from modules/rpn_frames.py) instead of ``grad_mid`` (F or 1,128,H,W)."""
    fs = S.fs
    F, T, Rt, Vt = fs.F, fs.T, fs.Rt, fs.Vt
    H, W, D3, C3 = S.H, S.W, S.D3, S.C3
    dev = fs.voxels.device
    gm = grad_mid.contiguous() if g_cl is None else None
    if g_cl is not None:
        g = g_cl
    elif gm.shape[0] == 1 and F > 1:
        # one gradient map for every frame: transposed once, written F times (instead of a transposition and a repeat of it)
        g = torch.empty((F * D3, H, W, C3), dtype=torch.float32, device=dev)
        X.check(X.lib.mvx_bev_to_cl_broadcast(X.ptr(gm), X.ptr(g), D3, H, W, C3, F, X.stream()), 'mvx_bev_to_cl_broadcast')
    else:
        g = torch.empty((F * D3, H, W, C3), dtype=torch.float32, device=dev)
        X.check(X.lib.mvx_cl_to_bev_frames(X.ptr(g), X.ptr(gm), D3, H, W, C3, 1, F, X.stream()), 'mvx_cl_to_bev_frames')

    def tap_sums(dz, planes, Cn, tile_flags=None, inactive=None):
        Tt = torch.empty((planes, 9, Cn), dtype=torch.float32, device=dev)
        if 'tap_sums' in KNOCKOUT:
            return Tt
        ws = _hip.workspace(X.lib.mvx_plane_tap_sums_workspace_bytes(planes, Cn), dev, 'tap_sums')
        X.check(X.lib.mvx_plane_tap_sums(X.ptr(dz), planes, H, W, Cn, X.ptr(tile_flags), X.ptr(inactive), X.ptr(Tt), X.ptr(ws),
                                         ws.numel(), X.stream()), 'mvx_plane_tap_sums')
        return Tt

    def tap_sums_side(name, dz, planes, Cn, tile_flags=None, inactive=None):
        """The tap sums of a layer on the SIDE stream, in front of their first reader (the layer's weight gradient); returns
        (T, event): the main stream goes straight from the BatchNorm backward to the input-gradient gather and waits for the
        event only in front of input_grad_sums.  (T, None): computed inline."""
        if not (TAPS_ON_SIDE and _hip.ASYNC_WGRAD) or 'tap_sums' in KNOCKOUT:
            return _mut(name, tap_sums(dz, planes, Cn, tile_flags, inactive)), None
        main = torch.cuda.current_stream(dev)
        with _hip._SideStream(dz, tile_flags, inactive):
            Tt = _mut(name, tap_sums(dz, planes, Cn, tile_flags, inactive))
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        Tt.record_stream(main)                  # allocated on the side stream's pool, read by input_grad_sums on the main stream
        return Tt, ev

    def dgrad_tiles(rec, dz, bflag):
        w = rec['w']
        co, ci = w.shape[0], w.shape[1]
        dx = torch.empty((F * rec['din'], H, W, ci), dtype=torch.float32, device=dev)
        wpd = rec['m']._packer(True, rec.get('split') or 0)
        counter = None
        if _hip.KERNEL_TIMERS is not None:
            if _hip.EXEC_STAGES is None:
                _hip.EXEC_STAGES = torch.zeros((1,), dtype=torch.int64, device=dev)
            counter = _hip.EXEC_STAGES
        if rec.get('split'):
            with _hip._Timed('conv3d_gather_tiles', F * _hip.conv_flops(rec['din'], rec['dout'], H, W, co, ci, rec['sd'], rec['pd'], True)
                             if _hip.KERNEL_TIMERS is not None else 0):
                _hip.bind_amax(rec['split'], dz)
                X.check(X.lib.mvx_conv3d_dgrad_tiles_split_frames(X.ptr(dz), X.ptr(wpd), X.ptr(dx), rec['din'], rec['dout'], H, W, ci,
                                                                  co, rec['sd'], rec['pd'], _hip.split_flags(rec['split']), X.ptr(bflag),
                                                                  X.ptr(counter), F, X.stream()), 'mvx_conv3d_dgrad_tiles_split_frames')
            return dx
        if 'gather_dgrad' in KNOCKOUT:
            return dx
        with _hip._Timed('conv3d_gather_tiles', F * _hip.conv_flops(rec['din'], rec['dout'], H, W, co, ci, rec['sd'], rec['pd'], True)
                         if _hip.KERNEL_TIMERS is not None else 0):
            X.check(X.lib.mvx_conv3d_dgrad_tiles_frames(X.ptr(dz), X.ptr(wpd), X.ptr(dx), rec['din'], rec['dout'], H, W, ci, co,
                                                        rec['sd'], rec['pd'], X.ptr(bflag), X.ptr(counter),
                                                        X.ptr(_hip._work_counter(dev)), F, X.stream()), 'mvx_conv3d_dgrad_tiles_frames')
        return dx

    def input_grad_sums(rec, Tt):
        w = rec['w']
        A = torch.empty((F * rec['din'], w.shape[1]), dtype=torch.float32, device=dev)
        X.check(X.lib.mvx_conv3d_input_grad_sums_frames(X.ptr(w), X.ptr(Tt), rec['din'], rec['dout'], w.shape[1], w.shape[0],
                                                        rec['sd'], rec['pd'], X.ptr(A), F, X.stream()), 'mvx_conv3d_input_grad_sums_frames')
        return A

    def bn_bwd_tiles(gin, y, mi, c_bg, y_bg, A, bflag, planes, Cn, bias, want_inactive):
        dz = torch.empty_like(y)
        ws = _hip.workspace(X.lib.mvx_bn_relu_backward_tiles_workspace_bytes_frames(planes, H, W, Cn, F), dev, 'bn_tiles')
        inact = torch.empty((F * planes, Cn), dtype=torch.float32, device=dev) if want_inactive else None
        amax = _hip.new_amax(dev)                                             # max |dz| over the written tiles (zeroed by the call)
        if 'bn_bwd_tiles' in KNOCKOUT:
            return dz, inact
        # algorithmic bytes (timing runs only): the flagged 8x16 tiles, two passes reading dyhat and y, the second writing dz
        nbytes = (lambda: bflag.ne(0).sum() * (128 * Cn * 4 * 5)) if _hip.KERNEL_TIMERS is not None else 0
        with _hip._timed_bytes('bn_relu_backward_tiles', nbytes):
            X.check(X.lib.mvx_bn_relu_backward_tiles_frames(X.ptr(gin), X.ptr(y), X.ptr(mi), X.ptr(c_bg), X.ptr(y_bg), X.ptr(A),
                                                            X.ptr(bflag), planes, H, W, Cn, X.ptr(dz), X.ptr(_grad_of(bias)),
                                                            X.ptr(inact), X.ptr(amax), _hip.FLAG_ACCUMULATE, X.ptr(ws), ws.numel(), F,
                                                            X.stream()), 'mvx_bn_relu_backward_tiles_frames')
        return _hip.tag_amax(dz, amax), inact

    # ---- conv3: dense gradient in, restricted gradient + closed-form plane sums out
    r3, r2 = S.convs[1], S.convs[0]
    dz3 = bn_relu_backward(g, r3['y'], r3['mi'], fs, X.ROWS_GRID, None, _grad_of(r3['b']))
    T3, ev3 = tap_sums_side('T3', dz3, F * r3['dout'], r3['w'].shape[0])
    _wgrad_bg(r3, dz3, T3, F, H, W)
    g2 = dgrad_tiles(r3, dz3, r3['bflag_in'])
    if ev3 is not None:
        torch.cuda.current_stream(dev).wait_event(ev3)
    A2 = _mut('A2', input_grad_sums(r3, T3))
    # ---- conv2
    dz2, inact2 = bn_bwd_tiles(g2, r2['y'], r2['mi'], r2['c_out'], r2['ybg_out'], A2, r2['bflag_out'], r2['dout'],
                               r2['w'].shape[0], r2['b'], True)
    inact2 = _mut('inact2', inact2)
    T2, ev2 = tap_sums_side('T2', dz2, F * r2['dout'], r2['w'].shape[0], r2['bflag_out'], inact2)
    _wgrad_bg(r2, dz2, T2, F, H, W)
    g1 = dgrad_tiles(r2, dz2, r2['bflag_in'])
    if ev2 is not None:
        torch.cuda.current_stream(dev).wait_event(ev2)
    A1 = _mut('A1', input_grad_sums(r2, T2))
    # ---- conv1 (voxel-GEMM factorisation): gradient only next to the voxels
    c1 = S.conv1
    w1 = c1['w']
    cout, cin = w1.shape[0], w1.shape[1]
    dz1, _ = bn_bwd_tiles(g1, c1['y'], c1['mi'], c1['c'], c1['ybg'], A1, c1['tflag'], c1['D1'], cout, c1['b'], False)
    G = torch.empty((Vt, 27 * cout), dtype=torch.float32, device=dev)
    cm = model.backbone.cml.conv1
    X.check(X.lib.mvx_sparse_conv_gather_dz_frames(X.ptr(dz1), X.ptr(fs.coords), Vt, X.ptr(G), c1['D0'], c1['D1'], H, W, cout,
                                                   cm._sd, cm._pd, fs.desc.ref(), X.stream()), 'mvx_sparse_conv_gather_dz_frames')
    _hip.tag_amax(G, _hip.amax_of(dz1))                                    # G's rows are rows of dz1
    # conv1's weight gradient ((27 * cout, cin) = G^T feat, then reordered into the parameter's layout): off the main stream like
    # the other weight gradients -- the main stream goes on to the input gradient
    with (_hip._SideStream(c1['feat'], G) if TAPS_ON_SIDE else _hip._Inline()):
        dw_all = _hip.linear_wgrad(c1['feat'], G)
    with _hip._SideStream(dw_all):
        _grad_of(w1).add_(dw_all.reshape(3, 3, 3, cout, cin).permute(3, 4, 0, 1, 2))
    # dfeat = G w_all: the weight as a row-major [cin][27 cout] matrix, so that both operands are read along k
    dfeat, _ = _hip.linear_forward(G, c1['w_all'].t().contiguous(), None, relu=False, want_stats=False, label='linear_dgrad',
                                   split=_hip.row_split('dgrad'))
    return dfeat


def _rows_dgrad(dz, w2):
    """dx = dz w (rows x K): the input gradient of a row layer."""
    if 'lin_dgrad' in KNOCKOUT:
        return torch.empty((dz.shape[0], w2.shape[1]), dtype=torch.float32, device=dz.device)
    gx, _ = _hip.linear_forward(dz, _hip.transposed_weight(w2), None, relu=False, want_stats=False, label='linear_dgrad',
                                split=_hip.row_split('dgrad'))
    return gx


def rows_backward(model, S, dfeat):
    """Backward of rows_forward from dL/d(voxel features) (Vt,128); needs _hip.ASYNC_WGRAD set by the caller."""
    fs = S.fs
    F, T, Rt, Vt = fs.F, fs.T, fs.Rt, fs.Vt
    dev = dfeat.device
    # ---- FCN + max
    x, w, b, y, mi, am = S.head
    Cn = w.shape[0]
    dyh = torch.empty((Rt + Vt, Cn), dtype=torch.float32, device=dev)
    X.check(X.lib.mvx_segment_max_backward(X.ptr(dfeat), X.ptr(am), X.ptr(dyh), Vt, T, Cn, X.ptr(fs.voff), X.ptr(fs.vcnt), Rt,
                                           X.stream()), 'mvx_segment_max_backward')
    dz = bn_relu_backward(dyh, y, mi, fs, X.ROWS_VFE, fs.row_w, _grad_of(b), dz=dyh)
    _linear_wgrad_side(x, dz, w)
    gx = _rows_dgrad(dz, w)
    # ---- VFE 2, VFE 1
    for x, w, b, y, mi, am in reversed(S.vfe):
        Cn = w.shape[0]
        dyh = torch.empty((Rt + Vt, Cn), dtype=torch.float32, device=dev)
        with _hip._timed_bytes('vfe_max_concat_backward', (Rt + Vt) * Cn * 4 * 3 + Vt * Cn * 4):  # read [gx | gmax] + argmax, write dyhat
            X.check(X.lib.mvx_vfe_max_concat_backward(X.ptr(gx), X.ptr(am), X.ptr(dyh), Vt, T, Cn, X.ptr(fs.voff), X.ptr(fs.vcnt),
                                                      Rt, X.stream()), 'mvx_vfe_max_concat_backward')
        dz = bn_relu_backward(dyh, y, mi, fs, X.ROWS_VFE, fs.row_w, _grad_of(b), dz=dyh)
        w2 = w.reshape(w.shape[0], -1)
        if x.shape[1] != w2.shape[1] and 'lin_wgrad' not in KNOCKOUT:
            # rows with a padded pitch (23 -> 24): the gradient of the padded weight, its real columns added into the parameter's
            with _hip._SideStream(x, dz):
                dwp = _hip.linear_wgrad(x, dz)
                _grad_of(w).view(w2.shape).add_(dwp[:, :w2.shape[1]])
        else:
            _linear_wgrad_side(x, dz, w)
        gx = _rows_dgrad(dz, w)
    # ---- concat backward: gradient of the fused image features ([real rows | shared padded row per frame])
    Fc = S.fc
    gim = torch.empty((Rt + F, Fc), dtype=torch.float32, device=dev)
    scratch = torch.empty((F * Fc,), dtype=torch.float64, device=dev)
    X.check(X.lib.mvx_vfe_compact_input_backward_frames(X.ptr(gx), Fc, Rt, Vt, X.ptr(gim), X.ptr(scratch), fs.desc.ref(),
                                                        X.stream()), 'mvx_vfe_compact_input_backward_frames')
    # ---- fusion MLP, last layer first; the sampled features carry no gradient
    gx = gim
    for i in range(len(S.fusion) - 1, -1, -1):
        x, w, b, y, mi = S.fusion[i]
        xp = getattr(x, '_mvx_planes', None)
        if i == 0 and xp is not None and _hip.precut_ok(_hip.row_split('wgrad'), x.shape[0], x.shape[1], w.shape[0]):
            # the step's last and largest weight gradient on pre-cut operands: the BatchNorm backward writes dz as planes (no other
            # reader: the sampled features carry no gradient), the weight gradient moves both operands by DMA (rowgemm_pre.hip)
            # ... and in TAIL_PARTS row ranges: the weight gradient of a range runs on the side stream beside the apply pass of
            # the next one, so that what is left after the main stream's last kernel is the last range's product only
            first = True
            for dzp, lo, hi in bn_relu_backward_planes_parts(gx, y, mi, fs, X.ROWS_FUSION, fs.fusion_row_w, _grad_of(b),
                                                             _hip.TAIL_PARTS):
                if first:
                    _hip.mark_tail(dev)
                    first = False
                if 'lin_wgrad' not in KNOCKOUT:
                    _hip.linear_wgrad_pre(xp, dzp, accumulate_into=_grad_of(w).view(w.shape[0], -1), rows=(lo, hi))
            break
        dz = bn_relu_backward(gx, y, mi, fs, X.ROWS_FUSION, fs.fusion_row_w, _grad_of(b))
        if i == 0:
            _hip.mark_tail(dev)          # the step's last weight gradient follows: everything else of the bucket may go out (parallel.py)
        _linear_wgrad_side(x, dz, w)
        if i > 0:
            gx = _rows_dgrad(dz, w.reshape(w.shape[0], -1))          # dL/dyhat of layer i - 1
