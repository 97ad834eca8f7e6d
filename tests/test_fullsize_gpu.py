"""Full-size parity of the BENCHMARKED path: S2 ring frames (120,000 raw points -> 20,000 after the crops, grid
10x352x400, fusion on) through modules.pipeline.train_step_frame_set -- GPU crop + projection, batched voxelizer, ONE launch
per layer for all frames of the step (modules/frames.py), restricted backward, side-stream weight gradients, exactly what
bench.py times -- against the CPU oracle's forward + backward of the same frames.

Asserted: voxel indices and payload bit-exact; the middle map within 1e-4 (max-norm relative) of the f32 oracle AND of a
float64 run of the oracle (the yardstick; measured: HIP 6e-6, torch-CPU f32 4.6e-5 from it); element-wise, relative to
max(|value|, rms of the map), the HIP path stays below 1e-3 of the yardstick (the f32 oracle itself reaches 2.8e-3 on
near-constant channels whose BatchNorm divides by a tiny deviation).  Parameter gradients: the upstream gradient of the
benchmark is white noise, so every parameter gradient is the small residue of 1.4 M cancelling terms and fp32 -- torch-CPU's
as much as this path's -- is only good to about 1e-2 there (tools/fullsize_grad_check.py prints HIP / f32 oracle / f64 side
by side); the test requires the HIP gradients to be within 3e-2 of the float64 yardstick, and the two-frame set to equal
the sum of its frames run one at a time (frame sets of one frame, and the per-frame executor modules/tape.py).  The
distributions are written to gpurun_out/fullsize_parity.json."""
import json
import os

import numpy as np
import pytest
import torch

import mvx_oracle as O

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host_projection(pts):
    m = O.KITTI_CALIB['R0_rect'].astype(np.float32) @ O.KITTI_CALIB['Tr_velo_to_cam'].astype(np.float32)
    p = np.ones((4, pts.shape[0]), np.float32)
    p[:3] = pts[:, :3].T
    img = O.KITTI_CALIB['P2'].astype(np.float32) @ (m @ p)
    return (img[:2] / img[2]).T[:, ::-1].astype(np.float32)


def _percentiles(err):
    q = np.quantile(err, [0.5, 0.9, 0.99, 0.999, 1.0])
    return {'p50': float(q[0]), 'p90': float(q[1]), 'p99': float(q[2]), 'p999': float(q[3]), 'max': float(q[4])}


def _make_batch(frame_ids, dev, P_pts=20000):
    """Raw clouds resident on the GPU, as bench.py builds them."""
    from modules.data import Synthetic as S
    from modules.pipeline import FrameBatch
    n = len(frame_ids)
    raw = np.zeros((n, 120000, 4), np.float32)
    perms = np.zeros((n, P_pts), np.int32)
    fpn_cpu, kept = [], []
    for k, fid in enumerate(frame_ids):
        pc = O.synth_ring(fid, P_pts)
        assert pc.shape[0] == P_pts
        raw[k] = S.synth_raw_around(pc, fid, 120000)
        kept.append(pc)
        perms[k] = O.synth_perm(fid, P_pts)
        fpn_cpu.append([torch.from_numpy(f) for f in O.synth_fpn(fid)])
    fpn_dev = [[f[None].to(dev).contiguous(memory_format=torch.channels_last) for f in lv] for lv in fpn_cpu]
    batch = FrameBatch(None, torch.from_numpy(perms).to(dev), None, fpn_dev, raw=torch.from_numpy(raw).to(dev),
                       calib=S.KITTI_CALIB, cap_points=P_pts)
    return batch, raw, kept, perms, fpn_cpu


def _smooth_gradient(C=128, H=352, W=400):
    """dL/d(BEV map) (1,C,H,W): per channel an offset plus a product of low-frequency cosines, magnitude 1e-3."""
    g = np.random.default_rng(123)
    hh = np.arange(H, dtype=np.float64)[None, :, None] / H
    ww = np.arange(W, dtype=np.float64)[None, None, :] / W
    fa, fb = g.integers(0, 3, (C, 1, 1)), g.integers(0, 3, (C, 1, 1))
    ph = g.uniform(0, 2 * np.pi, (2, C, 1, 1))
    off = g.uniform(-0.5, 0.5, (C, 1, 1))
    pat = off + np.cos(2 * np.pi * fa * hh + ph[0]) * np.cos(2 * np.pi * fb * ww + ph[1])
    return torch.from_numpy((1e-3 * pat).astype(np.float32))[None]


def test_bench_path_matches_oracle_at_full_size():
    import modules.config as cfg
    import modules.pipeline as pl
    from MVXNet import MVXNet
    from modules import parallel
    from modules.pipeline import train_step_frame_set, train_step_frames
    assert pl.BATCHED and pl.PREP_STREAM, 'this test pins the default (benchmarked) execution mode'
    assert list(cfg.voxelshape) == [352, 400, 10]
    dev = torch.device('cuda')
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    batch, raw, kept, perms, fpn_cpu = _make_batch((0, 1), dev)
    torch.manual_seed(0)
    model = MVXNet().to(dev)
    hot = [(k, p) for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
    bucket = parallel.GradBucket([p for _, p in hot])
    g = torch.Generator(device='cpu').manual_seed(77)
    G = torch.randn((1, 128, 352, 400), generator=g) * 1e-3
    imsize = [370.0, 1224.0]

    def run(b, keep=None, fn=train_step_frame_set, g_up=None):
        bucket.zero()
        nv, st = fn(model, b, (G if g_up is None else g_up).to(dev), imsize, keep_mid=keep)
        torch.cuda.synchronize()
        assert int(torch.stack([s_.reshape(()) for s_ in st]).max()) == 0
        return nv, {k: p.grad.detach().clone() for k, p in hot}

    mids = []
    nvox, grads_both = run(batch, mids)                       # the benchmarked step: both frames in one frame set
    mids_again = []
    _, grads_again = run(batch, mids_again)
    for k, _ in hot:                                          # the step is reproducible bit for bit
        assert torch.equal(grads_both[k], grads_again[k]), k
    singles = [_make_batch((fid,), dev)[0] for fid in (0, 1)]
    single, single_mids = [], []
    for b in singles:                                         # frame sets of one frame
        m = []
        single.append(run(b, m)[1])
        single_mids.append(m[0])
    # Forward: the set equals its frames run alone up to the last bits of the BatchNorm sums.  Backward: a handful of the
    # 18 M conv3 outputs sit within 1e-7 of the ReLU kink, flip their mask between two evaluations that differ in the
    # last bit, and each flip moves a weight gradient (a random-walk sum of N white-noise terms, magnitude sqrt(N)) by one
    # term, i.e. ~1e-3; measured 2e-4 .. 1.1e-2 over the parameters (tools/frameset_diag.py).  Same order as the distance
    # of ANY fp32 evaluation -- torch-CPU's included -- from the float64 yardstick below.
    for f in range(2):
        assert float((mids[f] - single_mids[f]).abs().max() / single_mids[f].abs().max()) < 1e-5
    set_consistency = {k: float((grads_both[k] - (single[0][k] + single[1][k])).abs().max() / grads_both[k].abs().max())
                       for k, _ in hot}
    assert max(set_consistency.values()) < 3e-2, sorted(set_consistency.items(), key=lambda t: -t[1])[:3]
    tape_mid = []
    tape0 = run(singles[0], tape_mid, fn=train_step_frames)[1]          # the per-frame executor (modules/tape.py) on frame 0
    assert float((tape_mid[0] - single_mids[0]).abs().max() / single_mids[0].abs().max()) < 1e-5
    tape_consistency = {k: float((tape0[k] - single[0][k]).abs().max() / tape0[k].abs().max()) for k, _ in hot}
    assert max(tape_consistency.values()) < 3e-2, sorted(tape_consistency.items(), key=lambda t: -t[1])[:3]

    # ---- stage 1 of the path: crop + cropToSight + lidar2Img on the GPU against the oracle (a1-a3)
    points6, n_points = batch.prepared()
    assert n_points.tolist() == [20000, 20000]
    pts6 = points6.cpu().numpy()
    for k in range(2):
        ref = O.crop_to_sight(O.crop(raw[k], O.VELORANGE), O.KITTI_CALIB, (1224, 370))
        assert np.array_equal(ref, kept[k]) and np.array_equal(pts6[k, :, :4], ref), 'crop differs from the oracle'
        proj = O.lidar2img(ref, O.KITTI_CALIB, np.float32)[:, ::-1]
        np.testing.assert_allclose(pts6[k, :, 4:], proj, rtol=2e-5, atol=2e-3)
    # the oracle continues from the GPU's projected pixels: a last-bit difference of the f32 projection may move a point
    # across a pixel boundary of featureMaping's trunc(), which is a property of the input, not of the path under test
    res = pl._hip.voxelize(points6, batch.perms, n_points, cfg.velorange[0:3], cfg.voxelsize, 35, 9)

    P32 = {k: v.detach().cpu() for k, v in model.state_dict().items() if '.rpn.' not in k}
    report = {'frames': [], 'frame_set_vs_single_frames': max(set_consistency.values()),
              'frame_set_vs_per_frame_executor': max(tape_consistency.values())}
    for k in range(2):
        rv, ri, _ = O.group(pts6[k], perms[k], O.VELORANGE, O.voxelsize(), 35)
        V = rv.shape[0]
        assert nvox[k] == V
        assert np.array_equal(res.coords[k, :V, 1:].cpu().numpy(), ri.astype(np.int64)), 'voxel indices differ'
        assert np.array_equal(res.voxels[k, :V].cpu().numpy(), rv.astype(np.float32)), 'voxel payload differs'
        vox = torch.from_numpy(rv.astype(np.float32))
        idx = torch.from_numpy(np.concatenate([np.zeros((V, 1), np.int64), ri.astype(np.int64)], 1))
        with torch.no_grad():
            v23 = O.mvx_point_features(vox.clone(), fpn_cpu[k], torch.tensor(imsize), P32)
            ref = O.voxelnet_middle(v23, idx, O.strip_prefix(P32, 'backbone.'))
        got = mids[k].cpu()
        err = float((got - ref).abs().max() / ref.abs().max())
        diff = (got - ref).abs().reshape(-1).numpy()
        mag = ref.abs().reshape(-1).numpy()
        rms = float(np.sqrt(np.mean(mag.astype(np.float64) ** 2)))
        report['frames'].append({'voxels': int(V), 'mid_rel_maxnorm_vs_oracle_f32': err, 'mid_abs_max': float(mag.max()),
                                 'mid_rms': rms, 'mid_abs_err_max': float(diff.max()),
                                 'mid_elementwise_rel_floor_rms': _percentiles(diff / np.maximum(mag, rms)),
                                 'mid_elementwise_rel_floor_1e-3rms': _percentiles(diff / np.maximum(mag, 1e-3 * rms))})
        assert err < 1e-4, 'middle map differs from the oracle: %g' % err
        if k != 0:
            continue
        # ---- float64 yardstick, frame 0: forward + backward (~30 s of host time).  The sampling positions
        # (trunc of proj / region - eps, imhead/Pipe.py:62-65) are part of the reference's f32 semantics -- in f64 a few
        # points per frame fall into the neighbouring pixel -- so the yardstick samples in f32 and is exact from there on
        P64 = {n: v.double().clone().requires_grad_(True) for n, v in P32.items()}
        vz = vox.clone()
        imf = O.feature_mapping(vz, fpn_cpu[k], torch.tensor(imsize))        # zeroes the padded rows of vz in place
        imf64 = O.image_feature_fusion(imf.double(), P64, 'head.fusion.')
        v23_64 = torch.cat([vz[..., :7].double(), imf64], dim=-1)
        mid64 = O.voxelnet_middle(v23_64, idx, O.strip_prefix(P64, 'backbone.'))
        # (a) the TIGHT gradient check -- float64 with the HIP forward's ReLU masks and max rows, 1e-4 per parameter in every
        # arithmetic, with the 1 % mutation check of the restricted backward's closed forms -- is
        # test_hot_path_gradients_tight_with_shared_kinks_at_full_size below.  (Rounds 3 and 4 held a plain float64 comparison
        # under a smooth upstream gradient to 5e-3 .. 3e-2 here, bounds that had to move with every change of arithmetic because
        # they measured WHICH ReLUs flip, not the backward; removed in favour of the tight form.)
        # (b) the benchmark's white-noise upstream gradient: the reported worst case (every gradient the residue of 1.4 M
        # cancelling terms)
        mid64.backward(G.double())
        ref64 = mid64.detach().reshape(-1).numpy()
        mag64 = np.maximum(np.abs(ref64), rms)
        e_hip = np.abs(got.reshape(-1).numpy().astype(np.float64) - ref64) / mag64
        e_ref = np.abs(ref.reshape(-1).numpy().astype(np.float64) - ref64) / mag64
        y = {'hip': _percentiles(e_hip), 'oracle_f32': _percentiles(e_ref),
             'hip_rel_maxnorm': float(np.abs(got.reshape(-1).numpy() - ref64).max() / np.abs(ref64).max()),
             'oracle_f32_rel_maxnorm': float(np.abs(ref.reshape(-1).numpy() - ref64).max() / np.abs(ref64).max())}
        gr = {n: float((single[0][n].cpu().double() - P64[n].grad).abs().max() / P64[n].grad.abs().max()) for n, _ in hot}
        report['frames'][-1]['vs_float64'] = y
        report['param_grad_rel_maxnorm_vs_float64'] = gr
        with open(os.path.join(REPO, 'gpurun_out', 'fullsize_parity.json'), 'w') as fh:
            json.dump(report, fh, indent=1)
        assert y['hip_rel_maxnorm'] < 1e-4
        assert e_hip.max() < 1e-3, e_hip.max()
        assert max(gr.values()) < 3e-2, sorted(gr.items(), key=lambda t: -t[1])[:4]
    print(json.dumps(report))


# ---- the hot path's gradients, TIGHT: float64 with the HIP forward's kinks --------------------------------------------
_HOT_GRAD_BOUND = 1e-4          # 2-norm relative distance of every parameter gradient from float64 (shared ReLU masks and max rows)
_HOT_MUTATIONS = ('A2', 'A1', 'inact2', 'T3', 'T2')


class _SharedKinks:
    """Stands for torch.nn.functional inside the oracle: relu(x) = x * (the mask the HIP forward took at that layer), in call
    order; everything else is torch's."""

    def __init__(self, masks):
        import torch.nn.functional as Fn
        self._F, self.masks = Fn, list(masks)

    def relu(self, x):
        m = self.masks.pop(0)
        if m.dim() == 3 and x.dim() == 4:                 # the 1x1 CRB2d layers evaluate rows as a (1, C, V, T) image (imhead/Pipe.py:97-99)
            m = m.permute(2, 0, 1)[None]
        assert m.shape == x.shape, (m.shape, x.shape)
        return x * m

    def __getattr__(self, name):
        return getattr(self._F, name)


@pytest.mark.parametrize('math', ['bf16x6', 'f32', 'fp16x3'])
def test_hot_path_gradients_tight_with_shared_kinks_at_full_size(math):
    """the flip-free gradient check of the hot path (fusion MLP -> VFE -> conv1-3: imhead/Pipe.py:84-104,
    voxelnet/Pipe.py:5-43, VoxelNet.py:16-36) that the RPN chain got in round 4.

    Eleven Linear/Conv -> ReLU -> BatchNorm layers and three maxima over T are chaotic under kink flips: a pre-activation within
    1e-7 of zero (or two rows within 1e-7 of each other under a max) lands on different sides in fp32 and float64, and ONE flip
    moves every gradient below it by 1e-3 .. 1e-2 -- which is why the plain comparison of this file needs bounds of 5e-3 .. 3e-2
    and why those bounds moved with every change of arithmetic.  Here the float64 oracle (oracle/mvx_oracle.py, the reference's
    dense formulation, autograd) runs with the kinks of the HIP forward: relu(x) = x * [y_hip > 0] for every layer (masks
    expanded from the compact rows / taken from the saved conv outputs) and max over T = the row the HIP kernel picked.  The
    forward values then differ from plain float64 only at the flipped elements (by < 1e-6; BEV map asserted at 1e-4), and what
    remains in the backward is the ARITHMETIC of every kernel on the path -- compact rows with weighted BatchNorm sums, the
    closed forms of the restricted CML backward, the input-sparse first layer, the split matrix products -- held to 1e-4
    (2-norm, per parameter) in EVERY arithmetic, with the benchmark's white-noise upstream gradient and with a smooth one.
    Mutation check: each closed-form term of the restricted backward scaled by 1.01 (frames._MUTATE) must leave the bound by
    at least 3 x."""
    import modules.config as cfg
    import modules.pipeline as pl
    from MVXNet import MVXNet
    from modules import _hip, parallel
    from modules import frames as fr
    dev = torch.device('cuda')
    old_math, cfg.config['convmath'] = cfg.config.get('convmath', 'f32'), math
    old_poison = os.environ.get('MVX_POISON_BG')
    os.environ['MVX_POISON_BG'] = '1'               # voxel-free tiles of conv1's output are never written: NaN marks them
    try:
        batch, raw, kept, perms, fpn_cpu = _make_batch((0,), dev)
        torch.manual_seed(0)
        model = MVXNet().to(dev)
        hot = [(k, p) for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
        bucket = parallel.GradBucket([p for _, p in hot])
        imsize = [370.0, 1224.0]
        g = torch.Generator(device='cpu').manual_seed(77)
        G_noise = torch.randn((1, 128, 352, 400), generator=g) * 1e-3
        G_smooth = _smooth_gradient()
        fs, live, counts, status = pl.prepare_frame_set(batch)
        assert live == [0]

        def hip_step(g_up):
            """forward + backward of the frame-set executor on the prepared set; returns (mid, saved state, gradients)."""
            bucket.zero()
            model.prepack()
            old_sink, _hip.GRAD_SINK = _hip.GRAD_SINK, True
            _hip.arena_begin(dev, doubles=1 << 21)
            try:
                with torch.no_grad():
                    st = []
                    mid, S = fr.middle_forward(model, fs, [batch.fpn_levels[0]], imsize, st)
                    fr.middle_backward(model, S, g_up.to(dev))
            finally:
                _hip.GRAD_SINK = old_sink
                _hip.arena_end()
                _hip.join_side_stream()
            torch.cuda.synchronize()
            assert int(torch.stack([s_.reshape(()) for s_ in st]).max()) == 0
            return mid, S, {k: p.grad.detach().cpu().double().clone() for k, p in hot}

        mid, S, g_noise = hip_step(G_noise)
        # ---- the kinks of the HIP forward, in the oracle's dense layouts and call order
        V, T = fs.Vt, fs.T
        cnt = fs.vcnt.cpu().long()
        voff = fs.voff.cpu().long()
        t_idx = torch.arange(T)[None, :].expand(V, T)
        real = t_idx < cnt[:, None]
        rows_vfe = torch.where(real, voff[:, None] + t_idx, (fs.Rt + torch.arange(V))[:, None].expand(V, T))      # [real | pad per voxel]
        rows_fus = torch.where(real, voff[:, None] + t_idx, torch.full((V, T), fs.Rt))                          # [real | one pad row]
        masks = [(rec[3] > 0).cpu()[rows_fus].double() for rec in S.fusion]                                    # fusion MLP: 5 layers
        masks += [(rec[3] > 0).cpu()[rows_vfe].double() for rec in S.vfe]                                      # VFE 1, VFE 2
        masks += [(S.head[3] > 0).cpu()[rows_vfe].double()]                                                    # FCN(128)
        y1 = S.conv1['y'].cpu()
        b1 = S.conv1['b'].detach().cpu()
        m1 = torch.where(torch.isnan(y1), (b1 > 0)[None, None, None, :].expand_as(y1), y1 > 0)                 # unwritten tiles: ReLU(bias)
        masks += [m1.permute(3, 0, 1, 2)[None].double()]
        masks += [(rec['y'] > 0).cpu().permute(3, 0, 1, 2)[None].double() for rec in S.convs]
        arg = [rec[5].cpu().long() for rec in S.vfe] + [S.head[5].cpu().long()]                                # row the max took: local t
        assert all(int(a.max()) < T for a in arg)

        # ---- float64 oracle with those kinks (sampling in f32 like the plain yardstick above: part of the reference's f32 semantics)
        vox = fs.voxels.cpu().clone()           # (the padded rows are already zeroed in place, imhead/Pipe.py:54-59; the oracle does the same)
        idx = fs.coords.cpu()
        P32 = {k: v.detach().cpu() for k, v in model.state_dict().items() if '.rpn.' not in k}
        P64 = {n: v.double().clone().requires_grad_(True) for n, v in P32.items()}
        vz = vox.clone()
        imf = O.feature_mapping(vz, fpn_cpu[0], torch.tensor(imsize))
        old_F, old_vfe, old_vf = O.F, O.vfe, O.voxel_features
        args = list(arg)

        def vfe_shared(x, w, b, eps=O.EPS):
            y = O.fcn(x, w, b, eps)
            s = y.gather(1, args.pop(0)[:, None, :]).expand(-1, y.shape[1], -1)
            return torch.cat([y, s], dim=-1)

        def voxel_features_shared(x, p, eps=O.EPS):
            x = O.svfe(x, p, 'svfe.', eps)
            x = O.fcn(x, p['fcn.fc.weight'], p['fcn.fc.bias'], eps)
            return x.gather(1, args.pop(0)[:, None, :])[:, 0]
        O.F, O.vfe, O.voxel_features = _SharedKinks(masks), vfe_shared, voxel_features_shared
        try:
            imf64 = O.image_feature_fusion(imf.double(), P64, 'head.fusion.')
            v23 = torch.cat([vz[..., :7].double(), imf64], dim=-1)
            mid64 = O.voxelnet_middle(v23, idx, O.strip_prefix(P64, 'backbone.'))
            assert not O.F.masks and not args
        finally:
            O.F, O.vfe, O.voxel_features = old_F, old_vfe, old_vf
        e_mid = float((mid.cpu().double() - mid64.detach()).abs().max() / mid64.detach().abs().max())

        def ref_grads(g_up, retain):
            for v in P64.values():
                v.grad = None
            mid64.backward(g_up.double(), retain_graph=retain)
            return {n: P64[n].grad.clone() for n, _ in hot}

        def dist(got, ref):
            return {n: float((got[n] - ref[n]).norm() / ref[n].norm()) for n, _ in hot}
        ref_noise = ref_grads(G_noise, True)
        ref_smooth = ref_grads(G_smooth, False)
        d_noise = dist(g_noise, ref_noise)
        d_smooth = dist(hip_step(G_smooth)[2], ref_smooth)
        report = {'convmath': math, 'mid_rel_maxnorm_vs_float64_shared_kinks': e_mid,
                  'grad_2norm_white_noise_upstream': d_noise, 'grad_2norm_smooth_upstream': d_smooth, 'mutations': {}}
        # ---- every closed-form term of the restricted backward mutated by 1 %: the worst tensor must leave the bound by 3 x
        try:
            for term in _HOT_MUTATIONS:
                fr._MUTATE = {term: 1.01}
                report['mutations'][term] = max(dist(hip_step(G_smooth)[2], ref_smooth).values())
        finally:
            fr._MUTATE = {}
        os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(REPO, 'gpurun_out', 'hot_path_grads_%s.json' % math), 'w') as fh:
            json.dump(report, fh, indent=1)
        print(json.dumps(report))
        assert e_mid < 1e-4
        assert max(d_noise.values()) < _HOT_GRAD_BOUND, sorted(d_noise.items(), key=lambda t: -t[1])[:4]
        assert max(d_smooth.values()) < _HOT_GRAD_BOUND, sorted(d_smooth.items(), key=lambda t: -t[1])[:4]
        assert min(report['mutations'].values()) > 3 * _HOT_GRAD_BOUND, report['mutations']
    finally:
        cfg.config['convmath'] = old_math
        if old_poison is None:
            os.environ.pop('MVX_POISON_BG', None)
        else:
            os.environ['MVX_POISON_BG'] = old_poison


def test_whole_model_losses_match_oracle_at_full_size():
    """The --mode full path (pipeline.train_step_full: frame sets up to the CML output, RPN and VoxelLoss on this library's
    kernels) on one full-size S2 frame against the oracle run end to end in float64 from the voxels on: target lists
    bit-identical to the C oracle's classifyAnchors, classification and regression loss within 1e-4 relative."""
    import modules.config as cfg
    import modules.pipeline as pl
    from MVXNet import MVXNet
    from modules import Calc, parallel
    from modules.data import Preprocessing as pre
    from modules.voxelnet import VoxelLoss
    dev = torch.device('cuda')
    batch, raw, kept, perms, fpn_cpu = _make_batch((0,), dev)
    torch.manual_seed(0)
    model = MVXNet().to(dev)
    bucket = parallel.GradBucket([p for p in model.parameters() if p.requires_grad])
    anchors = pre.createAnchors(cfg.voxelshape[0] // 2, cfg.voxelshape[1] // 2, cfg.velorange, cfg.carsize)
    bevs = Calc.bbox3d2bev(anchors.reshape(anchors.shape[:2] + (-1, 7)))
    gg = np.random.default_rng(11)
    n = 8
    gt = torch.tensor(np.stack([gg.uniform(8, 60, n), gg.uniform(-30, 30, n), gg.uniform(-1.8, -0.6, n), gg.uniform(3.4, 4.4, n),
                                gg.uniform(1.5, 1.8, n), gg.uniform(1.4, 1.7, n),
                                gg.choice([0.0, np.pi / 2], n) + gg.normal(0, 0.05, n)], 1), dtype=torch.float32)
    gbev = Calc.bbox3d2bev(gt)
    pi, ni, gi = Calc.classifyAnchors(gbev, gt[:, [0, 1]], bevs.to(dev).contiguous(), cfg.velorange, 0.45, 0.6)
    rp, rn, rg = O.classify_anchors(O.bbox3d2bev(gt), gt[:, [0, 1]], O.bbox3d2bev(O.create_anchors(176, 200).reshape(176, 200, 2, 7)),
                                    O.VELORANGE, 0.45, 0.6)
    for a, b in zip(tuple(pi) + tuple(ni) + (gi,), tuple(rp) + tuple(rn) + (rg,)):
        assert np.array_equal(a.cpu().numpy(), np.asarray(b)), 'anchor lists differ from the C oracle'
    bucket.zero()
    out = pl.train_step_full(model, batch, [(pi, ni, gi, gt.to(dev))], VoxelLoss(), anchors.to(dev), cfg.imsize)
    torch.cuda.synchronize()
    assert torch.isfinite(bucket.flat).all() and len(out['loss']) == 1 and len(out['reg']) == 1
    # ---- the oracle, float64 from the voxels on (sampling positions in f32: see the test above)
    points6, n_points = batch.prepared()
    pts6 = points6.cpu().numpy()
    rv, ri, _ = O.group(pts6[0], perms[0], O.VELORANGE, O.voxelsize(), 35)
    V = rv.shape[0]
    trainable = [k for k, p in model.named_parameters() if p.requires_grad]
    vox = torch.from_numpy(rv.astype(np.float32))
    idx = torch.from_numpy(np.concatenate([np.zeros((V, 1), np.int64), ri.astype(np.int64)], 1))
    with torch.no_grad():
        imf = O.feature_mapping(vox, fpn_cpu[0], torch.tensor([370.0, 1224.0]))

    def oracle(dt):
        """The whole model forward + the gradient of the REAL loss (train.py:146-161: clsLoss + regLoss) through RPN, CML,
        VFE and fusion in dtype ``dt``: a structured upstream gradient."""
        Pd = {k: v.detach().cpu().to(dt) for k, v in model.state_dict().items()}
        for k in trainable:
            Pd[k].requires_grad_(True)
        imfd = O.image_feature_fusion(imf.to(dt), Pd, 'head.fusion.')
        v23 = torch.cat([vox[..., :7].to(dt), imfd], dim=-1)
        bb = O.strip_prefix(Pd, 'backbone.')
        mid = O.voxelnet_middle(v23, idx, bb)
        score, reg = O.rpn(mid, bb)
        cls, rl = O.voxel_loss(rp, rn, rg, gt.to(dt), score[0].permute(1, 2, 0), reg[0].permute(1, 2, 0),
                               O.create_anchors(176, 200).to(dt), 2)
        (cls + rl).backward()
        return float(cls.detach()), float(rl.detach()), {k: Pd[k].grad.double() for k in trainable}

    cls, rl, g64 = oracle(torch.float64)
    cls32, rl32, g32 = oracle(torch.float32)           # the reference's own arithmetic (torch fp32), as the yardstick
    e_cls = abs(out['cls'][0] - float(cls)) / abs(float(cls))
    e_reg = abs(out['reg'][0] - float(rl)) / abs(float(rl))
    print('whole model at full size vs float64 oracle: clsLoss %.6f (rel %.1e), regLoss %.6f (rel %.1e)' % (out['cls'][0], e_cls, out['reg'][0], e_reg))
    assert e_cls < 1e-4 and e_reg < 1e-4
    grads = dict(zip(trainable, [p.grad.detach().cpu().double() for k, p in model.named_parameters() if p.requires_grad]))
    # Every parameter gradient of the step against float64, with the torch-fp32 oracle's distance from float64 beside it.
    # At random initialisation this network is badly conditioned for the loss gradient (BatchNorm without affine and
    # eps = 1e-6 multiplies channels that are almost always zero after the ReLU by up to 1000): the reference's own fp32
    # arithmetic lands 13..44 % (2-norm) from float64 on every parameter upstream of the heads and 3e-3 on the losses when run on
    # 8 threads (tools/grad_conditioning_cpu.py; the figure moves with the thread count); this path keeps its BatchNorm sums in
    # f64 and lands at 1..7 %, losses 1e-5.
    # (On the 16-thread GPU box the fp32 oracle lands at 2e-4 .. 2.2e-1, median 5e-2, this path at 4e-5 .. 6.8e-2, median 2e-2;
    # on a few parameters -- rpn.deconv3, fusion.fcn3.bias -- the two fp32 evaluations sit at the SAME distance from
    # float64, i.e. they agree with each other better than either does with exact arithmetic.)
    # Asserted: < 1e-1 everywhere AND never further from float64 than 1.25 x the reference's own fp32 arithmetic (or
    # < 1e-3 outright).  The tight check of
    # the closed-form backward terms is the smooth-gradient + mutation test above; every RPN layer is compared with its
    # float64 counterpart on identical inputs at 2e-6 in tests/test_rpn_gpu.py.
    ge, ge2, ge2_f32 = {}, {}, {}
    for k in trainable:
        ref = g64[k]
        ge[k] = float((grads[k] - ref).abs().max() / ref.abs().max())
        ge2[k] = float((grads[k] - ref).norm() / ref.norm())
        ge2_f32[k] = float((g32[k] - ref).norm() / ref.norm())
    worst = sorted(ge2.items(), key=lambda t: -t[1])[:5]
    print('whole-model parameter gradients vs float64 (loss-derived upstream gradient): worst 2-norm %s; the fp32 oracle: %.2e .. %.2e'
          % (worst, min(ge2_f32.values()), max(ge2_f32.values())))
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(REPO, 'gpurun_out', 'fullsize_whole_model_grads.json'), 'w') as fh:
        json.dump({'rel_maxnorm': ge, 'rel_2norm': ge2, 'rel_2norm_oracle_f32': ge2_f32, 'cls_loss_rel': e_cls, 'reg_loss_rel': e_reg,
                   'cls_loss_rel_oracle_f32': abs(cls32 - cls) / abs(cls), 'reg_loss_rel_oracle_f32': abs(rl32 - rl) / abs(rl)}, fh, indent=1)
    for k in trainable:
        assert ge2[k] < 1e-1 and (ge2[k] < 1e-3 or ge2[k] < 1.25 * ge2_f32[k]), (k, ge2[k], ge2_f32[k])
