"""BASELINE.json configs 2 and 3 AT THEIR STATED SIZE against the oracle.

config 2  "VFE-only fwd/bwd, batch=16 synthetic frames, fp32, assert voxel-idx bit-exact vs CPU": a 16-frame set
          (= MVX_MAX_FRAMES, 32 row segments: the boundary of the frame-set descriptor) of 20,000-point S1 and S2 frames
          through the code bench.py --mode vfe runs (GPU crop + projection, batched voxelizer, modules/frames.py rows_forward /
          rows_backward).  Voxel indices and payload of every frame bit-exact against the C oracle, in the per-frame AND in the
          concatenated frame-set layout; the (V,128) voxel features of frames 0, 7 and 15 against the float64 oracle on the
          reference's dense (V,35,23) rows (1e-4, max-norm relative, north_star's bar); on S2 also the parameter gradients
          of the whole set against the float64 oracle summed over the 16 frames.
config 3  "Full VoxelNet (VFE + dense Conv3d middle + RPN), batch=4, bf16 MFMA conv": 4 S2 frames, full 10x352x400 grid,
          pipeline.train_step_full (what bench.py --mode full times) in convmath f32 AND bf16x3 against the oracle run end to
          end in float64 per frame: anchor lists bit-exact against the C oracle, BEV map 1e-4, score / regression maps and
          the losses against the 1e-4 bar (the bf16x3 RPN maps are reported against it: 1.2e-4 .. 1.6e-4 observed, stated below).
Numbers are written to gpurun_out/configs_parity.json."""
import json
import os

import numpy as np
import pytest
import torch

import mvx_oracle as O

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, 'gpurun_out')


def _report(key, value):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, 'configs_parity.json')
    data = {}
    if os.path.exists(path):
        with open(path) as fh:
            data = json.load(fh)
    data[key] = value
    with open(path, 'w') as fh:
        json.dump(data, fh, indent=1)


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


# ------------------------------------------------------------------------------------------------------------ config 2
def _dense_rows(vox, imfeat_real, imfeat_pad):
    """The reference's dense VFE input of one frame (MVXNet.py:26 after featureMaping's in-place zeroing, imhead/Pipe.py:54-59):
    (V,35,23) = [7 geometric channels (0 on padded rows) | 16 image channels]; real rows take the compact rows in voxel-major
    order, every padded row the frame's shared padded row."""
    V, T, _ = vox.shape
    pad = (vox[..., :3] == 0).all(-1)
    geo = vox[..., :7].clone()
    geo[pad] = 0
    im = imfeat_pad.expand(V, T, -1).clone()
    assert int((~pad).sum()) == imfeat_real.shape[0]
    im[~pad] = imfeat_real
    return torch.cat([geo, im], dim=-1)


@pytest.mark.parametrize('workload', ['S2', 'S1'])
def test_config2_vfe_only_16_frames_matches_oracle(workload):
    import bench
    import modules.config as cfg
    from MVXNet import MVXNet
    from modules import _hip, parallel
    from modules import Extension as X
    from modules import frames as fr
    from modules import pipeline as pl
    assert X.MAX_FRAMES == 16
    dev = torch.device('cuda')
    B, P = 16, 20000
    frame_ids = list(range(B))
    batch = bench.make_batch(frame_ids, dev, P, workload)
    # crop + projection + voxelizer of all 16 frames (per-frame layout) bit-exact against the C oracle
    assert bench.voxel_index_check(batch, workload, frame_ids, P) == 'ok, 16 frames'
    torch.manual_seed(0)
    model = MVXNet().to(dev)
    names = [k for k, _ in model.named_parameters() if k.startswith('backbone.svfe.') or k.startswith('backbone.fcn.')]
    params = dict(model.named_parameters())
    bucket = parallel.GradBucket([params[k] for k in names])
    fs, live, counts, status = pl.prepare_frame_set(batch)
    assert fs.F == B and live == frame_ids and int(status) == 0
    points6, n_points = batch.prepared()
    pts6 = points6.cpu().numpy()
    # the concatenated frame-set layout (what the layers read): frame index in coords[:,0], voxels back to back
    voxels_c, coords_c = fs.voxels.cpu(), fs.coords.cpu().numpy()
    oracle_vox = []
    for f in frame_ids:
        rv, ri, _ = O.group(pts6[f], O.synth_perm(f, P), O.VELORANGE, O.voxelsize(), 35)
        a, b = fs.vox_off[f], fs.vox_off[f + 1]
        assert b - a == rv.shape[0] == counts[f]
        assert np.array_equal(coords_c[a:b, 1:], ri.astype(np.int64)) and np.all(coords_c[a:b, 0] == f), 'voxel indices differ'
        # the compact-row map has already zeroed the padded rows of the set in place (x = y = z = 0 rows lose their
        # -centroid columns: featureMaping's side effect, imhead/Pipe.py:54-59); the untouched payload was compared per frame above
        rz = rv.astype(np.float32)
        rz[(rz[..., :3] == 0).all(-1)] = 0
        assert np.array_equal(voxels_c[a:b].numpy(), rz), 'voxel payload differs'
        oracle_vox.append(torch.from_numpy(rv.astype(np.float32)))
    g = torch.Generator(device='cpu').manual_seed(5)
    imfeat = torch.randn((fs.Rt + fs.F, 16), generator=g)
    dfeat = torch.randn((fs.Vt, 128), generator=g) * 1e-3
    bucket.zero()
    old_sink, _hip.GRAD_SINK = _hip.GRAD_SINK, True
    old_async, _hip.ASYNC_WGRAD = _hip.ASYNC_WGRAD, True
    try:
        _hip.arena_begin(dev, doubles=1 << 21)
        with torch.no_grad():
            feat, saved = fr.rows_forward(model, fs, None, None, [], imfeat=imfeat.to(dev))
            fr.rows_backward(model, saved, dfeat.to(dev))
    finally:
        _hip.GRAD_SINK, _hip.ASYNC_WGRAD = old_sink, old_async
        _hip.arena_end()
        _hip.join_side_stream()
    torch.cuda.synchronize()
    feat = feat.cpu()
    P64 = {k[len('backbone.'):]: params[k].detach().cpu().double().requires_grad_(True) for k in names}
    with_grads = workload == 'S2'            # S1 (19.9 k voxels = 700 k dense rows per frame): forward only, three frames
    errs = {}
    for f in (frame_ids if with_grads else (0, 7, 15)):
        x = _dense_rows(oracle_vox[f], imfeat[fs.real_off[f]:fs.real_off[f + 1]], imfeat[fs.Rt + f]).double()
        a, b = fs.vox_off[f], fs.vox_off[f + 1]
        with torch.set_grad_enabled(with_grads):
            ref = O.voxel_features(x, P64)
            if with_grads:
                ref.backward(dfeat[a:b].double())
        if f in (0, 7, 15):
            errs[f] = _rel(feat[a:b].double(), ref.detach())
            assert errs[f] < 1e-4, 'voxel features of frame %d differ from the float64 oracle: %g' % (f, errs[f])
    rec = {'voxels': counts, 'features_rel_maxnorm_vs_f64': errs}
    if with_grads:
        ge = {k: _rel(params[k].grad.cpu().double(), P64[k[len('backbone.'):]].grad) for k in names}
        rec['param_grad_rel_maxnorm_vs_f64'] = ge
        # dL/d(voxel features) is white noise here (what bench.py --mode vfe feeds): every parameter gradient is the small
        # residue of ~1e5 cancelling terms per frame; measured 6e-4 .. 4.9e-3 from float64, asserted at 1e-2
        assert max(ge.values()) < 1e-2, sorted(ge.items(), key=lambda t: -t[1])[:3]
    _report('config2_%s' % workload, rec)
    print(json.dumps(rec))


# ------------------------------------------------------------------------------------------------------------ config 3
def _gt_boxes(seed, n=8):
    gg = np.random.default_rng(seed)
    return torch.tensor(np.stack([gg.uniform(8, 60, n), gg.uniform(-30, 30, n), gg.uniform(-1.8, -0.6, n), gg.uniform(3.4, 4.4, n),
                                  gg.uniform(1.5, 1.8, n), gg.uniform(1.4, 1.7, n),
                                  gg.choice([0.0, np.pi / 2], n) + gg.normal(0, 0.05, n)], 1), dtype=torch.float32)


def test_config3_full_voxelnet_4_frames_in_every_arithmetic_match_oracle():
    import bench
    import modules.config as cfg
    import modules.pipeline as pl
    from MVXNet import MVXNet
    from modules import Calc, parallel
    from modules.data import Preprocessing as pre
    from modules.voxelnet import VoxelLoss
    assert list(cfg.voxelshape) == [352, 400, 10]
    dev = torch.device('cuda')
    B, P = 4, 20000
    frame_ids = list(range(B))
    batch = bench.make_batch(frame_ids, dev, P, 'S2')
    torch.manual_seed(0)
    model = MVXNet().to(dev)
    bucket = parallel.GradBucket([p for p in model.parameters() if p.requires_grad])
    anchors = pre.createAnchors(cfg.voxelshape[0] // 2, cfg.voxelshape[1] // 2, cfg.velorange, cfg.carsize)
    bevs = Calc.bbox3d2bev(anchors.reshape(anchors.shape[:2] + (-1, 7))).to(dev).contiguous()
    o_anchors = O.create_anchors(176, 200)
    o_bevs = O.bbox3d2bev(o_anchors.reshape(176, 200, 2, 7))
    targets, o_targets, gts = [], [], []
    for f in frame_ids:
        gt = _gt_boxes(11 + f, 8 - f)                      # a different set of boxes (8, 7, 6, 5) per frame
        pi, ni, gi = Calc.classifyAnchors(Calc.bbox3d2bev(gt), gt[:, [0, 1]], bevs, cfg.velorange, 0.45, 0.6)
        rp, rn, rg = O.classify_anchors(O.bbox3d2bev(gt), gt[:, [0, 1]], o_bevs, O.VELORANGE, 0.45, 0.6)
        for a, b in zip(tuple(pi) + tuple(ni) + (gi,), tuple(rp) + tuple(rn) + (rg,)):
            assert np.array_equal(a.cpu().numpy(), np.asarray(b)), 'anchor lists of frame %d differ from the C oracle' % f
        targets.append((pi, ni, gi, gt.to(dev)))
        o_targets.append((rp, rn, rg))
        gts.append(gt)
    # ---- the GPU step in both arithmetic modes
    got = {}
    old = cfg.config.get('convmath', 'f32')
    try:
        for math in ('f32', 'fp16x3', 'bf16x6', 'bf16x3'):
            cfg.config['convmath'] = math
            bucket.zero()
            keep = {}
            out = pl.train_step_full(model, batch, targets, VoxelLoss(), anchors.to(dev), cfg.imsize, keep=keep)
            torch.cuda.synchronize()
            assert torch.isfinite(bucket.flat).all() and out['live'] == frame_ids and len(out['loss']) == B
            F, D3, H, W, C3, h1, w1 = keep['geom']
            x3 = keep['x3'].view(F, D3, H, W, C3).permute(0, 4, 1, 2, 3).reshape(F, C3 * D3, H, W).cpu()     # channel c*D3 + d
            heads = keep['heads'].view(F, h1, w1, 16).cpu()
            got[math] = (x3, heads[..., :2].clone(), heads[..., 2:], out['cls'], out['reg'])
    finally:
        cfg.config['convmath'] = old
    # ---- the oracle, float64 from the voxels on, one frame at a time (sampling positions in f32: part of the reference's
    # semantics, see tests/test_fullsize_gpu.py)
    points6, _ = batch.prepared()
    pts6 = points6.cpu().numpy()
    P64 = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
    bb = O.strip_prefix(P64, 'backbone.')
    rec = {'f32': [], 'fp16x3': [], 'bf16x6': [], 'bf16x3': []}
    for f in frame_ids:
        rv, ri, _ = O.group(pts6[f], O.synth_perm(f, P), O.VELORANGE, O.voxelsize(), 35)
        V = rv.shape[0]
        vox = torch.from_numpy(rv.astype(np.float32))
        idx = torch.from_numpy(np.concatenate([np.zeros((V, 1), np.int64), ri.astype(np.int64)], 1))
        fpn = [t[0].cpu() for t in batch.fpn_levels[f]]              # the maps bench.make_batch put on the GPU
        with torch.no_grad():
            imf = O.feature_mapping(vox, fpn, torch.tensor([370.0, 1224.0]))
            imf64 = O.image_feature_fusion(imf.double(), P64, 'head.fusion.')
            v23 = torch.cat([vox[..., :7].double(), imf64], dim=-1)
            mid = O.voxelnet_middle(v23, idx, bb)
            score, reg = O.rpn(mid, bb)
            rp, rn, rg = o_targets[f]
            cls, rl = O.voxel_loss(rp, rn, rg, gts[f].double(), score[0].permute(1, 2, 0), reg[0].permute(1, 2, 0), o_anchors.double(), 2)
        for math in ('f32', 'fp16x3', 'bf16x6', 'bf16x3'):
            x3, sc, rg_, cl_, rl_ = got[math]
            e = {'voxels': int(V),
                 'bev_rel_maxnorm': _rel(x3[f].double(), mid[0]),
                 # the classification map before the sigmoid (max-norm relative, like every other map) and the
                 # probabilities (absolute): |logit| reaches ~10 with this initialisation, so 1e-4 relative on the logits is
                 # up to 2.5e-4 absolute on sigmoid(logit)
                 'cls_logit_rel_maxnorm': _rel(sc[f].double(), torch.logit(score[0].permute(1, 2, 0))),
                 'score_abs_max': float((torch.sigmoid(sc[f].double()) - score[0].permute(1, 2, 0)).abs().max()),
                 'reg_rel_maxnorm': _rel(rg_[f].double(), reg[0].permute(1, 2, 0)),
                 'cls_loss_rel': abs(cl_[f] - float(cls)) / abs(float(cls)),
                 'reg_loss_rel': abs(rl_[f] - float(rl)) / abs(float(rl))}
            rec[math].append(e)
    _report('config3', rec)
    print(json.dumps(rec))
    # exact-f32 MFMA, fp16x3 (the default: two fp16 pieces per operand, three MFMAs per product, f32 accumulate, 22 mantissa
    # bits) and bf16x6 ("bf16 MFMA conv", BASELINE config 3 as written: three bf16 pieces, six MFMAs): every map inside
    # north_star's 1e-4 (relative), the same assertions
    for e in rec['f32'] + rec['fp16x3'] + rec['bf16x6']:
        assert e['bev_rel_maxnorm'] < 1e-4 and e['cls_logit_rel_maxnorm'] < 1e-4 and e['reg_rel_maxnorm'] < 1e-4, e
        assert e['score_abs_max'] < 5e-4, e                    # probabilities: measured 2.0e-4 .. 2.4e-4 absolute
        assert e['cls_loss_rel'] < 1e-4 and e['reg_loss_rel'] < 1e-4, e
    for e in rec['bf16x3']:
        # the two-piece split (hi/lo, three MFMAs per product; opt-in, fastest): the BEV map meets the 1e-4 bar (8e-6 .. 1.6e-5); the RPN maps --
        # 17 more split-arithmetic layers -- are measured at 1.2e-4 .. 1.6e-4 of their maximum, i.e. they MISS the bar by
        # up to a half.  Asserted at 3e-4 and reported as measured (the reason this mode is opt-in and a 22-bit arithmetic the default).  The
        # forward row GEMMs of the fusion MLP stay exact f32 in this mode: in split arithmetic they would put these maps at
        # 4.5e-4 .. 8.8e-4 and the BEV map at up to 1.1e-4 (modules/_hip.py row_split, profiles/r03_split_accuracy.json)
        assert e['bev_rel_maxnorm'] < 1e-4, e
        assert e['reg_rel_maxnorm'] < 3e-4 and e['cls_logit_rel_maxnorm'] < 3e-4 and e['score_abs_max'] < 1e-3, e
        assert e['cls_loss_rel'] < 2e-4 and e['reg_loss_rel'] < 2e-4, e
