"""The region proposal network on HIP kernels (modules/rpn_frames.py, csrc/rpn.hip, mvx_conv2d_*_frames) against the
reference fixture and the CPU oracle: score / regression maps and every parameter gradient, frame sets of 1 and 3."""
import os

import numpy as np
import pytest
import torch

import mvx_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(autouse=True, params=['32-channel units', '64-channel units'])
def gather_units(request):
    """The f32 gather runs small launches as 32-channel ("narrow") workgroup units (csrc/conv3d.hip launch_gather,
    MVX_TUNE_GATHER_NARROW_MAX_UNITS = key 2): every test of this file with narrow units forced and with narrow units off."""
    from modules import Extension as X
    X.check(X.lib.mvx_tuning_set(2, (1 << 60) if request.param.startswith('32') else 0), 'mvx_tuning_set')
    yield request.param
    X.check(X.lib.mvx_tuning_set(2, -1), 'mvx_tuning_set')


def test_narrow_and_wide_gather_units_give_identical_maps():
    """The same 2-D layers (stride 1; stride 2 in space-to-depth form, forward and input gradient) with 64- and 32-channel
    units: outputs bit-identical (the K order of an output element does not depend on the unit width), BatchNorm sums
    equal to f64 rounding (atomics in a different order)."""
    from modules import Extension as X
    from modules import _hip
    from modules import rpn_frames as rf
    import modules.config as cfg
    F, h, w = 3, 21, 37                                   # ragged: no multiple of the 8 x 16 tile
    g = torch.Generator().manual_seed(3)
    got = {}
    old_math, cfg.config['convmath'] = cfg.config.get('convmath', 'f32'), 'f32'      # the unit widths of the exact-f32 gather
    try:
        _narrow_wide_body(X, _hip, rf, cfg, F, h, w, g, got)
    finally:
        cfg.config['convmath'] = old_math
        X.check(X.lib.mvx_tuning_set(2, -1), 'mvx_tuning_set')


def _narrow_wide_body(X, _hip, rf, cfg, F, h, w, g, got):
    for cin, cout, flags in ((128, 128, 0), (64, 192, 0), (4 * 64, 128, rf.TAPS2)):
        x = torch.randn((F, h, w, cin), generator=g).to(DEV)
        wt = (torch.randn((cout, cin, 3, 3), generator=g) * 0.05).to(DEV)
        b = torch.randn((cout,), generator=g).to(DEV)
        dz = torch.randn((F, h, w, cout), generator=g).to(DEV)
        for units, limit in (('wide', 0), ('narrow', 1 << 60)):
            X.check(X.lib.mvx_tuning_set(2, limit), 'mvx_tuning_set')
            y, mi = rf._conv(x, _hip.conv3d_pack(wt, False), b, F, h, w, cin, cout, flags, cfg.eps)
            dx = rf._dgrad(dz, _hip.conv3d_pack(wt, True), F, h, w, cin, cout, flags)
            got[(cin, cout, flags, units)] = (y.clone(), mi.clone(), dx.clone())
        a, bq = got[(cin, cout, flags, 'wide')], got[(cin, cout, flags, 'narrow')]
        assert torch.equal(a[0], bq[0]) and torch.equal(a[2], bq[2]), (cin, cout, flags)
        assert rel(bq[1].double(), a[1].double()) < 1e-6



def test_stride2_and_stride1_layers_in_split_arithmetic_match_exact_f32():
    """The RPN's 3x3 layers in convmath fp16x3 / bf16x6 / bf16x3 against the exact-f32 kernels on the same inputs: forward (with the
    BatchNorm statistics), input gradient and weight gradient, stride 1 and stride 2 (MVX_FLAG_TAPS2: the 2x2 window on the
    space-to-depth image, with the structurally zero (window tap, parity) blocks skipped in all three arithmetics), on a
    ragged map and with both gather unit shapes of the split kernels."""
    from modules import Extension as X
    from modules import _hip
    from modules import rpn_frames as rf
    import modules.config as cfg
    F, h, w = 2, 21, 37
    g = torch.Generator().manual_seed(13)
    old = cfg.config.get('convmath', 'f32')
    try:
        for cin0, cout, s2 in ((128, 128, False), (64, 128, True), (128, 256, True)):
            wt = (torch.randn((cout, cin0, 3, 3), generator=g) * 0.05).to(DEV)
            w_eff = rf._s2d_weight(wt, 1) if s2 else wt
            cin = w_eff.shape[1]
            flags = rf.TAPS2 if s2 else 0
            x = torch.randn((F, h, w, cin), generator=g).to(DEV)
            b = torch.randn((cout,), generator=g).to(DEV)
            dz = torch.randn((F, h, w, cout), generator=g).to(DEV)
            res = {}
            for math, units in (('f32', 0), ('fp16x3', 0), ('fp16x3', 1 << 60), ('bf16x6', 0), ('bf16x6', 1 << 60), ('bf16x3', 0)):
                cfg.config['convmath'] = math
                sp = _hip.split_pieces()
                X.check(X.lib.mvx_tuning_set(1, units), 'mvx_tuning_set')          # MVX_TUNE_SPLIT16_MIN_UNITS: 16x16 / 8x16 units
                if sp:
                    w3 = torch.zeros(w_eff.shape[:2] + (3, 3, 3), device=DEV)
                    w3[:, :, 1] = w_eff
                    wf, wd = _hip.conv3d_pack(w3, False, split=sp), _hip.conv3d_pack(w3, True, split=sp)
                else:
                    wf, wd = _hip.conv3d_pack(w_eff, False), _hip.conv3d_pack(w_eff, True)
                y, mi = rf._conv(x, wf, b, F, h, w, cin, cout, flags, cfg.eps)
                dx = rf._dgrad(dz, wd, F, h, w, cin, cout, flags)
                dw = rf._wgrad(x, dz, F, h, w, cin, cout, flags)
                _hip.join_side_stream()
                torch.cuda.synchronize()
                res[(math, units)] = (y.clone(), mi.clone(), dx.clone(), dw.clone())
            ref = res[('f32', 0)]
            for key, tol in ((('fp16x3', 0), 3e-6), (('fp16x3', 1 << 60), 3e-6), (('bf16x6', 0), 3e-6), (('bf16x6', 1 << 60), 3e-6), (('bf16x3', 0), 3e-5)):
                r = res[key]
                for name, a, bb in zip(('y', 'mean_inv', 'dx', 'dw'), r, ref):
                    assert rel(a, bb) < tol * (10 if name == 'mean_inv' else 1), (cin0, cout, s2, key, name, rel(a, bb))
                if s2:      # the structural zeros of the rearranged kernel stay exact zeros in its gradient
                    assert float(r[3][w_eff == 0].abs().max()) == 0.0
    finally:
        cfg.config['convmath'] = old
        X.check(X.lib.mvx_tuning_set(1, 768), 'mvx_tuning_set')


def _to_planes(mid):
    """(F,128,H,W) BEV map (channel = c*2+d) -> channels-last planes [F*2][H][W][64] (what CML produces)."""
    F, _, H, W = mid.shape
    return mid.view(F, 64, 2, H, W).permute(0, 2, 3, 4, 1).reshape(F * 2, H, W, 64).contiguous()


def _from_planes(g, F):
    _, H, W, _ = g.shape
    return g.view(F, 2, H, W, 64).permute(0, 4, 1, 2, 3).reshape(F, 128, H, W)


def _load_rpn(P):
    from modules.voxelnet.Pipe import RPN
    rpn = RPN().to(DEV)
    rpn.load_state_dict({k[len('rpn.'):]: v for k, v in P.items()})
    return rpn


def test_rpn_forward_matches_reference_fixture(golden):
    from modules import rpn_frames as rf
    g = golden('voxelnet_small')
    P = O.rpn_params(golden('rpn_shapes'))
    rpn = _load_rpn(P)
    mid = torch.from_numpy(g['mid'])[None].to(DEV)
    heads, S = rf.rpn_forward(rpn, _to_planes(mid), 1, 2, mid.shape[2], mid.shape[3], 64)
    score, reg = rf.split_heads(heads, 1, S['h1'], S['w1'])
    # the fixture's deepest maps are 2 x 3 sites: BatchNorm over six values amplifies fp32 rounding (the reference's own
    # fp32 result is that far from exact arithmetic); the tight comparison is the float64 one below on larger maps
    assert float((score[0].cpu() - torch.from_numpy(g['score'])).abs().max()) < 1e-4
    assert rel(reg[0].cpu(), torch.from_numpy(g['reg'])) < 5e-4


def _nchw_input_of(rec):
    """The layer's input as the oracle sees it: NCHW float64 (stride-2 layers store the space-to-depth image)."""
    if rec['kind'] == 's1':
        return rec['x'].cpu().double().permute(0, 3, 1, 2).contiguous()
    xs = rec['x'].cpu().double()
    F, hh, ww, _ = xs.shape
    pl, Cf = rec['planes'], rec['cfull']
    full = xs.view(F, hh, ww, 2, 2, pl, Cf).permute(0, 5, 6, 1, 3, 2, 4).reshape(F, pl, Cf, hh * 2, ww * 2)
    return full.permute(0, 2, 1, 3, 4).reshape(F, Cf * pl, hh * 2, ww * 2).contiguous()          # channel = c*planes + d


def _nchw_grad_of(g, rec, F):
    """Input gradient of a layer in the same NCHW view."""
    g = g.cpu().double()
    if rec['kind'] == 's1' or rec['planes'] == 1:
        return g.reshape(F, g.shape[-3], g.shape[-2], g.shape[-1]).permute(0, 3, 1, 2)
    pl = rec['planes']
    _, H, W, Cf = g.shape
    return g.view(F, pl, H, W, Cf).permute(0, 4, 1, 2, 3).reshape(F, Cf * pl, H, W)


@pytest.mark.parametrize('F', [1, 3])
def test_rpn_forward_backward_match_oracle(golden, F):
    """Every layer of the HIP RPN, forward and backward, against the float64 oracle layer evaluated on the SAME input and
    the same upstream gradient (per frame: batch-1 BatchNorm): 1e-5.  The end-to-end maps: 1e-4.  (End-to-end gradients
    through 17 ReLU + BatchNorm layers on maps this small are decided by the handful of activations within 3e-5 of the
    ReLU kink -- the per-layer fp32 error of the MFMA accumulation chain, 1.5e-6, accumulated over the stack -- so they are
    only required to be close, 5e-2; tools/rpn_diag.py prints the breakdown.)"""
    from modules import _hip, parallel
    from modules import rpn_frames as rf
    P = O.rpn_params(golden('rpn_shapes'))
    rpn = _load_rpn(P)
    bucket = parallel.GradBucket(list(rpn.parameters()))
    H, W = 64, 96
    gen = torch.Generator().manual_seed(3)
    mids = torch.randn((F, 128, H, W), generator=gen)
    d_heads = torch.randn((F, H // 2, W // 2, 16), generator=gen) * 0.1
    bucket.zero()
    heads, S = rf.rpn_forward(rpn, _to_planes(mids.to(DEV)), F, 2, H, W, 64)
    S['capture'] = []
    g_in = rf.rpn_backward(rpn, S, d_heads.reshape(-1, 16).to(DEV))
    _hip.join_side_stream()
    torch.cuda.synchronize()
    P64 = {k: v.double() for k, v in P.items()}
    # ---- end to end, forward
    gx64 = []
    Pg = {k: v.clone().requires_grad_(True) for k, v in P64.items()}
    for f in range(F):
        x = mids[f:f + 1].double().requires_grad_(True)
        score, reg = O.rpn(x, Pg)
        out = torch.cat([torch.log(score / (1 - score)), reg], dim=1)[0].permute(1, 2, 0)
        got = heads.view(F, H // 2, W // 2, 16)[f].cpu().double()
        assert rel(got[..., 2:], out[..., 2:].detach()) < 1e-4
        assert float((torch.sigmoid(got[..., :2]) - score[0].permute(1, 2, 0).detach()).abs().max()) < 1e-4
        (out * d_heads[f].double()).sum().backward()
        gx64.append(x.grad[0])
    g_mid = _from_planes(g_in, F).cpu().double()
    assert max(rel(g_mid[f], gx64[f]) for f in range(F)) < 5e-2
    assert max(rel(p.grad.cpu().double(), Pg['rpn.' + k].grad) for k, p in rpn.named_parameters()) < 2e-1
    # ---- layer by layer on identical inputs
    names = ('blk1', 'blk2', 'blk3')
    cap = {k: (g, dx) for k, g, dx in S['capture']}
    worst = 0.0
    for bi, name in enumerate(names):
        layers = S['blocks'][bi]['layers']
        for li, rec in enumerate(layers):
            w_ = P64['rpn.%s.%d.conv.weight' % (name, li)].clone().requires_grad_(True)
            b_ = P64['rpn.%s.%d.conv.bias' % (name, li)].clone().requires_grad_(True)
            xin = _nchw_input_of(rec)
            nxt = layers[li + 1]['x'] if li + 1 < len(layers) else S['blocks'][bi]['out']
            ours_out = nxt.cpu().double().permute(0, 3, 1, 2)
            g_up, g_dx = cap[(bi, li)]
            g_up = g_up.cpu().double().permute(0, 3, 1, 2)
            dw_sum, db_sum = torch.zeros_like(w_), torch.zeros_like(b_)
            on_kink = False
            for f in range(F):                                   # batch-1 forwards: per-frame statistics
                xf = xin[f:f + 1].clone().requires_grad_(True)
                yh = O.crb2d(xf, w_, b_, 2 if li == 0 else 1, 1)
                assert rel(ours_out[f:f + 1], yh.detach()) < 1e-5, (name, li, 'forward')
                gw, gb, gxf = torch.autograd.grad((yh * g_up[f:f + 1]).sum(), (w_, b_, xf))
                dw_sum += gw
                db_sum += gb
                e = rel(_nchw_grad_of(g_dx, rec, F)[f:f + 1], gxf)
                # a pre-activation within 3e-7 of zero (|y| ~ 1 elsewhere: inside ANY fp32 evaluation's rounding) decides its ReLU
                # either way; the flipped site then moves this layer's gradients by ~1e-2 through the BatchNorm statistics (seen
                # with F = 3 on blk1.1: min |y| = 7.2e-8, 1145 of 196608 elements off; located with a one-off script, since removed).  Such a frame is
                # only required to be close; test_rpn_and_loss_gradients_tight_at_full_size shares the masks instead
                pre = torch.nn.functional.conv2d(xf.detach(), w_.detach(), b_.detach(), stride=2 if li == 0 else 1, padding=1)
                kink = float(pre.abs().min()) < 3e-7
                on_kink = on_kink or kink
                if not kink:
                    worst = max(worst, e)
                assert e < (5e-2 if kink else 1e-5), (name, li, 'input gradient', e)
            m = getattr(rpn, name)[li]
            e_w, e_b = rel(m.conv.weight.grad.cpu().double(), dw_sum), rel(m.conv.bias.grad.cpu().double(), db_sum)
            if not on_kink:
                worst = max(worst, e_w, e_b)
            assert e_w < (5e-2 if on_kink else 1e-5) and e_b < (5e-2 if on_kink else 1e-5), (name, li, e_w, e_b)
    # deconvolutions and heads: same-input check with the exact upstream gradient (the heads are linear)
    W_heads = torch.cat([P64['rpn.cls.weight'].view(2, 768), P64['rpn.reg.weight'].view(14, 768)])
    h1, w1 = H // 2, W // 2
    g_up_all = (d_heads.reshape(-1, 16).double() @ W_heads).view(F, h1, w1, 768).permute(0, 3, 1, 2)
    for name, sl, s_, pad, xin in (('deconv1', slice(0, 256), 1, 1, S['blocks'][0]['out']), ('deconv2', slice(256, 512), 2, 0, S['blocks'][1]['out']),
                                   ('deconv3', slice(512, 768), 4, 0, S['blocks'][2]['out'])):
        w_ = P64['rpn.%s.deconv.weight' % name].clone().requires_grad_(True)
        b_ = P64['rpn.%s.deconv.bias' % name].clone().requires_grad_(True)
        xo = xin.cpu().double().permute(0, 3, 1, 2).contiguous()
        dw_sum, db_sum = torch.zeros_like(w_), torch.zeros_like(b_)
        for f in range(F):
            yh = O.decrb2d(xo[f:f + 1], w_, b_, s_, pad)
            up_ours = S['up'].view(F, h1, w1, 768)[f, :, :, sl].cpu().double().permute(2, 0, 1)[None]
            assert rel(up_ours, yh.detach()) < 1e-5, (name, 'forward')
            gw, gb = torch.autograd.grad((yh * g_up_all[f:f + 1, sl]).sum(), (w_, b_))
            dw_sum += gw
            db_sum += gb
        m = getattr(rpn, name)
        e_w, e_b = rel(m.deconv.weight.grad.cpu().double(), dw_sum), rel(m.deconv.bias.grad.cpu().double(), db_sum)
        worst = max(worst, e_w, e_b)
        assert e_w < 1e-5 and e_b < 1e-5, (name, e_w, e_b)
    print('worst same-input layer error (float64 yardstick): %.2e' % worst)


def test_rpn_hip_agrees_with_the_module_path_at_full_size():
    """Full-size maps (400x352): the HIP frame-set RPN against the nn.Module RPN (stock PyTorch-ROCm) on the same weights."""
    from modules import rpn_frames as rf
    from modules.voxelnet.Pipe import RPN
    torch.manual_seed(0)
    rpn = RPN().to(DEV)
    mid = torch.randn((1, 128, 352, 400), device=DEV)
    with torch.no_grad():
        s_ref, r_ref = rpn.forward_torch(mid)
        heads, S = rf.rpn_forward(rpn, _to_planes(mid), 1, 2, 352, 400, 64)
        score, reg = rf.split_heads(heads, 1, S['h1'], S['w1'])
    assert float((score - s_ref).abs().max()) < 1e-4
    assert rel(reg, r_ref) < 1e-4


def test_rpn_bf16x3_mode_stays_within_the_feature_bar(golden):
    """convmath: bf16x3 -- the RPN's 3x3 convolutions (forward, dgrad, wgrad; stride-2 layers through their rearranged 3x3
    kernel) on the split-MFMA kernels.  Through 17 layers the split's 2e-5 per product reaches 1.3e-4 on the head maps of
    these small 64x96 inputs (deepest maps 8x12 sites per frame): above the 1e-4 feature bar, which is why this mode is
    opt-in; bounded here at 3e-4, gradients loosely (ReLU-kink sensitivity of small maps)."""
    import modules.config as cfg
    from modules import _hip, parallel
    from modules import rpn_frames as rf
    P = O.rpn_params(golden('rpn_shapes'))
    rpn = _load_rpn(P)
    bucket = parallel.GradBucket(list(rpn.parameters()))
    F, H, W = 2, 64, 96
    gen = torch.Generator().manual_seed(5)
    x_cl = _to_planes(torch.randn((F, 128, H, W), generator=gen).to(DEV))
    d_heads = (torch.randn((F * (H // 2) * (W // 2), 16), generator=gen) * 0.1).to(DEV)
    res = {}
    old = cfg.config.get('convmath', 'f32')
    try:
        for math in ('f32', 'bf16x3', 'bf16x6', 'fp16x3'):
            cfg.config['convmath'] = math
            bucket.zero()
            heads, S = rf.rpn_forward(rpn, x_cl, F, 2, H, W, 64)
            g = rf.rpn_backward(rpn, S, d_heads)
            _hip.join_side_stream()
            torch.cuda.synchronize()
            res[math] = (heads.clone(), g.clone(), bucket.flat.clone())
    finally:
        cfg.config['convmath'] = old
    assert rel(res['bf16x3'][0], res['f32'][0]) < 3e-4
    assert rel(res['bf16x6'][0], res['f32'][0]) < 5e-5          # bf16x6 (three pieces, fp32-grade): both are ~2e-5 from float64
    assert rel(res['fp16x3'][0], res['f32'][0]) < 5e-5          # fp16x3 (two fp16 pieces, 22 bits): likewise
    for math in ('bf16x3', 'bf16x6', 'fp16x3'):
        assert rel(res[math][1], res['f32'][1]) < 1e-1          # input gradient: the ReLU-kink sensitivity of these small maps
        a, b = res[math][2], res['f32'][2]
        off = 0
        for k, p in rpn.named_parameters():
            ga, gb = a[off:off + p.numel()], b[off:off + p.numel()]
            off += p.numel()
            # 2-norm, not max-norm: a handful of ReLU flips on the 8x12-site maps moves single weight-gradient entries by tens
            # of per cent in either arithmetic (tools/rpn_diag.py); the gradient as a whole must agree
            assert float((ga - gb).norm() / gb.norm().clamp_min(1e-30)) < 1e-1, (math, k)


def test_rpn_full_size_maps_match_the_float64_oracle(golden):
    """The RPN at the benchmark's size (352x400 BEV map -> 176x200 heads, one frame inside a 2-frame set) against the oracle
    evaluated in float64 on the CPU: score within 1e-4 absolute, regression map within 1e-4 of its maximum (north_star's
    bar for the RPN maps), where BatchNorm is well conditioned (>= 2,200 sites per channel in the deepest maps)."""
    from modules import rpn_frames as rf
    P = O.rpn_params(golden('rpn_shapes'))
    rpn = _load_rpn(P)
    F, H, W = 2, 352, 400
    gen = torch.Generator().manual_seed(11)
    mids = torch.randn((F, 128, H, W), generator=gen)
    import modules.config as cfg
    old = cfg.config.get('convmath', 'f32')
    cfg.config['convmath'] = 'f32'
    try:
        heads, S = rf.rpn_forward(rpn, _to_planes(mids.to(DEV)), F, 2, H, W, 64)
    finally:
        cfg.config['convmath'] = old
    got = heads.view(F, H // 2, W // 2, 16)[1].cpu().double()
    torch.set_num_threads(max(1, torch.get_num_threads()))
    with torch.no_grad():
        score, reg = O.rpn(mids[1:2].double(), {k: v.double() for k, v in P.items()})
    e_score = float((torch.sigmoid(got[..., :2]) - score[0].permute(1, 2, 0)).abs().max())
    e_reg = rel(got[..., 2:], reg[0].permute(1, 2, 0))
    print('full-size RPN vs float64 oracle: score %.2e (abs), reg %.2e (max-norm rel)' % (e_score, e_reg))
    assert e_score < 1e-4 and e_reg < 1e-4
    # the same maps in convmath: bf16x3 (opt-in): reported, and bounded at 5e-4
    import modules.config as cfg
    old = cfg.config.get('convmath', 'f32')
    cfg.config['convmath'] = 'bf16x3'
    try:
        heads2, _ = rf.rpn_forward(rpn, _to_planes(mids.to(DEV)), F, 2, H, W, 64)
    finally:
        cfg.config['convmath'] = old
    got2 = heads2.view(F, H // 2, W // 2, 16)[1].cpu().double()
    e2_score = float((torch.sigmoid(got2[..., :2]) - score[0].permute(1, 2, 0)).abs().max())
    e2_reg = rel(got2[..., 2:], reg[0].permute(1, 2, 0))
    print('  bf16x3: score %.2e (abs), reg %.2e (max-norm rel)' % (e2_score, e2_reg))
    assert e2_score < 5e-4 and e2_reg < 5e-4
    # ... and in the 22+-bit split arithmetics -- bf16x6 (three bf16 pieces per operand) and fp16x3 (two fp16 pieces, the
    # default) --: north_star's 1e-4 bar like the exact-f32 mode
    for math in ('bf16x6', 'fp16x3'):
        cfg.config['convmath'] = math
        try:
            heads3, _ = rf.rpn_forward(rpn, _to_planes(mids.to(DEV)), F, 2, H, W, 64)
        finally:
            cfg.config['convmath'] = old
        got3 = heads3.view(F, H // 2, W // 2, 16)[1].cpu().double()
        e3_score = float((torch.sigmoid(got3[..., :2]) - score[0].permute(1, 2, 0)).abs().max())
        e3_reg = rel(got3[..., 2:], reg[0].permute(1, 2, 0))
        print('  %s: score %.2e (abs), reg %.2e (max-norm rel)' % (math, e3_score, e3_reg))
        assert e3_score < 1e-4 and e3_reg < 1e-4, math


def test_rpn_module_runs_on_the_hip_node_and_matches_float64():
    """modules.voxelnet.Pipe.RPN.forward -- the reference's own interface (voxelnet/Pipe.py:67-75), (1,128,H,W) in, (score,
    reg) out, under autograd -- IS the HIP path (RPNFunction over modules/rpn_frames.py, one frame, one plane of 128
    channels), not MIOpen: outputs, input gradient and every parameter gradient against a float64 CPU evaluation of the
    same module, next to the torch / MIOpen path (forward_torch) as the fp32 comparator."""
    import copy
    from modules import _hip
    from modules.voxelnet.Pipe import RPN
    torch.manual_seed(12)
    rpn = RPN().to(DEV)
    x0 = torch.randn((1, 128, 176, 200), device=DEV)        # deepest maps 22 x 25 sites: BatchNorm reasonably conditioned
    res = {}
    launches = {}
    for mode in ('hip', 'torch'):
        rpn.zero_grad()
        x = x0.clone().requires_grad_(True)
        l0 = _hip.X.lib.mvx_launch_count()
        s, r = rpn(x) if mode == 'hip' else rpn.forward_torch(x)
        (s.square().sum() + r.square().sum()).backward()
        torch.cuda.synchronize()
        launches[mode] = _hip.X.lib.mvx_launch_count() - l0
        res[mode] = (s.detach().cpu(), r.detach().cpu(), x.grad.cpu(), {k: p.grad.cpu() for k, p in rpn.named_parameters()})
    assert launches['hip'] > 100 and launches['torch'] == 0, launches      # the module's forward runs this library's kernels
    ref = copy.deepcopy(rpn).cpu().double()
    ref.zero_grad()
    x = x0.cpu().double().requires_grad_(True)
    s, r = ref(x)
    (s.square().sum() + r.square().sum()).backward()
    f64 = (s.detach(), r.detach(), x.grad, {k: p.grad for k, p in ref.named_parameters()})
    assert float((res['hip'][0] - f64[0]).abs().max()) < 1e-4 and rel(res['hip'][1], f64[1]) < 1e-4
    assert rel(res['hip'][2], f64[2]) < 3 * rel(res['torch'][2], f64[2]) + 1e-3

    def rel2(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    table = {k: (rel2(res['hip'][3][k], f64[3][k]), rel2(res['torch'][3][k], f64[3][k])) for k in f64[3]}
    worst = max(table.items(), key=lambda t: t[1][0])
    print('RPN module on the HIP node vs float64 (2-norm): worst parameter %s hip %.2e, torch / MIOpen path %.2e; median hip %.2e, '
          'median torch %.2e' % (worst[0], worst[1][0], worst[1][1], float(np.median([v[0] for v in table.values()])),
                                 float(np.median([v[1] for v in table.values()]))))
    for k, (e_hip, e_ref) in table.items():
        # two fp32 evaluations of 17 ReLU + BatchNorm layers: either may be the closer one on a given parameter
        assert e_hip < 3 * e_ref + 1e-2, (k, e_hip, e_ref)
    # gradients accumulate like any autograd node: two forward / backward passes without zero_grad give twice the gradient
    rpn.zero_grad()
    for _ in range(2):
        x = x0.clone().requires_grad_(True)
        s, r = rpn(x)
        (s.square().sum() + r.square().sum()).backward()
    torch.cuda.synchronize()
    k0 = 'blk1.0.conv.weight'
    assert rel(dict(rpn.named_parameters())[k0].grad.cpu(), 2 * res['hip'][3][k0]) < 1e-5


# The RPN + loss gradients against float64 WITH THE SAME ReLU MASKS (see the test): measured 2-norm distances on MI355X are in
# profiles/r04_rpn_loss_grads.json; measured <= 8e-6 in exact f32 and <= 6.7e-6 in bf16x6; asserted at 1e-4 for every tensor, which every 1 % mutation of a backward term
# breaks (2.2e-3 .. 6.2e-2)
_RPN_GRAD_BOUND = 1e-4
_RPN_MUTATIONS = ('heads_dgrad', 'deconv2_dgrad', 'deconv4_dgrad', 'deconv1_dgrad', 'bn_bwd_b0', 'bn_bwd_b1', 'bn_bwd_b2',
                  's1_dgrad_b0', 's1_dgrad_b1', 's1_dgrad_b2', 's2_dgrad_b1', 's2_dgrad_b2')


def test_rpn_and_loss_gradients_tight_at_full_size():
    """The RPN + VoxelLoss backward chain (voxelnet/Pipe.py:45-75, voxelnet/Loss.py:15-45) at the benchmark's size -- the part
    of `--mode full` that the hot-path gradient check does not cover -- against float64: the REAL loss (classification +
    regression on anchor targets of eight boxes) on a 352x400 BEV map, every rpn.* parameter gradient and the gradient handed
    to the CML.

    What makes this comparison TIGHT: the float64 evaluation uses the ReLU masks of the HIP forward (y > 0 of every layer,
    taken from the saved activations).  Seventeen Conv-ReLU-BatchNorm layers are chaotic under ReLU flips: an activation within
    1e-6 of zero lands on different sides in fp32 and float64, and ONE such flip in a deep layer moves every gradient below it
    by 1e-3 .. 1e-2 (torch's own CPU fp32 against float64 on this network: 4e-4 .. 1e-2 with a loss-shaped and with a smooth
    upstream gradient alike, uniform below the flipped layer; tools/dbg_rpn_grads.py).  With the masks shared, what is left
    is the backward ARITHMETIC of every layer kind -- heads, kernel = stride deconvolutions (pixel shuffle backwards), the
    stride-1 transposed convolution, stride-1 and stride-2 (space-to-depth) 3x3 layers, BatchNorm backward -- and that is
    held to 1e-4 (2-norm) per tensor (measured 3e-6 median, 8e-6 worst).  The forward values differ from the plain float64 evaluation only at the flipped
    elements, by < 1e-5 (losses asserted at 1e-4).  **Mutation check**: each backward term scaled by 1.01 through
    rpn_frames._MUTATE must break the bound."""
    import json
    import torch.nn.functional as Fn
    import modules.config as cfg
    from modules import _hip, parallel, Calc
    from modules import rpn_frames as rf
    from modules.data import Preprocessing as pre
    from modules.pipeline import heads_loss
    from modules.voxelnet import VoxelLoss
    from modules.voxelnet.Pipe import RPN
    F_, H, W = 1, 352, 400
    gen = torch.Generator().manual_seed(31)
    rpn = RPN().to(DEV)
    P = {}
    for k, p in rpn.state_dict().items():
        if k.endswith('weight'):
            fan = p.shape[1] * p.shape[2] * p.shape[3] if 'deconv' not in k else p.shape[0]      # ConvTranspose2d: (Cin, Cout, k, k)
            v = torch.randn(p.shape, generator=gen) / np.sqrt(fan)
            if k.startswith(('cls', 'reg')):
                v = v * 0.3
        else:       # biases of the BatchNorm-ed layers at +0.5: no channel is almost dead (BatchNorm without affine, eps 1e-6)
            v = torch.zeros(p.shape) if k.startswith(('cls', 'reg')) else torch.full(p.shape, 0.5)
        P['rpn.' + k] = v
    rpn.load_state_dict({k[4:]: v for k, v in P.items()})
    mid = torch.randn((F_, 128, H, W), generator=gen)
    anchors = pre.createAnchors(H // 2, W // 2, cfg.velorange, cfg.carsize)
    bevs = Calc.bbox3d2bev(anchors.reshape(anchors.shape[:2] + (-1, 7)))
    gg = np.random.default_rng(11)
    n = 8
    gt = torch.tensor(np.stack([gg.uniform(8, 60, n), gg.uniform(-30, 30, n), gg.uniform(-1.8, -0.6, n), gg.uniform(3.4, 4.4, n),
                                gg.uniform(1.5, 1.8, n), gg.uniform(1.4, 1.7, n),
                                gg.choice([0.0, np.pi / 2], n) + gg.normal(0, 0.05, n)], 1), dtype=torch.float32)
    pi, ni, gi = Calc.classifyAnchors(Calc.bbox3d2bev(gt), gt[:, [0, 1]], bevs.to(DEV).contiguous(), cfg.velorange, 0.45, 0.6)
    params = list(rpn.named_parameters())
    bucket = parallel.GradBucket([p for _, p in params])
    crit = VoxelLoss()

    def backward(S, d_heads):
        bucket.zero()
        old_async, _hip.ASYNC_WGRAD = _hip.ASYNC_WGRAD, True
        try:
            g = rf.rpn_backward(rpn, S, d_heads)
            _hip.join_side_stream()
        finally:
            _hip.ASYNC_WGRAD = old_async
        torch.cuda.synchronize()
        out = {k: p.grad.detach().cpu().double().clone() for k, p in params}
        out['d_mid'] = _from_planes(g, F_).cpu().double()
        return out

    with torch.no_grad():
        heads, S = rf.rpn_forward(rpn, _to_planes(mid.to(DEV)), F_, 2, H, W, 64)
        losses, has_reg, d_heads = heads_loss(heads, F_, S['h1'], S['w1'], [(pi, ni, gi, gt.to(DEV))], crit, anchors.to(DEV))
        got = backward(S, d_heads)
    assert has_reg[0]
    # ---- the ReLU masks of the HIP forward, in the oracle's NCHW layout
    masks = {}
    for bi, blk in enumerate(S['blocks']):
        for li, rec in enumerate(blk['layers']):
            masks[('blk', bi, li)] = (rec['y'] > 0).permute(0, 3, 1, 2).cpu().double()
    masks['d1'] = (S['d1']['y'] > 0).permute(0, 3, 1, 2).cpu().double()
    for rec in S['dk']:
        s_, hk, wk, co = rec['s'], rec['h'], rec['w'], rec['cout']
        t = (rec['t'] > 0).view(F_, hk, wk, s_, s_, co).permute(0, 5, 1, 3, 2, 4).reshape(F_, co, hk * s_, wk * s_)
        masks[('dk', s_)] = t.cpu().double()
    # ---- float64 with autograd, ReLU replaced by the multiplication with those masks
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    m64 = mid.double().requires_grad_(True)

    def bn(y):
        return Fn.batch_norm(y, None, None, None, None, True, 0.0, O.EPS)
    x = m64
    outs = []
    for bi, (name, nl) in enumerate((('blk1', 4), ('blk2', 6), ('blk3', 6))):
        for li in range(nl):
            x = bn(Fn.conv2d(x, P64['rpn.%s.%d.conv.weight' % (name, li)], P64['rpn.%s.%d.conv.bias' % (name, li)],
                             2 if li == 0 else 1, 1) * masks[('blk', bi, li)])
        outs.append(x)
    ups = [bn(Fn.conv_transpose2d(outs[0], P64['rpn.deconv1.deconv.weight'], P64['rpn.deconv1.deconv.bias'], 1, 1) * masks['d1']),
           bn(Fn.conv_transpose2d(outs[1], P64['rpn.deconv2.deconv.weight'], P64['rpn.deconv2.deconv.bias'], 2, 0) * masks[('dk', 2)]),
           bn(Fn.conv_transpose2d(outs[2], P64['rpn.deconv3.deconv.weight'], P64['rpn.deconv3.deconv.bias'], 4, 0) * masks[('dk', 4)])]
    up = torch.cat(ups, dim=1)
    score = torch.sigmoid(Fn.conv2d(up, P64['rpn.cls.weight'], P64['rpn.cls.bias']))
    reg = Fn.conv2d(up, P64['rpn.reg.weight'], P64['rpn.reg.bias'])
    cls, rl = O.voxel_loss([t.cpu().numpy() for t in pi], [t.cpu().numpy() for t in ni], gi.cpu().numpy(), gt.double(),
                           score[0].permute(1, 2, 0), reg[0].permute(1, 2, 0), O.create_anchors(H // 2, W // 2).double(), 2)
    (cls + rl).backward()
    ref = {k[4:]: v.grad for k, v in P64.items()}
    ref['d_mid'] = m64.grad
    e_loss = (abs(float(losses[0, 0]) - float(cls)) / abs(float(cls)), abs(float(losses[0, 1]) - float(rl)) / abs(float(rl)))

    def dist(g):
        return {k: float((g[k] - ref[k]).norm() / ref[k].norm()) for k in ref}
    d0 = dist(got)
    report = {'convmath': cfg.config.get('convmath', 'f32'), 'loss_rel': e_loss, 'rel_2norm_vs_float64_same_masks': d0, 'mutations': {}}
    # ---- every backward term mutated by 1 %: the worst tensor must leave the bound
    try:
        for name in _RPN_MUTATIONS:
            rf._MUTATE = {name: 1.01}
            with torch.no_grad():
                dm = dist(backward(S, d_heads))
            report['mutations'][name] = max(dm.values())
    finally:
        rf._MUTATE = {}
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(REPO, 'gpurun_out', 'rpn_loss_grads.json'), 'w') as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report))
    assert e_loss[0] < 1e-4 and e_loss[1] < 1e-4
    assert max(d0.values()) < _RPN_GRAD_BOUND, sorted(d0.items(), key=lambda t: -t[1])[:5]
    assert min(report['mutations'].values()) > _RPN_GRAD_BOUND, report['mutations']


@pytest.mark.parametrize('kind,args,hw', [('conv', (128, 128, 3, 1, 1), (40, 48)), ('conv', (64, 128, 3, 2, 1), (40, 48)),
                                          ('conv', (128, 256, 3, 2, 1), (24, 32)), ('deconv', (128, 256, 3, 1, 1), (24, 32)),
                                          ('deconv', (128, 256, 2, 2, 0), (20, 24)), ('deconv', (256, 256, 4, 4, 0), (10, 12))])
def test_standalone_blocks_run_on_hip_nodes_and_match_float64(kind, args, hw):
    """CRB2d / DeCRB2d called ON THEIR OWN (reference modules/layers/Blocks.py:31-51 -- not through RPN.forward) are one
    autograd node each on this library's kernels by default (modules/layers/Block2d.py: 3x3 stride 1, 3x3 stride 2 through the
    space-to-depth form, transposed 3x3 stride 1, kernel = stride deconvolution): library launches are counted, output, input
    gradient and parameter gradients are compared with a float64 CPU evaluation of the same module."""
    import copy
    from modules import Extension as X
    from modules.layers import CRB2d, DeCRB2d
    torch.manual_seed(3)
    m = (CRB2d if kind == 'conv' else DeCRB2d)(*args).to(DEV)
    for p in m.parameters():                          # alive channels: BatchNorm without affine on a small map
        if p.dim() == 1:
            p.data.fill_(0.3)
    x0 = torch.randn((1, args[0]) + hw)
    G = torch.randn(m(x0.to(DEV)).shape)
    n0 = X.lib.mvx_launch_count()
    x = x0.to(DEV).requires_grad_(True)
    y = m(x)
    (y * G.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert X.lib.mvx_launch_count() - n0 >= 6, 'the block did not run on the HIP kernels'
    ref = copy.deepcopy(m).cpu().double()
    ref.zero_grad()
    xr = x0.double().requires_grad_(True)
    from modules.layers import Blocks
    Blocks._IN_FORWARD_TORCH[0] = True                # the torch comparator, chosen explicitly (a bare CPU call raises)
    try:
        yr = ref(xr)
    finally:
        Blocks._IN_FORWARD_TORCH[0] = False
    (yr * G.double()).sum().backward()
    assert rel(y.detach().cpu().double(), yr.detach()) < 2e-5
    n2 = lambda a, b: float((a.cpu().double() - b).norm() / b.norm())
    assert n2(x.grad, xr.grad) < 1e-3
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert n2(p.grad, q.grad) < 1e-3, k


def test_uncovered_standalone_block_warns_once_before_running_on_torch():
    """a CRB2d the MFMA tiles do not cover (here a batch of two) runs on the torch modules -- with a
    RuntimeWarning naming the reason, once."""
    import warnings
    from modules.layers import CRB2d
    from modules.layers import Blocks
    m = CRB2d(64, 64, 3, 1, 1).to(DEV)
    x = torch.randn(2, 64, 16, 16, device=DEV)
    Blocks._WARNED.clear()
    with pytest.warns(RuntimeWarning, match='batch size 2'):
        y = m(x)
    assert y.shape == (2, 64, 16, 16)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        m(x)                                                  # second call: no second warning
