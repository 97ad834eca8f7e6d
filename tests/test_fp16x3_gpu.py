"""fp16x3 arithmetic (MVX_FLAG_SPLIT_F16, csrc/split_common.h): two fp16 pieces per f32 operand, three MFMAs per product.  The
pieces carry 22 mantissa bits only inside fp16's exponent range, so gradient operands are scaled by a power of two taken from a
device-side max |value| that the producing kernel writes.  Checked here: the producers' amax is exact, scaled operands are
fp32-grade at any magnitude and with outlier rows, the scaling is exactly neutral (powers of two), the binding is consumed by
one launch, and forward operands are never scaled by data."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))


def to_cl(x):
    return x.permute(1, 2, 3, 0).contiguous()


def test_producers_write_the_exact_maximum():
    """mvx_tensor_amax (tail of 1-3 elements included) and the two BatchNorm-backward entries: the float they leave behind is
    bit-equal to max |dz| of what they wrote."""
    from modules import _hip
    g = torch.Generator().manual_seed(0)
    for n in (4096, 4099, 3, 1 << 20):
        t = (torch.randn((n + 4,), generator=g) * 3e-6).to(DEV)[:n]        # a view that starts 16-byte aligned
        _hip.tensor_amax(t)
        assert float(_hip.amax_of(t)) == float(t.abs().max())
    for rows, C, scale in ((5000, 128, 1e-5), (777, 16, 40.0), (12345, 64, 1.0)):
        y = torch.randn((rows, C), generator=g).to(DEV)
        gup = (torch.randn((rows, C), generator=g) * scale).to(DEV)
        st = torch.stack([y.relu().double().sum(0), (y.relu().double() ** 2).sum(0)])[None].repeat(_hip.STATS_REPLICAS, 1, 1)
        st[1:] = 0
        mi = _hip.bn_finalize(st.contiguous(), rows, 1e-6)
        dz, _ = _hip.bn_relu_backward(gup, y, mi, rows, True)
        assert _hip.amax_of(dz) is not None and float(_hip.amax_of(dz)) == float(dz.abs().max())


@pytest.mark.parametrize('scale', [1.0, 3e-7, 2e4])
@pytest.mark.parametrize('R,K,N', [(3000, 768, 128), (5000, 128, 768), (2100, 128, 16)])
def test_row_gemm_gradients_at_any_magnitude(R, K, N, scale):
    """dz of magnitude `scale` (gradients of a mean loss sit near 1e-6): with its range tag the input gradient dz W and the
    weight gradient dz^T x are fp32-grade against float64; without the tag the same fp16 kernel is not (the reason for the
    tags), and the Python wrappers then route the call to bf16x6 instead."""
    from modules import _hip
    g = torch.Generator().manual_seed(R + N)
    x = torch.randn((R, K), generator=g).to(DEV)
    dz = (torch.randn((R, N), generator=g) * scale).to(DEV)
    wT = (torch.randn((K, N), generator=g) / np.sqrt(N)).to(DEV)          # row-major [K][N]: dx = dz wT^T
    ref_w = dz.double().t() @ x.double()
    ref_x = dz.double() @ wT.double().t()
    _hip.tensor_amax(dz)
    dw = _hip.linear_wgrad(x, dz, split=4)
    dx, _ = _hip.linear_forward(dz, wT, None, relu=False, want_stats=False, label='linear_dgrad', split=4)
    assert rel_err(dw, ref_w) < 2e-6 and rel_err(dx, ref_x) < 2e-6
    if scale < 1e-3 and (R, K, N) == (3000, 768, 128):     # (the 128 x 768 case is a skinny problem: split-K exact-f32 kernel)
        raw = dz.clone()                                                  # no tag: the bare kernel (label None = a forward call)
        dxr, _ = _hip.linear_forward(raw, wT, None, relu=False, want_stats=False, split=4)
        assert rel_err(dxr, ref_x) > 1e-4
        dxs, _ = _hip.linear_forward(raw, wT, None, relu=False, want_stats=False, label='linear_dgrad', split=4)
        assert rel_err(dxs, ref_x) < 2e-6                                 # routed to bf16x6


def test_outlier_rows_keep_the_typical_rows_accurate():
    """The fusion MLP's shared padded row stands for ~6e5 dense rows and its dz is that much larger than a real row's: amax is
    scaled to the TOP of fp16's range ([2^14, 2^15)), so rows 1e6 times smaller than the outlier still keep ~20 bits."""
    from modules import _hip
    g = torch.Generator().manual_seed(1)
    R, K, N = 20000, 128, 128
    x = torch.randn((R, K), generator=g).to(DEV)
    dz = (torch.randn((R, N), generator=g) * 1e-5).to(DEV)
    dz[-1] *= 1e6
    wT = (torch.randn((K, N), generator=g) * 0.05).to(DEV)
    ref = dz.double() @ wT.double().t()
    _hip.tensor_amax(dz)
    dx, _ = _hip.linear_forward(dz, wT, None, relu=False, want_stats=False, label='linear_dgrad', split=4)
    assert rel_err(dx, ref) < 2e-6
    assert rel_err(dx[:-1], ref[:-1]) < 1e-5
    dw = _hip.linear_wgrad(x, dz, split=4)
    assert rel_err(dw, dz.double().t() @ x.double()) < 2e-6


@pytest.mark.parametrize('scale', [1.0, 1e-6])
def test_convolution_gradients_at_any_magnitude(scale):
    """The gather (input gradient) and the 4-wave weight-gradient kernel with a tagged dz of magnitude `scale` against float64,
    and the identical call on dz * 2^-24 tagged with ITS range: bit-equal after scaling back (the scale factors are powers of
    two: the rounding of every product and sum is unchanged)."""
    from modules import _hip
    g = torch.Generator().manual_seed(7)
    cin = cout = 64
    din, H, W, sd, pd = 3, 40, 48, 1, 1
    x = torch.randn((cin, din, H, W), generator=g)
    w = torch.randn((cout, cin, 3, 3, 3), generator=g) / np.sqrt(27 * cin)
    dout = _hip.conv_out_depth(din, sd, pd)
    dz = torch.randn((cout, dout, H, W), generator=g) * scale
    xg, wg = x[None].double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv3d(xg, wg, None, (sd, 1, 1), (pd, 1, 1)).backward(dz[None].double())
    xc, dzc, wd = to_cl(x).to(DEV), to_cl(dz).to(DEV), w.to(DEV)
    _hip.tensor_amax(dzc)
    wpd = _hip.conv3d_pack(wd, True, split=4)
    dx = _hip.conv3d_dgrad(dzc, wpd, din, cin, sd, pd, split=4)
    dw = _hip.conv3d_wgrad(xc, dzc, sd, pd, split=4)
    assert rel_err(dx.cpu().permute(3, 0, 1, 2), xg.grad[0]) < 4e-6
    assert rel_err(dw.cpu(), wg.grad) < 4e-6
    small = (dzc * 2.0 ** -24).contiguous()
    _hip.tensor_amax(small)
    dx2 = _hip.conv3d_dgrad(small, wpd, din, cin, sd, pd, split=4)
    dw2 = _hip.conv3d_wgrad(xc, small, sd, pd, split=4)
    assert torch.equal(dx2 * 2.0 ** 24, dx) and torch.equal(dw2 * 2.0 ** 24, dw)
    if scale < 1e-3:
        bare = dzc.clone()                                                # the same kernel without the range
        assert rel_err(_hip.conv3d_dgrad(bare, wpd, din, cin, sd, pd, split=4).cpu().permute(3, 0, 1, 2), xg.grad[0]) > 1e-4


def test_binding_is_consumed_by_one_launch():
    """mvx_split_operand_amax applies to the calling thread's NEXT split launch only -- also when that call ends up on a kernel
    that does not use it.  A wildly wrong range (1e-30: scale 2^125, every operand overflows fp16) ruins exactly one call."""
    from modules import _hip
    from modules import Extension as X
    g = torch.Generator().manual_seed(2)
    x = torch.randn((1000, 128), generator=g).to(DEV)
    w = (torch.randn((256, 128), generator=g) * 0.1).to(DEV)
    ref = x.double() @ w.double().t()
    bad = torch.full((1,), 1e-30, device=DEV)

    def run(split):
        return _hip.linear_forward(x, w, None, relu=False, want_stats=False, split=split)[0]

    X.check(X.lib.mvx_split_operand_amax(X.ptr(bad), None), 'mvx_split_operand_amax')
    y1 = X.lib.mvx_linear_forward      # the raw entry: the wrapper would re-bind
    out = torch.empty((1000, 256), device=DEV)
    X.check(y1(X.ptr(x), 128, X.ptr(w), 128, 0, None, X.ptr(out), 256, None, None, 1000, 128, 256,
               _hip.split_flags(4, True), None, 0, X.stream()), 'mvx_linear_forward')
    assert not bool(torch.isfinite(out).all())                            # the bound range was used ...
    X.check(y1(X.ptr(x), 128, X.ptr(w), 128, 0, None, X.ptr(out), 256, None, None, 1000, 128, 256,
               _hip.split_flags(4, True), None, 0, X.stream()), 'mvx_linear_forward')
    assert rel_err(out, ref) < 2e-6                                       # ... once
    # consumed by a call that runs the exact-f32 kernel (n <= 64) as well
    X.check(X.lib.mvx_split_operand_amax(X.ptr(bad), None), 'mvx_split_operand_amax')
    w16 = w[:16].contiguous()
    o16 = torch.empty((1000, 16), device=DEV)
    X.check(y1(X.ptr(x), 128, X.ptr(w16), 128, 0, None, X.ptr(o16), 16, None, None, 1000, 128, 16,
               _hip.split_flags(4, True), None, 0, X.stream()), 'mvx_linear_forward')
    assert rel_err(o16, x.double() @ w16.double().t()) < 2e-6
    assert rel_err(run(4), ref) < 2e-6


def test_a_rejected_call_drops_its_binding():
    """a gather entry point used to consume the binding only after its argument checks -- a dgrad call rejected by
    validation left the gradient's range bound, and the next forward launch scaled its activations by it.  Every failed
    MVX_CHECK_ARG / geometry check now drops the binding."""
    from modules import _hip
    from modules import Extension as X
    g = torch.Generator().manual_seed(3)
    x = torch.randn((1000, 128), generator=g).to(DEV)
    w = (torch.randn((256, 128), generator=g) * 0.1).to(DEV)
    bad = torch.full((1,), 1e-30, device=DEV)
    dz = torch.randn((2, 8, 16, 64), generator=g).to(DEV)
    wpd = _hip.conv3d_pack((torch.randn((64, 64, 3, 3, 3), generator=g) * 0.05).to(DEV), True, split=4)
    dx = torch.empty((2, 8, 16, 64), device=DEV)
    X.check(X.lib.mvx_split_operand_amax(X.ptr(bad), None), 'mvx_split_operand_amax')
    # stride 3 is not a geometry of this library: rejected before any launch
    rc = X.lib.mvx_conv3d_dgrad_split(X.ptr(dz), X.ptr(wpd), X.ptr(dx), 2, 2, 8, 16, 64, 64, 3, 1, _hip.split_flags(4), X.stream())
    assert rc < 0
    out = torch.empty((1000, 256), device=DEV)
    X.check(X.lib.mvx_linear_forward(X.ptr(x), 128, X.ptr(w), 128, 0, None, X.ptr(out), 256, None, None, 1000, 128, 256,
                                     _hip.split_flags(4, True), None, 0, X.stream()), 'mvx_linear_forward')
    assert rel_err(out, x.double() @ w.double().t()) < 2e-6              # the stale 1e-30 range would have made this inf / NaN


def test_foreign_forward_inputs_take_the_coarse_scale_or_bf16x6(golden):
    """A forward scale must not depend on which tensor the executor holds: a frame set and one of its frames have different
    maxima, fine scales would round elements with subnormal low pieces differently and flip ReLUs between the two executors
    (tools/dbg_fp16_seeds.py: 3e-2 on 4 of 6 seeds with a fine forward scale, 8e-7 with this rule).  Inputs produced by this
    library's BatchNorms are in range as they are and take no tag; a FOREIGN input (the first fusion layer reads sampled image
    features; any stand-alone FCN) runs fp16x3 with the coarse scale (8-binade steps) when it carries a range tag, bf16x6
    otherwise.  Here: the rule, and values far outside fp16's range through such a layer, both ways."""
    from modules import _hip
    from modules.layers.Blocks import fcn_rows
    import modules.config as cfg
    t = torch.ones((8,), device=DEV)
    assert _hip.foreign_split(4) == (3, 0) and _hip.foreign_split(4, t) == (3, 0) and _hip.foreign_split(3, t) == (3, 0)
    assert _hip.foreign_split(0, t) == (0, 0)
    _hip.tensor_amax(t)
    assert _hip.foreign_split(4, t) == (4, _hip.FLAG_AMAX_COARSE)
    old = cfg.config.get('convmath', 'f32')
    cfg.config['convmath'] = 'fp16x3'
    try:
        g = torch.Generator().manual_seed(3)
        x0 = torch.randn((3000, 768), generator=g)
        w0 = (torch.randn((768, 768), generator=g) / 28)
        for mag in (3e5, 2e-6, 1.0):                                      # |x| up to ~1.5e6 (beyond fp16), ~1e-5 (under it), ~5
            for tagged in (True, False):
                x = (x0 * mag).to(DEV)
                if tagged:
                    _hip.tensor_amax(x)
                w = w0.clone().to(DEV).requires_grad_(True)
                b = torch.zeros((768,), device=DEV, requires_grad=True)
                l0 = _hip.X.lib.mvx_launch_count()
                out = fcn_rows(x, w, b)                                   # foreign=True is the default
                y = torch.relu(x.double() @ w.detach().double().t())
                ref = (y - y.mean(0)) / torch.sqrt(y.var(0, unbiased=False) + cfg.eps)
                assert bool(torch.isfinite(out).all()) and rel_err(out, ref) < 1e-5, (mag, tagged, rel_err(out, ref))
                out.square().sum().backward()
                assert bool(torch.isfinite(w.grad).all()), (mag, tagged)
        # the coarse scale: x and 2 x fall on the same 8-binade step, are cut into the same pieces (no element small enough for a
        # subnormal low piece here) and give results that differ by exactly that factor
        x1 = torch.sign(x0) * (0.5 + torch.rand(x0.shape, generator=g))
        xa, xb = (x1 * 3.0).to(DEV), (x1 * 3.0 * 2.0).to(DEV)             # a factor 2 is exact in every arithmetic
        _hip.tensor_amax(xa), _hip.tensor_amax(xb)
        wd = w0.to(DEV)
        ya, _ = _hip.linear_forward(xa, wd, None, relu=False, want_stats=False, split=4, foreign=True)
        yb, _ = _hip.linear_forward(xb, wd, None, relu=False, want_stats=False, split=4, foreign=True)
        assert torch.equal(ya * 2.0, yb)
    finally:
        cfg.config['convmath'] = old


def test_weight_range_guard_is_loud():
    """fp16 pieces carry weights times 2^8, so |w| >= 255.9 overflows the high piece -- behind a BatchNorm that
    would hide it.  Every weight an fp16x3 kernel reads is range-checked on the device (once per parameter version) into a
    status word; raise_on_status turns it into an error at the step's status check.  The same weight in bf16x6 is fine."""
    import modules.config as cfg
    from modules import _hip
    from modules import Extension as X
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((600, 128), generator=g).to(DEV)
    w = torch.nn.Parameter((torch.randn((256, 128), generator=g) * 0.1).to(DEV))
    st = _hip.fp16_weight_status(dev)
    st.zero_()
    _hip.linear_forward(x, w, None, relu=False, want_stats=False, split=4)
    assert int(st) == 0
    _hip.raise_on_status(int(st))                                  # nothing to report
    with torch.no_grad():
        w[3, 5] = 300.0
    y4, _ = _hip.linear_forward(x, w, None, relu=False, want_stats=False, split=4)
    assert int(st) & _hip.STATUS_F16_WEIGHT_RANGE
    assert not bool(torch.isfinite(y4).all())                      # what the guard is about: inf / NaN without any other sign
    with pytest.raises(X.MvxHipError, match='fp16x3'):
        _hip.raise_on_status(int(st))
    st.zero_()
    y6, _ = _hip.linear_forward(x, w, None, relu=False, want_stats=False, split=3)      # bf16 pieces: the range of f32
    assert int(st) == 0 and rel_err(y6, x.double() @ w.detach().double().t()) < 2e-6
    # ... and a convolution kernel packed for fp16 pieces
    wc = torch.nn.Parameter((torch.randn((64, 64, 3, 3, 3), generator=g) * 0.05).to(DEV))
    _hip.conv3d_pack(wc, False, split=4)
    assert int(st) == 0
    with torch.no_grad():
        wc[1, 2, 0, 1, 2] = -260.0
    _hip.conv3d_pack(wc, False, split=4)
    assert int(st) & _hip.STATUS_F16_WEIGHT_RANGE
    st.zero_()
