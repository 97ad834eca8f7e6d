"""GPU parity of the MFMA conv3d / BatchNorm / scatter kernels against the CPU oracle
(torch-CPU conv3d + batch_norm) and the reference fixtures.  fp32, tolerance 1e-4 relative
to the tensor scale (north_star: 'within 1e-4 rel for fp32 features')."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mvx_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def to_cl(x_ncdhw):              # (C,D,H,W) -> (D,H,W,C) contiguous
    return x_ncdhw.permute(1, 2, 3, 0).contiguous()


GEOMS = [  # cin, cout, din, H, W, sd, pd
    (128, 64, 10, 16, 24, 2, 1),
    (64, 64, 5, 16, 24, 1, 0),
    (64, 64, 3, 16, 24, 2, 1),
    (64, 64, 4, 11, 37, 1, 1),      # ragged patch edges
    (32, 128, 3, 9, 17, 1, 1),      # cout = 128 -> two channel blocks
]


@pytest.mark.parametrize('cin,cout,din,H,W,sd,pd', GEOMS)
def test_conv3d_forward_dgrad_wgrad(cin, cout, din, H, W, sd, pd):
    from modules import _hip
    g = torch.Generator().manual_seed(cin * 7 + H)
    x = torch.randn((cin, din, H, W), generator=g)
    x[:, :, ::3, ::2] = 0                      # some exact zeros, like the sparse grid
    w = torch.randn((cout, cin, 3, 3, 3), generator=g) / np.sqrt(27 * cin)
    b = torch.randn((cout,), generator=g) * 0.1
    x.requires_grad_(True); w.requires_grad_(True); b.requires_grad_(True)
    y = F.relu(F.conv3d(x[None].double(), w.double(), b.double(), (sd, 1, 1), (pd, 1, 1)))[0]
    G = torch.randn(y.shape, generator=g).double()
    # gradient of sum(relu(conv) * G): dz = G * (y > 0)
    (y * G).sum().backward()
    dz_ref = (G * (y > 0)).float()

    xc = to_cl(x.detach()).to(DEV)
    wd = w.detach().to(DEV)
    wpk = _hip.conv3d_pack(wd, False)
    out, stats = _hip.conv3d_forward(xc, wpk, b.detach().to(DEV), cout, sd, pd, relu=True, want_stats=True)
    got = out.cpu().permute(3, 0, 1, 2)
    assert rel_err(got, y.detach()) < 1e-5
    st = stats.sum(0).cpu().numpy()
    yf = y.detach().numpy().reshape(cout, -1)
    np.testing.assert_allclose(st[0], yf.sum(1), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(st[1], (yf ** 2).sum(1), rtol=1e-5, atol=1e-3)

    dzc = to_cl(dz_ref).to(DEV)
    if cout == 64:
        dw = _hip.conv3d_wgrad(xc, dzc, sd, pd).cpu()
        assert rel_err(dw, w.grad) < 1e-5
    if cin % 64 == 0:
        wpk_d = _hip.conv3d_pack(wd, True)
        dx = _hip.conv3d_dgrad(dzc, wpk_d, din, cin, sd, pd).cpu().permute(3, 0, 1, 2)
        assert rel_err(dx, x.grad) < 1e-5


def test_batchnorm_relu_forward_backward():
    from modules import _hip
    g = torch.Generator().manual_seed(3)
    for rows, C in ((5000, 64), (1234, 16), (777, 128), (300, 768)):
        z = torch.randn((rows, C), generator=g).double() * 2 + 0.3
        z.requires_grad_(True)
        y = F.relu(z)
        yh = F.batch_norm(y.T[None], None, None, None, None, True, 0.0, 1e-6)[0].T
        G = torch.randn((rows, C), generator=g).double()
        (yh * G).sum().backward()
        yd = y.detach().float().to(DEV)
        stats = _hip.row_stats(yd)
        mi = _hip.bn_finalize(stats, rows, 1e-6)
        out = _hip.bn_apply(yd, mi)
        assert rel_err(out.cpu(), yh.detach()) < 2e-5
        dz, db = _hip.bn_relu_backward(G.float().to(DEV), yd, mi, rows)
        assert rel_err(dz.cpu(), z.grad) < 2e-5
        assert rel_err(db.cpu(), z.grad.sum(0)) < 1e-4


def test_scatter_gather_roundtrip(golden):
    from modules import _hip
    g = golden('voxelnet_small')
    idx = torch.from_numpy(g['idx']).to(DEV)
    feat = torch.from_numpy(g['feat']).to(DEV)
    H, W, D = [int(v) for v in g['voxelshape']]
    grid, status = _hip.scatter_voxels(feat, idx, (D, H, W))
    assert int(status) == 0
    ref = O.reindex(torch.from_numpy(g['feat']), torch.from_numpy(g['idx']), [H, W, D])[0]   # (C,D,H,W)
    assert torch.equal(grid.cpu(), ref.permute(1, 2, 3, 0))       # pure data movement: bit-exact
    back = _hip.gather_voxels(grid, idx, feat.shape[0])
    assert torch.equal(back, feat)


def test_cml_stack_matches_reference_fixture(golden):
    """conv1..conv3 (+ReLU+BN) of the reference run on the fixture's dense grid."""
    from modules import _hip
    g = golden('voxelnet_small')
    P = O.strip_prefix(O.make_params(7), 'backbone.')
    H, W, D = [int(v) for v in g['voxelshape']]
    grid, _ = _hip.scatter_voxels(torch.from_numpy(g['feat']).to(DEV), torch.from_numpy(g['idx']).to(DEV), (D, H, W))
    x = grid
    for (name, stride, pad), key in zip(O.CML_GEOM, ('conv1', 'conv2', 'conv3')):
        w = P[name + '.weight'].to(DEV)
        b = P[name + '.bias'].to(DEV)
        y, stats = _hip.conv3d_forward(x, _hip.conv3d_pack(w, False), b, w.shape[0], stride[0], pad[0])
        mi = _hip.bn_finalize(stats, y.numel() // y.shape[-1], 1e-6)
        x = _hip.bn_apply(y, mi)
        ref = g[key]                                  # (C, D, H, W)
        assert rel_err(x.cpu().permute(3, 0, 1, 2), ref) < 1e-4, key


@pytest.fixture(autouse=True, params=['32-channel units', '64-channel units'])
def gather_units(request):
    """The f32 gather runs small launches as 32-channel ("narrow") workgroup units (csrc/conv3d.hip launch_gather,
    MVX_TUNE_GATHER_NARROW_MAX_UNITS = key 2) -- each test of this file runs with narrow units forced and with
    narrow units off."""
    from modules import Extension as X
    X.check(X.lib.mvx_tuning_set(2, (1 << 60) if request.param.startswith('32') else 0), 'mvx_tuning_set')
    yield request.param
    X.check(X.lib.mvx_tuning_set(2, -1), 'mvx_tuning_set')


@pytest.fixture(params=['8x16 units', '16x16 units'])
def split_units(request):
    """The bf16x3 gather has two workgroup shapes (csrc/conv3d_split.hip: 8 x 16 sites, and 16 x 16 sites with 64-site wave
    tiles, picked by the number of units of a launch): run the test with each forced."""
    from modules import Extension as X
    X.check(X.lib.mvx_tuning_set(1, 0 if request.param.startswith('16') else 1 << 60), 'mvx_tuning_set')      # MVX_TUNE_SPLIT16_MIN_UNITS
    yield request.param
    X.check(X.lib.mvx_tuning_set(1, 768), 'mvx_tuning_set')


@pytest.mark.parametrize('pieces,tol', [(2, 2e-5), (3, 4e-6), (4, 4e-6)])
@pytest.mark.parametrize('cin,cout,din,H,W,sd,pd', GEOMS[:4])
def test_conv3d_bf16x3_split_accuracy(cin, cout, din, H, W, sd, pd, split_units, pieces, tol):
    """Split kernels against float64.  bf16x3 (two pieces): 2e-5, well inside the 1e-4 feature bar; bf16x6 (three pieces = the
    whole f32 mantissa, six MFMAs per product): 4e-6 (an f32 accumulation chain over up to K = 3,456 products) AND never more
    than twice the exact-f32 kernel's own distance from float64 on the same inputs (+ 5e-7).  fp16x3 (code 4: two fp16 pieces,
    22 mantissa bits, three MFMAs; operands of unit scale need no range tag): the bf16x6 bounds."""
    from modules import _hip
    split = pieces
    g = torch.Generator().manual_seed(cin + 3 * H)
    x = torch.randn((cin, din, H, W), generator=g)
    w = torch.randn((cout, cin, 3, 3, 3), generator=g) / np.sqrt(27 * cin)
    b = torch.randn((cout,), generator=g) * 0.1
    y = F.relu(F.conv3d(x[None].double(), w.double(), b.double(), (sd, 1, 1), (pd, 1, 1)))[0]
    xc = to_cl(x).to(DEV)
    wd = w.to(DEV)
    out, stats = _hip.conv3d_forward(xc, _hip.conv3d_pack(wd, False, split=split), b.to(DEV), cout, sd, pd, split=split)
    assert rel_err(out.cpu().permute(3, 0, 1, 2), y) < tol
    if pieces >= 3:
        o32, _ = _hip.conv3d_forward(xc, _hip.conv3d_pack(wd, False), b.to(DEV), cout, sd, pd)
        assert rel_err(out.cpu().permute(3, 0, 1, 2), y) < 2 * rel_err(o32.cpu().permute(3, 0, 1, 2), y) + 5e-7
    np.testing.assert_allclose(stats.sum(0)[0].cpu().numpy(), y.numpy().reshape(cout, -1).sum(1), rtol=1e-4, atol=5e-2)
    if cin % 64 == 0:
        dz = torch.randn(y.shape, generator=g)
        xg = x[None].double().requires_grad_(True)
        F.conv3d(xg, w.double(), None, (sd, 1, 1), (pd, 1, 1)).backward(dz[None].double())
        dx = _hip.conv3d_dgrad(to_cl(dz).to(DEV), _hip.conv3d_pack(wd, True, split=split), din, cin, sd, pd, split=split)
        assert rel_err(dx.cpu().permute(3, 0, 1, 2), xg.grad[0]) < tol
    if cout == 64:
        dz = torch.randn(y.shape, generator=g)
        wg = w.double().requires_grad_(True)
        F.conv3d(x[None].double(), wg, None, (sd, 1, 1), (pd, 1, 1)).backward(dz[None].double())
        dw = _hip.conv3d_wgrad(xc, to_cl(dz).to(DEV), sd, pd, split=split)
        assert rel_err(dw.cpu(), wg.grad) < tol


@pytest.mark.parametrize('split', [False, 2, 3, 4])
def test_conv3d_full_size_adjoint_identities(split):
    """BASELINE-size grid (conv2 geometry, 5x352x400x64): the three passes must be mutually adjoint,
    <dz, conv(x)> = <dgrad(dz), x> = <wgrad(x, dz), w> -- a size-independent check that needs no CPU
    reference.  Inner products in float64 on the GPU."""
    from modules import _hip
    g = torch.Generator().manual_seed(9)
    H, W, cin, cout, din, sd, pd = 352, 400, 64, 64, 5, 1, 0
    dout = _hip.conv_out_depth(din, sd, pd)
    x = torch.randn((din, H, W, cin), generator=g).to(DEV)
    dz = torch.randn((dout, H, W, cout), generator=g).to(DEV)
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) / np.sqrt(27 * cin)).to(DEV)
    y, _ = _hip.conv3d_forward(x, _hip.conv3d_pack(w, False, split=split), None, cout, sd, pd, relu=False,
                               want_stats=False, split=split)
    dx = _hip.conv3d_dgrad(dz, _hip.conv3d_pack(w, True, split=split), din, cin, sd, pd, split=split)
    dw = _hip.conv3d_wgrad(x, dz, sd, pd, split=split)
    a = float((dz.double() * y.double()).sum())
    b = float((dx.double() * x.double()).sum())
    c = float((dw.double() * w.double()).sum())
    scale = float(dz.double().norm() * y.double().norm())
    tol = 2e-5 if split == 2 else 2e-6            # bf16x6 and fp16x3 are held to the exact-f32 bound
    assert abs(a - b) / scale < tol and abs(a - c) / scale < tol, (a, b, c, scale)


@pytest.mark.gpu
@pytest.mark.parametrize('pieces', [2, 3, 4])
@pytest.mark.parametrize('shape', [(5, 40, 48, 1, 0), (3, 37, 53, 2, 1), (10, 24, 35, 2, 1)])
def test_background_rewrite_equals_dense(shape, split_units, pieces):
    """conv3d_forward_bg / conv3d_wgrad_bg (constant fill of voxel-free tiles, closed-form constant term)
    against the dense kernels on an input that IS a background plus a few active sites -- including
    sizes that are no multiple of the 8x16 tile and depth padding."""
    from modules import _hip
    din, H, W, sd, pd = shape
    cin = cout = 64
    dev = torch.device('cuda')
    g = torch.Generator(device='cpu').manual_seed(5)
    dout = _hip.conv_out_depth(din, sd, pd)
    # source activity: a few sites, with the per-plane constant everywhere else
    act = torch.zeros((din, H, W), dtype=torch.uint8)
    for _ in range(6):
        act[int(torch.randint(0, din, (1,), generator=g)), int(torch.randint(0, H, (1,), generator=g)),
            int(torch.randint(0, W, (1,), generator=g))] = 1
    act[0, 0, 0] = 1                                   # a corner site
    c_in = torch.randn((din, cin), generator=g)
    x = c_in[:, None, None, :].expand(din, H, W, cin).clone()
    noise = torch.randn((din, H, W, cin), generator=g)
    x = torch.where(act[..., None].bool(), noise, x).contiguous().to(dev)
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.05).to(dev)
    b = torch.randn((cout,), generator=g).to(dev)
    act_d, c_d = act.to(dev), c_in.to(dev)
    # halo flags of the source tensor itself: dilate an "index grid" stand-in through an identity? no --
    # build them with the library: flags of a mask are produced for the DST of a dilation, so compute the
    # source flags from the mask directly here (test-side restatement)
    th, tw = 8, 16
    ty, tx = (H + th - 1) // th, (W + tw - 1) // tw
    hflag = torch.zeros((din, ty * tx), dtype=torch.int32)
    for d in range(din):
        for t in range(ty * tx):
            y0, x0 = (t // tx) * th - 1, (t % tx) * tw - 1
            sub = act[d, max(y0, 0):min(y0 + th + 2, H), max(x0, 0):min(x0 + tw + 2, W)]
            hflag[d, t] = int(sub.any())
    bg_in = _hip.Background(c_d, act_d, hflag.to(dev))
    out_mask, out_hflag = _hip.activity_dilate(act_d, False, din, H, W, sd, pd, mark_border=True)
    # the library's halo flags of the OUTPUT follow the same definition as the test-side ones of the input
    om = out_mask.cpu()
    for d in range(dout):
        for t in range(ty * tx):
            y0, x0 = (t // tx) * th - 1, (t % tx) * tw - 1
            assert int(out_hflag[d, t]) == int(om[d, max(y0, 0):min(y0 + th + 2, H), max(x0, 0):min(x0 + tw + 2, W)].any())
    wpk = _hip.conv3d_pack(w, False)
    bg_pre = _hip.conv3d_background(w, c_d, din, sd, pd)
    y_bg, st_bg = _hip.conv3d_forward_bg(x, wpk, b, cout, sd, pd, bg_in, out_mask, bg_pre)
    y_dn, st_dn = _hip.conv3d_forward(x, wpk, b, cout, sd, pd)
    scale = float(y_dn.abs().max())
    assert float((y_bg - y_dn).abs().max()) < 2e-5 * scale
    assert torch.allclose(st_bg.sum(0), st_dn.sum(0), rtol=1e-5, atol=1e-5 * float(st_dn.sum(0).abs().max()))
    # background sites are bit-equal to one another (what the wgrad skip relies on)
    bgsites = (out_mask == 0)
    for d in range(dout):
        rows = y_bg[d][bgsites[d]]
        if rows.shape[0] > 1:
            assert bool((rows == rows[0]).all())
    dz = torch.randn((dout, H, W, cout), generator=g).to(dev)
    dw_bg = _hip.conv3d_wgrad_bg(x, dz, sd, pd, bg_in)
    dw_dn = _hip.conv3d_wgrad(x, dz, sd, pd)
    assert float((dw_bg - dw_dn).abs().max()) < 2e-5 * float(dw_dn.abs().max())
    # the bf16x3 forms of the same two entry points against their dense bf16x3 counterparts
    wps = _hip.conv3d_pack(w, False, split=pieces)
    ys_bg, _ = _hip.conv3d_forward_bg(x, wps, b, cout, sd, pd, bg_in, out_mask, bg_pre, split=pieces)
    ys_dn, _ = _hip.conv3d_forward(x, wps, b, cout, sd, pd, split=pieces)
    assert float((ys_bg - ys_dn).abs().max()) < 1e-4 * scale
    dws_bg = _hip.conv3d_wgrad_bg(x, dz, sd, pd, bg_in, split=pieces)
    dws_dn = _hip.conv3d_wgrad(x, dz, sd, pd, split=pieces)
    assert float((dws_bg - dws_dn).abs().max()) < 1e-4 * float(dws_dn.abs().max())
    assert float((dws_bg - dw_dn).abs().max()) < 1e-4 * float(dw_dn.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('cin,cout,H,W', [(128, 128, 50, 44), (128, 256, 25, 22), (256, 256, 19, 35)])
def test_conv2d_3x3_on_the_conv3d_kernels(cin, cout, H, W):
    """nn.Conv2d 3x3 / stride 1 / padding 1 (the RPN blocks) as a depth-1 conv3d with pad_d = 1: forward, input
    gradient and 2-D weight gradient against torch (CPU, float64 accumulate)."""
    from modules import _hip
    import torch.nn.functional as F
    dev = torch.device('cuda')
    g = torch.Generator(device='cpu').manual_seed(9)
    x = torch.randn((H, W, cin), generator=g)
    w = torch.randn((cout, cin, 3, 3), generator=g) * 0.05
    b = torch.randn((cout,), generator=g)
    dz = torch.randn((H, W, cout), generator=g)
    xd, wd, bd, dzd = x.to(dev), w.to(dev), b.to(dev), dz.to(dev)
    y, _ = _hip.conv3d_forward(xd.unsqueeze(0), _hip.conv3d_pack(wd, False), bd, cout, 1, 1, relu=False, want_stats=False)
    dx = _hip.conv3d_dgrad(dzd.unsqueeze(0), _hip.conv3d_pack(wd, True), 1, cin, 1, 1)
    dw = _hip.conv3d_wgrad(xd.unsqueeze(0), dzd.unsqueeze(0), 1, 1, two_d=True)
    assert dw.shape == (cout, cin, 3, 3)
    xr = x.double().permute(2, 0, 1)[None].requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, b.double(), padding=1)
    yr.backward(dz.double().permute(2, 0, 1)[None])
    assert rel_err(y[0].cpu(), yr[0].permute(1, 2, 0).detach()) < 1e-5
    assert rel_err(dx[0].cpu(), xr.grad[0].permute(1, 2, 0)) < 1e-5
    assert rel_err(dw.cpu(), wr.grad) < 1e-5


@pytest.mark.gpu
def test_split_gather_workgroup_shapes_give_identical_outputs():
    """The two workgroup shapes of the bf16x3 gather accumulate every output in the same order (stage by stage, tap by tap,
    k by k): forward and input gradient must be BIT-identical, on a size that is no multiple of either tile (the last
    16 x 16 unit covers a single 8 x 16 tile) and with more than one output-channel block."""
    from modules import _hip
    from modules import Extension as X
    g = torch.Generator().manual_seed(21)
    din, H, W, cin, cout, sd, pd = 3, 40, 53, 64, 128, 1, 1
    x = torch.randn((din, H, W, cin), generator=g).to(DEV)
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) / np.sqrt(27 * cin)).to(DEV)
    b = (torch.randn((cout,), generator=g) * 0.1).to(DEV)
    dz = torch.randn((din, H, W, cout), generator=g).to(DEV)
    wf, wd = _hip.conv3d_pack(w, False, split=True), _hip.conv3d_pack(w, True, split=True)
    res = {}
    try:
        for tag, val in (('8', 1 << 60), ('16', 0)):
            X.check(X.lib.mvx_tuning_set(1, val), 'mvx_tuning_set')
            y, st = _hip.conv3d_forward(x, wf, b, cout, sd, pd, split=True)
            dx = _hip.conv3d_dgrad(dz, wd, din, cin, sd, pd, split=True)
            torch.cuda.synchronize()
            res[tag] = (y.clone(), st.sum(0).clone(), dx.clone())
    finally:
        X.check(X.lib.mvx_tuning_set(1, 768), 'mvx_tuning_set')
    assert torch.equal(res['8'][0], res['16'][0]) and torch.equal(res['8'][2], res['16'][2])
    assert torch.allclose(res['8'][1], res['16'][1], rtol=1e-6, atol=1e-6 * float(res['8'][1].abs().max()))
