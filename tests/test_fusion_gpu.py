"""Image-fusion branch on the GPU against the reference fixtures: featureMaping (dense and
compact), the fusion MLP with gradients, and MVXNet-minus-extractor end to end."""
import numpy as np
import pytest
import torch

import mvx_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel_err(a, b):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def test_feature_mapping_dense_matches_reference(golden):
    from modules.imhead import featureMaping
    g = golden('feature_mapping')
    vox = torch.from_numpy(g['voxels_in'].copy())[None].to(DEV)
    feats = [torch.from_numpy(g[k])[None].to(DEV) for k in ('f0', 'f1', 'f2')]
    out = featureMaping(vox, feats, [None], torch.from_numpy(g['imsize_hw']).to(DEV))[0]
    assert np.array_equal(vox[0].cpu().numpy(), g['voxels_after'])       # in-place side effect
    got = out.cpu().numpy()
    assert got.shape == g['out'].shape
    np.testing.assert_allclose(got, g['out'], rtol=1e-6, atol=1e-6)
    zero = np.all(g['voxels_after'][..., :3] == 0, axis=-1)
    assert np.all(got[zero] == 0)


def test_fusion_mlp_dense_and_compact_match_reference(golden):
    from modules.imhead import ImageFeatureFusion
    from modules.imhead.Pipe import ExpandRowsFunction
    g = golden('fusion')
    P = O.strip_prefix(O.make_params(7), 'head.fusion.')
    fus = ImageFeatureFusion()
    fus.load_state_dict(P)
    fus = fus.to(DEV)
    x = torch.from_numpy(g['x'])[None].to(DEV).requires_grad_(True)
    G = torch.from_numpy(g['G']).to(DEV)
    y = fus(x)
    assert rel_err(y[0], g['out']) < 1e-4
    (y[0] * G).sum().backward()
    dense_grads = {k: p.grad.clone() for k, p in fus.named_parameters()}
    for k, p in fus.named_parameters():
        if 'grad.' + k in g.files:
            assert rel_err(p.grad, g['grad.' + k]) < 5e-3, k
        else:
            assert rel_err(p.grad.reshape(p.shape[0], -1)[:8, :64], g['gradslice.' + k]) < 5e-3, k
    assert rel_err(x.grad[0][:, :, :32], g['grad_x']) < 5e-3

    # compact evaluation: real rows + one shared zero row with BatchNorm weight = #padded rows
    fus.zero_grad()
    xd = torch.from_numpy(g['x']).reshape(-1, 768)
    real = (xd.abs().sum(1) != 0)
    nr = int(real.sum())
    row_map = torch.full((xd.shape[0],), -1, dtype=torch.int32)
    row_map[real] = torch.arange(nr, dtype=torch.int32)
    comp = torch.cat([xd[real], torch.zeros(1, 768)], 0).to(DEV)
    row_w = torch.ones(nr + 1, device=DEV)
    row_w[nr] = xd.shape[0] - nr
    yc = fus.forward_rows(comp, row_w, xd.shape[0])
    yd = ExpandRowsFunction.apply(yc, row_map.to(DEV), nr)
    assert rel_err(yd, y[0].reshape(-1, 16)) < 2e-5
    (yd * G.reshape(-1, 16)).sum().backward()
    for k, p in fus.named_parameters():
        assert rel_err(p.grad, dense_grads[k]) < 1e-4, k


def test_mvxnet_without_extractor_matches_reference(golden):
    import modules.config as cfg
    from MVXNet import MVXNet
    g = golden('mvxnet_small')
    old = list(cfg.config['voxelshape'])
    cfg.config['voxelshape'] = [int(v) for v in g['voxelshape']]
    try:
        model = MVXNet()
        P = O.make_params(7)
        sd = model.state_dict()
        for k in sd:
            if k in P:
                sd[k] = P[k]
        model.load_state_dict(sd)
        model = model.to(DEV)
        vox = torch.from_numpy(g['voxels'].copy())[None].to(DEV)
        feats = [torch.from_numpy(g[k])[None].to(DEV) for k in ('f0', 'f1', 'f2')]
        idx = torch.from_numpy(g['idx']).to(DEV)
        imsize = torch.from_numpy(g['imsize_hw']).to(DEV)
        v23 = model.point_features(vox, feats, [None], imsize)
        assert rel_err(v23[0], g['v23']) < 1e-4
        feat = model.backbone.voxel_features(v23)
        assert rel_err(feat, g['feat']) < 2e-4
        mid = model.backbone.middle(v23, idx)
        assert rel_err(mid[0], g['mid']) < 2e-4
        (mid[0] * torch.from_numpy(g['G']).to(DEV)).sum().backward()
        # Parameter gradients.  On this tiny grid the reference's OWN fp32 gradients sit 1-6 % away
        # from exact arithmetic (bias gradients in front of a BatchNorm are pure cancellation), so
        # the yardstick is the float64 oracle: the HIP path may not be further from it than 3x the
        # reference's fp32 rounding noise.
        g64 = golden('mvxnet_small_f64')
        for k, p in model.named_parameters():
            if p.grad is None or 'rpn' in k:
                continue
            if 'grad.' + k in g.files:
                e_ref = rel_err(g['grad.' + k], g64['grad.' + k])
                e_hip = rel_err(p.grad, g64['grad.' + k])
                assert e_hip < 3 * e_ref + 1e-3, (k, e_hip, e_ref)
            else:
                _, l1 = g['gradproj.' + k]
                gn = p.grad.cpu().numpy().astype(np.float64)
                assert abs(np.abs(gn).sum() - l1) / l1 < 3e-2, k
        # compact evaluation of fusion + VFE (real rows + one padded row per voxel) == dense evaluation
        dense_grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        model.zero_grad()
        vox2 = torch.from_numpy(g['voxels'].copy())[None].to(DEV)
        mid_c = model.middle(vox2, feats, idx, [None], imsize, compact=True)
        assert rel_err(mid_c, mid) < 2e-5
        (mid_c[0] * torch.from_numpy(g['G']).to(DEV)).sum().backward()
        for k, p in model.named_parameters():
            if k not in dense_grads:
                continue
            if 'grad.' + k in g64.files:                   # same float64 yardstick as above
                e_ref = rel_err(g['grad.' + k], g64['grad.' + k])
                e_hip = rel_err(p.grad, g64['grad.' + k])
                assert e_hip < 3 * e_ref + 1e-3, (k, e_hip, e_ref)
            else:
                assert rel_err(p.grad, dense_grads[k]) < 2e-2, k
        # f64 oracle: the 1e-4 bar against exact arithmetic
        P64 = {k: v.double() for k, v in P.items()}
        vox64 = torch.from_numpy(g['voxels'].copy()).double()
        v23r = O.mvx_point_features(vox64, [torch.from_numpy(g[k]).double() for k in ('f0', 'f1', 'f2')],
                                    torch.from_numpy(g['imsize_hw']).double(), P64)
        assert rel_err(v23[0], v23r) < 1e-4
    finally:
        cfg.config['voxelshape'] = old


def test_mvxnet_forward_compact_equals_the_dense_formulation(golden):
    """MVXNet.forward (MVXNet.py:21-27) -- the interface train.py calls -- runs fusion + VFE on compact rows, the first CML
    layer on the voxel rows, the CML chain with the tile-restricted backward and the RPN as one HIP node; ``compact=False``
    is the reference's dense (1,N,T,23) formulation through the same modules.  Score / regression maps and every parameter
    gradient of the two must agree (the in-place zeroing of the padded voxel rows included)."""
    import modules.config as cfg
    from MVXNet import MVXNet
    g = golden('mvxnet_small')
    old = list(cfg.config['voxelshape'])
    cfg.config['voxelshape'] = [int(v) for v in g['voxelshape']]
    try:
        torch.manual_seed(4)
        model = MVXNet().to(DEV)
        feats = [torch.from_numpy(g[k])[None].to(DEV) for k in ('f0', 'f1', 'f2')]
        idx = torch.from_numpy(g['idx']).to(DEV)
        imsize = torch.from_numpy(g['imsize_hw']).to(DEV)
        res = {}
        from modules import _hip
        launches = {}
        for compact in (True, 'modules', False):           # single autograd node / per-module path on compact rows / dense
            model.zero_grad()
            l0 = _hip.X.lib.mvx_launch_count()
            vox = torch.from_numpy(g['voxels'].copy())[None].to(DEV)
            score, reg = model(vox, feats, idx, [None], imsize, compact=compact)
            assert score.shape[:2] == (1, 2) and reg.shape[:2] == (1, 14)
            (score.square().sum() + reg.square().sum()).backward()
            torch.cuda.synchronize()
            launches[compact] = _hip.X.lib.mvx_launch_count() - l0
            res[compact] = (score.detach().clone(), reg.detach().clone(), vox.clone(),
                            {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        assert all(v > 100 for v in launches.values()), launches          # every variant runs this library's kernels
        for variant in (True, 'modules'):
            assert torch.equal(res[variant][2], res[False][2])                # padded rows zeroed in place, every way
            assert rel_err(res[variant][0], res[False][0]) < 1e-4 and rel_err(res[variant][1], res[False][1]) < 1e-3
            assert set(res[variant][3]) == set(res[False][3])
            for k in res[False][3]:
                a, b = res[variant][3][k].double(), res[False][3][k].double()
                # tiny grid: 16 BatchNorms over <= 96 sites in the RPN amplify fp32 rounding; compared in the 2-norm
                assert float((a - b).norm() / b.norm().clamp_min(1e-30)) < 5e-2, (variant, k)
        # no-grad call: the same maps, status words checked inside
        with torch.no_grad():
            vox = torch.from_numpy(g['voxels'].copy())[None].to(DEV)
            s2, r2 = model(vox, feats, idx, [None], imsize)
        assert rel_err(s2, res[True][0]) < 1e-6 and rel_err(r2, res[True][1]) < 1e-6
    finally:
        cfg.config['voxelshape'] = old
