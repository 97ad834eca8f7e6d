"""The float4 forms of the VFE glue kernels (csrc/vfe.hip: vfe_bn_max4, vfe_max_concat_bwd4, segmax_bwd4) against the scalar forms
they replace on 16-byte aligned tensors: the same entry points on tensors that start 4 bytes off alignment run the scalar kernels,
and the two must agree BIT FOR BIT (values, argmax with ties, gradient sums in row order) -- dense rows and compact rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _off(t):
    """A copy of t whose storage starts 4 bytes past a 16-byte boundary (the entry points then take the scalar kernels)."""
    buf = torch.empty((t.numel() + 1,), dtype=t.dtype, device=t.device)
    v = buf[1:].view(t.shape)
    v.copy_(t)
    assert v.data_ptr() % 16 == 4
    return v


@pytest.mark.parametrize('C', [16, 32, 64, 128])
@pytest.mark.parametrize('compact', [False, True])
def test_vector_and_scalar_vfe_kernels_agree_bit_for_bit(C, compact):
    from modules import Extension as X
    g = torch.Generator().manual_seed(C + int(compact))
    V, T = 700, 35
    cnt = torch.randint(1, T + 1, (V,), generator=g)
    cnt[:5] = T                                              # full voxels: no padded row takes part in the max
    if compact:
        n_real = int(cnt.sum())
        voff = (torch.cumsum(cnt, 0) - cnt).to(torch.int32).to(DEV)
        vcnt = cnt.to(torch.int32).to(DEV)
        rows = n_real + V
        vo, vc = X.ptr(voff), X.ptr(vcnt)
    else:
        n_real, rows, vo, vc = 0, V * T, None, None
    y = torch.randn((rows, C), generator=g)
    y[torch.rand((rows, C), generator=g) < 0.3] = 0.5       # ties: the first maximum must win in both forms
    y = y.to(DEV)
    mi = torch.stack([torch.randn((C,), generator=g) * 0.1, torch.rand((C,), generator=g) + 0.5]).to(DEV).contiguous()
    res = {}
    for kind, wrap in (('vec', lambda t: t.clone()), ('scalar', _off)):
        yy = wrap(y)
        out, am = wrap(torch.zeros((rows, 2 * C), device=DEV)), wrap(torch.zeros((V, C), dtype=torch.int32, device=DEV))
        X.check(X.lib.mvx_vfe_bn_max_concat(X.ptr(yy), X.ptr(mi), X.ptr(out), X.ptr(am), V, T, C, vo, vc, n_real, X.stream()), 'concat')
        feat, am2 = wrap(torch.zeros((V, C), device=DEV)), wrap(torch.zeros((V, C), dtype=torch.int32, device=DEV))
        X.check(X.lib.mvx_bn_segment_max(X.ptr(yy), X.ptr(mi), X.ptr(feat), X.ptr(am2), V, T, C, vo, vc, n_real, X.stream()), 'segmax')
        gup = wrap(torch.randn((rows, 2 * C), generator=torch.Generator().manual_seed(1)).to(DEV))
        dyh = wrap(torch.zeros((rows, C), device=DEV))
        X.check(X.lib.mvx_vfe_max_concat_backward(X.ptr(gup), X.ptr(am), X.ptr(dyh), V, T, C, vo, vc, n_real, X.stream()), 'concat bwd')
        df = wrap(torch.randn((V, C), generator=torch.Generator().manual_seed(2)).to(DEV))
        dyh2 = wrap(torch.full((rows, C), 7.0, device=DEV))
        X.check(X.lib.mvx_segment_max_backward(X.ptr(df), X.ptr(am2), X.ptr(dyh2), V, T, C, vo, vc, n_real, X.stream()), 'segmax bwd')
        torch.cuda.synchronize()
        res[kind] = [t.clone() for t in (out, am, feat, am2, dyh, dyh2)]
    for a, b, name in zip(res['vec'], res['scalar'], ('concat', 'argmax', 'segmax', 'argmax of segmax', 'concat backward', 'segmax backward')):
        if compact or name not in ('concat',):
            assert torch.equal(a, b), name
        else:
            assert torch.equal(a, b), name
    # and against a plain restatement of the forward (dense layout)
    if not compact:
        yh = ((y - mi[0]) * mi[1]).view(V, T, C)
        mx = yh.max(1).values
        assert torch.equal(res['vec'][2], mx)
        assert torch.equal(res['vec'][0].view(V, T, 2 * C)[..., C:], mx[:, None, :].expand(V, T, C))


@pytest.mark.parametrize('W', [48, 400, 37, 50])
@pytest.mark.parametrize('sd,pd,din', [(1, 0, 5), (2, 1, 5), (2, 1, 10), (1, 1, 3)])
def test_activity_dilation_vector_and_scalar_forms_match_max_pooling(W, sd, pd, din):
    """mvx_activity_dilate (csrc/activity.hip): a site of the output is active iff any site of its 3x3x3 receptive field is (or
    it touches the image border, with mark_border) -- for W % 4 == 0 the four-sites-per-thread kernel runs, otherwise the scalar
    one; both against torch max-pooling, from an index grid and from a byte mask."""
    from modules import _hip
    import torch.nn.functional as F
    H = 40
    g = torch.Generator().manual_seed(W + din)
    act = (torch.rand((din, H, W), generator=g) < 0.02)
    act[0, 0, 0] = True
    act[-1, H - 1, W - 1] = True
    dout = _hip.conv_out_depth(din, sd, pd)
    ref = F.max_pool3d(act[None, None].float(), 3, (sd, 1, 1), (pd, 1, 1))[0, 0] > 0
    assert ref.shape[0] == dout
    idx = torch.where(act, torch.arange(act.numel()).view(act.shape), torch.full(act.shape, -1)).to(torch.int32).to(DEV)
    for src, is_index in ((idx, True), (act.to(torch.uint8).to(DEV), False)):
        for border in (False, True):
            mask, hflag = _hip.activity_dilate(src, is_index, din, H, W, sd, pd, mark_border=border)
            want = ref.clone()
            if border:
                want[:, 0, :] = True
                want[:, -1, :] = True
                want[:, :, 0] = True
                want[:, :, -1] = True
            assert torch.equal(mask.cpu().bool(), want), (is_index, border)
