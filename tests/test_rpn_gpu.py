"""The region proposal network on HIP kernels (modules/rpn_frames.py, csrc/rpn.hip, mvx_conv2d_*_frames) against the
reference fixture and the CPU oracle: score / regression maps and every parameter gradient, frame sets of 1 and 3."""
import numpy as np
import pytest
import torch

import mvx_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


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
    assert float((score[0].cpu() - torch.from_numpy(g['score'])).abs().max()) < 1e-4
    assert rel(reg[0].cpu(), torch.from_numpy(g['reg'])) < 1e-4


@pytest.mark.parametrize('F', [1, 3])
def test_rpn_forward_backward_match_oracle(golden, F):
    from modules import _hip, parallel
    from modules import rpn_frames as rf
    P = O.rpn_params(golden('rpn_shapes'))
    rpn = _load_rpn(P)
    bucket = parallel.GradBucket(list(rpn.parameters()))
    H, W = 32, 48
    gen = torch.Generator().manual_seed(3)
    mids = torch.randn((F, 128, H, W), generator=gen)
    d_heads = torch.randn((F, H // 2, W // 2, 16), generator=gen) * 0.1
    bucket.zero()
    heads, S = rf.rpn_forward(rpn, _to_planes(mids.to(DEV)), F, 2, H, W, 64)
    g_in = rf.rpn_backward(rpn, S, d_heads.reshape(-1, 16).to(DEV))
    _hip.join_side_stream()
    torch.cuda.synchronize()
    g_mid = _from_planes(g_in, F).cpu()
    # oracle: F independent batch-1 forwards in float64 (per-frame BatchNorm), summed parameter gradients
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    for f in range(F):
        x = mids[f:f + 1].double().requires_grad_(True)
        score, reg = O.rpn(x, P64)
        logits = torch.log(score / (1 - score))
        out = torch.cat([logits, reg], dim=1)[0].permute(1, 2, 0)            # (h, w, 16)
        got = heads.view(F, H // 2, W // 2, 16)[f].cpu()
        assert rel(got[..., 2:], out[..., 2:].detach().float()) < 1e-4
        assert float((torch.sigmoid(got[..., :2]) - score[0].permute(1, 2, 0).detach().float()).abs().max()) < 1e-4
        (out * d_heads[f].double()).sum().backward()
        assert rel(g_mid[f], x.grad[0].float()) < 2e-3, f
    for k, p in rpn.named_parameters():
        ref = P64['rpn.' + k].grad.float()
        assert rel(p.grad.cpu(), ref) < 2e-3, k


def test_rpn_hip_agrees_with_the_module_path_at_full_size():
    """Full-size maps (400x352): the HIP frame-set RPN against the nn.Module RPN (stock PyTorch-ROCm) on the same weights."""
    from modules import rpn_frames as rf
    from modules.voxelnet.Pipe import RPN
    torch.manual_seed(0)
    rpn = RPN().to(DEV)
    mid = torch.randn((1, 128, 352, 400), device=DEV)
    with torch.no_grad():
        s_ref, r_ref = rpn(mid)
        heads, S = rf.rpn_forward(rpn, _to_planes(mid), 1, 2, 352, 400, 64)
        score, reg = rf.split_heads(heads, 1, S['h1'], S['w1'])
    assert float((score - s_ref).abs().max()) < 1e-4
    assert rel(reg, r_ref) < 1e-4
