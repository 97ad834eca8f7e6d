"""VFE / SVFE / CML / RPN with the reference's class names and attributes
(modules/voxelnet/Pipe.py), on HIP kernels.

A VFE layer is ONE autograd node: row GEMM (+ReLU, + BatchNorm sums in the epilogue) ->
fused normalise + per-voxel max + concat.  Its backward is the mirrored chain.  The max runs
over all T rows, padded rows included, exactly like Pipe.py:14-16."""
import torch
from torch import nn

import modules.config as cfg
from modules import _hip
from modules.layers import FCN, CRB2d, CRB3d, DeCRB2d
from modules.layers.Blocks import _as_rows


class VFEFunction(torch.autograd.Function):
    """rows -> [BN(ReLU(fc rows)), per-voxel max].  ``cr``: CompactRows or None (dense [V][T] rows)."""

    @staticmethod
    def forward(ctx, x, w, b, V, T, eps, cr):
        row_w = cr.row_w if cr is not None else None
        y, mi = _hip.linear_forward(x, w, b, relu=True, want_stats=True, row_w=row_w, finalize=(V * T, eps),
                                    split=_hip.row_split('vfe'))
        out, am = _hip.vfe_bn_max_concat(y, mi, V, T, cr)
        ctx.save_for_backward(x, w, y, mi, am)
        ctx.vt = (V, T, cr)
        ctx.params = (w, b)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, y, mi, am = ctx.saved_tensors
        V, T, cr = ctx.vt
        dyh = _hip.vfe_max_concat_backward(g.contiguous(), am, V, T, cr)
        dz, db = _hip.bn_relu_backward(dyh, y, mi, V * T, True, dz=dyh, row_w=cr.row_w if cr is not None else None,
                                       dbias_out=_hip.bias_sink_of(ctx.params[1]))
        db = _hip.accumulate_grad(ctx.params[1], db)
        dw = _hip.linear_wgrad(x, dz, accumulate_into=_hip.sink_of(ctx.params[0]))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _hip.rows_dgrad(dz, w)
        return dx, dw, db, None, None, None, None


class FCNMaxFunction(torch.autograd.Function):
    """rows (V*T or compact, K) -> max_t BN(ReLU(fc)) (V, N): VoxelNet.py:28-33 as one node."""

    @staticmethod
    def forward(ctx, x, w, b, V, T, eps, cr):
        row_w = cr.row_w if cr is not None else None
        y, mi = _hip.linear_forward(x, w, b, relu=True, want_stats=True, row_w=row_w, finalize=(V * T, eps),
                                    split=_hip.row_split('vfe'))
        out, am = _hip.bn_segment_max(y, mi, V, T, cr)
        ctx.save_for_backward(x, w, y, mi, am)
        ctx.vt = (V, T, cr)
        ctx.params = (w, b)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, y, mi, am = ctx.saved_tensors
        V, T, cr = ctx.vt
        dyh = _hip.segment_max_backward(g.contiguous(), am, V, T, cr)
        dz, db = _hip.bn_relu_backward(dyh, y, mi, V * T, True, dz=dyh, row_w=cr.row_w if cr is not None else None,
                                       dbias_out=_hip.bias_sink_of(ctx.params[1]))
        db = _hip.accumulate_grad(ctx.params[1], db)
        dw = _hip.linear_wgrad(x, dz, accumulate_into=_hip.sink_of(ctx.params[0]))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _hip.rows_dgrad(dz, w)
        return dx, dw, db, None, None, None, None


class CompactInputFunction(torch.autograd.Function):
    """VFE-1 input in compact form from the voxel rows and the compact fusion output
    (the concat of MVXNet.py:26, without materialising the dense (1,N,T,23) tensor)."""

    @staticmethod
    def forward(ctx, imfeat_c, vox2d, cr):
        ctx.cr = cr
        ctx.F = imfeat_c.shape[1]
        return _hip.vfe_compact_input(vox2d, imfeat_c.contiguous(), cr)

    @staticmethod
    def backward(ctx, g):
        return _hip.vfe_compact_input_backward(g.contiguous(), ctx.F, ctx.cr), None, None


class VFE(nn.Module):
    """(batch=1, N, T, cin) -> (1, N, T, 2*cout) (reference Pipe.py:5-18)."""

    def __init__(self, cin, cout, sampleNum):
        super().__init__()
        self.fcn = FCN(cin, cout)
        self.sampleNum = sampleNum

    def forward(self, x):
        b, n, t, _ = x.shape
        out = VFEFunction.apply(_as_rows(x), self.fcn.fc.weight, self.fcn.fc.bias, b * n, t, cfg.eps, None)
        return out.reshape(b, n, t, out.shape[-1])

    def forward_compact(self, rows, cr):
        """Compact rows (n_real + V, cin) -> (n_real + V, 2*cout); exact inside MVXNet."""
        return VFEFunction.apply(rows, self.fcn.fc.weight, self.fcn.fc.bias, cr.V, cr.T, cfg.eps, cr)


class SVFE(nn.Module):
    """VFE(23->16) then VFE(32->64): (1, N, T, 23) -> (1, N, T, 128) (reference Pipe.py:20-29)."""

    def __init__(self, sampleNum=35):
        super().__init__()
        self.vfe1 = VFE(7 + 16, 16, sampleNum)
        self.vfe2 = VFE(32, 64, sampleNum)

    def forward(self, x):
        return self.vfe2(self.vfe1(x))

    def forward_compact(self, rows, cr):
        return self.vfe2.forward_compact(self.vfe1.forward_compact(rows, cr), cr)


class CML(nn.Module):
    """Three Conv3d-ReLU-BN blocks (reference Pipe.py:31-43)."""

    def __init__(self):
        super().__init__()
        self.conv1 = CRB3d(128, 64, 3, (2, 1, 1), (1, 1, 1))
        self.conv2 = CRB3d(64, 64, 3, 1, (0, 1, 1))
        self.conv3 = CRB3d(64, 64, 3, (2, 1, 1), 1)

    def forward(self, x):
        return self.conv3(self.conv2(self.conv1(x)))


class RPNFunction(torch.autograd.Function):
    """The whole RPN as ONE autograd node on this library's kernels (modules/rpn_frames.py, a frame set of one frame):
    channels-last planes [planes][H][W][Cp] (BEV channel c*planes + d) -> heads (H/2*W/2, 16) = [cls logits | reg].
    Backward: rpn_frames.rpn_backward; the parameter gradients go into the existing .grad buffers when the training
    pipeline opened them (_hip.GRAD_SINK), otherwise they are returned to the autograd engine."""

    @staticmethod
    def forward(ctx, x_cl, rpn, planes, H, W, Cp, *params):
        from modules import rpn_frames as rf
        heads, S = rf.rpn_forward(rpn, x_cl.contiguous(), 1, planes, H, W, Cp)
        ctx.rpn, ctx.S, ctx.n_params = rpn, S, len(params)
        ctx.params = params
        return heads

    @staticmethod
    def backward(ctx, g):
        from modules import rpn_frames as rf
        params = ctx.params
        direct = _hip.GRAD_SINK and all(p.grad is not None and p.grad.is_contiguous() for p in params)
        if direct:
            gx = rf.rpn_backward(ctx.rpn, ctx.S, g)
            grads = (None,) * len(params)
        else:
            flat = torch.zeros((sum(p.numel() for p in params),), dtype=torch.float32, device=g.device)
            views, off = [], 0
            for p in params:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            with rf.grad_targets({id(p): v for p, v in zip(params, views)}):
                gx = rf.rpn_backward(ctx.rpn, ctx.S, g)
            grads = tuple(views)
        if not _hip.ASYNC_WGRAD:
            _hip.join_side_stream(g.device)        # the weight gradients were produced on the side stream
        ctx.S = None
        return (gx if ctx.needs_input_grad[0] else None, None, None, None, None, None) + grads


class RPN(nn.Module):
    """Region proposal network (reference Pipe.py:45-75), same topology and state-dict keys.  On the GPU the forward and
    backward run on this library's kernels as one autograd node (RPNFunction over modules/rpn_frames.py: the MFMA gather /
    weight-gradient kernels for the 3x3 layers, row GEMMs for the kernel = stride deconvolutions and the heads; per-frame
    BatchNorm statistics); ``rpn_hip: False`` in config.yml (or a CPU tensor, or batch > 1) takes the torch modules
    (CRB2d / DeCRB2d -> MIOpen), which remain as the comparison path of the tests."""

    def __init__(self):
        super().__init__()
        self.blk1 = nn.Sequential(CRB2d(128, 128, 3, 2, 1), *[CRB2d(128, 128, 3, 1, 1) for _ in range(3)])
        self.blk2 = nn.Sequential(CRB2d(128, 128, 3, 2, 1), *[CRB2d(128, 128, 3, 1, 1) for _ in range(5)])
        self.blk3 = nn.Sequential(CRB2d(128, 256, 3, 2, 1), *[CRB2d(256, 256, 3, 1, 1) for _ in range(5)])
        self.deconv1 = DeCRB2d(128, 256, 3, 1, 1)
        self.deconv2 = DeCRB2d(128, 256, 2, 2, 0)
        self.deconv3 = DeCRB2d(256, 256, 4, 4, 0)
        self.cls = nn.Conv2d(768, 2, 1, 1, 0)
        self.reg = nn.Conv2d(768, 14, 1, 1, 0)

    def _hip_ok(self, x, H, W):
        return bool(cfg.config.get('rpn_hip', True)) and x.is_cuda and x.dtype == torch.float32 and H % 8 == 0 and W % 8 == 0

    def forward_cl(self, x_cl, planes, H, W, Cp):
        """The RPN on the channels-last planes of the CML output [planes][H][W][Cp] (BEV channel c*planes + d, the reshape
        of VoxelNet.py:36 folded into the first layer's weight): (score (1,2,H/2,W/2), reg (1,14,H/2,W/2))."""
        params = [p for p in self.parameters()]
        heads = RPNFunction.apply(x_cl, self, planes, H, W, Cp, *params)
        v = heads.view(1, H // 2, W // 2, 16)
        return torch.sigmoid(v[..., :2]).permute(0, 3, 1, 2), v[..., 2:].permute(0, 3, 1, 2)

    def forward_torch(self, x):
        """The per-module path: every block through its own forward.  On the torch modules (MIOpen) this is the comparison
        path of the tests; config ``crb2d_hip: force`` sends the blocks through their stand-alone HIP nodes instead."""
        from modules.layers import Blocks
        old, Blocks._IN_FORWARD_TORCH[0] = Blocks._IN_FORWARD_TORCH[0], True
        try:
            x1 = self.blk1(x)
            x2 = self.blk2(x1)
            x3 = self.blk3(x2)
            up = torch.concat([self.deconv1(x1), self.deconv2(x2), self.deconv3(x3)], dim=1)
            return torch.sigmoid(self.cls(up)), self.reg(up)
        finally:
            Blocks._IN_FORWARD_TORCH[0] = old

    def forward(self, x):
        if x.dim() == 4 and x.shape[0] == 1 and self._hip_ok(x, x.shape[2], x.shape[3]):
            H, W = x.shape[2], x.shape[3]
            return self.forward_cl(x[0].permute(1, 2, 0), 1, H, W, x.shape[1])      # one plane of C channels per site
        return self.forward_torch(x)
