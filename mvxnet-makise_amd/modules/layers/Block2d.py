"""Stand-alone 2-D blocks on this library's kernels: one autograd node per CRB2d / DeCRB2d call.

The reference's CRB2d (Conv2d -> ReLU -> BatchNorm2d, modules/layers/Blocks.py:31-40) and DeCRB2d (ConvTranspose2d -> ReLU ->
BatchNorm2d, Blocks.py:42-51) are used by its RPN only (voxelnet/Pipe.py:45-75), where this package runs the whole RPN as ONE
node (voxelnet/Pipe.py RPNFunction over modules/rpn_frames.py).  Code that calls the blocks DIRECTLY gets the same kernels
through this module, with the machinery of rpn_frames for a frame set of one frame:

  kind 's1'  3x3, stride 1, padding 1            MFMA gather / weight-gradient kernels (csrc/conv3d.hip, conv3d_split.hip)
  kind 's2'  3x3, stride 2, padding 1            space-to-depth image + 2x2 tap window (MVX_FLAG_TAPS2), structural zeros skipped
  kind 'd1'  ConvTranspose2d 3x3, stride 1, p 1  = convolution with the flipped kernel, channel axes swapped
  kind 'dk'  ConvTranspose2d kernel = stride     row GEMM [sites x Cin] . [Cin x s*s*Cout] + normalising pixel shuffle

Channel counts must suit the 64-wide MFMA tiles (see ``kind_of``); anything else, CPU tensors and batch > 1 stay on the torch
modules (MIOpen), which are also what ``RPN.forward_torch`` -- the comparator of the tests -- runs."""
import torch

import modules.config as cfg
from modules import _hip
from modules import Extension as X
from modules import rpn_frames as rf

R = _hip.STATS_REPLICAS


def kind_of(module_kind, k, s, p, cin, cout):
    """Which kernel family runs a block with these hyper-parameters, or None (torch)."""
    if module_kind == 'conv':
        if k == 3 and p == 1 and s == 1 and cin % 64 == 0 and cout % 64 == 0:
            return 's1'
        if k == 3 and p == 1 and s == 2 and cin % 16 == 0 and cout % 64 == 0:      # 4 * cin a multiple of 64
            return 's2'
        return None
    if k == 3 and s == 1 and p == 1 and cin % 64 == 0 and cout % 64 == 0:
        return 'd1'
    if k == s and p == 0 and s in (2, 4) and cin % 4 == 0 and cout % 4 == 0:
        return 'dk'
    return None


class _Pack:
    """Kernel-layout copy of one block's (possibly rearranged) weight per (arithmetic, direction), refreshed when it changes."""

    def __init__(self):
        self.packs = rf._Packs()

    def get(self, key, w, make, for_dgrad):
        return self.packs.get(key, w, make, for_dgrad)


class Block2dFunction(torch.autograd.Function):
    """x (1, Cin, H, W) in any memory format -> BN(ReLU(conv(x))) as logical NCHW over channels-last storage."""

    @staticmethod
    def forward(ctx, x, w, b, kind, stride, pack):
        # a stand-alone block reads a tensor of unknown range: under convmath fp16x3 it runs in bf16x6 (_hip.foreign_input_math)
        with _hip.foreign_input_math(x) as math:
            ctx.math = math
            return Block2dFunction._forward(ctx, x, w, b, kind, stride, pack)

    @staticmethod
    def backward(ctx, g):
        with _hip.foreign_input_math(math=ctx.math):
            return Block2dFunction._backward(ctx, g)

    @staticmethod
    def _forward(ctx, x, w, b, kind, stride, pack):
        _hip.require_plain_batchnorm()
        eps = cfg.eps
        xc = x[0].permute(1, 2, 0).contiguous()[None]            # (1,H,W,C): no copy for channels_last input
        _, h, wd, cin = xc.shape
        S = {'kind': kind, 'x': xc}
        if kind == 's1':
            cout = w.shape[0]
            wpk = pack.get(('s1',), w, lambda: w, False)
            y, mi = rf._conv(xc, wpk, b, 1, h, wd, cin, cout, 0, eps)
            out = rf._bn_apply(y, mi, 1)
        elif kind == 's2':
            cout = w.shape[0]
            xs = rf._s2d(xc, 1, 1, h, wd, cin)
            wpk = pack.get(('s2',), w, lambda: rf._s2d_weight(w, 1), False)
            y, mi = rf._conv(xs, wpk, b, 1, h // 2, wd // 2, 4 * cin, cout, rf.TAPS2, eps)
            out = rf._bn_apply(y, mi, 1)
            S['xs'] = xs
        elif kind == 'd1':
            cout = w.shape[1]
            wpk = pack.get(('d1',), w, lambda: w.flip(2, 3).transpose(0, 1).contiguous(), False)
            y, mi = rf._conv(xc, wpk, b, 1, h, wd, cin, cout, 0, eps)
            out = rf._bn_apply(y, mi, 1)
        else:                                                     # 'dk': kernel = stride
            s = stride
            cout = w.shape[1]
            w_all = w.permute(2, 3, 1, 0).reshape(s * s * cout, cin).contiguous()      # row (i*s+j)*Cout + co
            xr = xc.view(h * wd, cin)
            t, _ = _hip.linear_forward(xr, w_all, b.repeat(s * s), relu=True, want_stats=False, split=_hip.row_split('rpn'),
                                       foreign=not getattr(x, '_mvx_lib', False))
            stats = torch.empty((1, R, 2, cout), dtype=torch.float64, device=x.device)
            X.check(X.lib.mvx_row_stats_frames(X.ptr(t), X.ptr(stats), t.numel() // cout, cout, 1, X.stream()), 'mvx_row_stats_frames')
            mi = torch.empty((1, 2, cout), dtype=torch.float32, device=x.device)
            X.check(X.lib.mvx_bn_finalize_frames(X.ptr(stats), float(h * wd * s * s), float(eps), X.ptr(mi), cout, 1, X.stream()),
                    'mvx_bn_finalize_frames')
            out = torch.empty((1, h * s, wd * s, cout), dtype=torch.float32, device=x.device)
            X.check(X.lib.mvx_d2s_bn_apply_frames(X.ptr(t), X.ptr(mi), X.ptr(out), 1, h, wd, s, cout, cout, 0, 0, X.stream()),
                    'mvx_d2s_bn_apply_frames')
            y = t
            S.update(xr=xr, w_all=w_all, s=s)
        S.update(y=y, mi=mi, h=h, w=wd, cin=cin, cout=cout)
        ctx.S, ctx.pack = S, pack
        ctx.params = (w, b)
        return _hip.mark_lib(out[0].permute(2, 0, 1).unsqueeze(0))   # logical NCHW, channels_last storage

    @staticmethod
    def _backward(ctx, g):
        S, pack = ctx.S, ctx.pack
        w, b = ctx.params
        kind, h, wd, cin, cout = S['kind'], S['h'], S['w'], S['cin'], S['cout']
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        gc = g[0].permute(1, 2, 0).contiguous()[None]            # (1,H',W',Cout)
        need_dx = ctx.needs_input_grad[0]
        dx = None
        with torch.no_grad(), rf.grad_targets({id(w): dw, id(b): db}):
            if kind in ('s1', 'd1'):
                dz = rf._bn_bwd(gc, S['y'], S['mi'], 1, b)
                if kind == 's1':
                    rf._wgrad(S['x'], dz, 1, h, wd, cin, cout, 0, into=dw)
                    if need_dx:
                        dx = rf._dgrad(dz, pack.get(('s1',), w, lambda: w, True), 1, h, wd, cin, cout, 0)
                else:
                    dwc = rf._wgrad(S['x'], dz, 1, h, wd, cin, cout, 0)          # gradient of the flipped / transposed kernel
                    with _hip._SideStream(dwc, dw):
                        dw.add_(dwc.transpose(0, 1).flip(2, 3))
                    if need_dx:
                        wcd = pack.get(('d1',), w, lambda: w.flip(2, 3).transpose(0, 1).contiguous(), True)
                        dx = rf._dgrad(dz, wcd, 1, h, wd, cin, cout, 0)
            elif kind == 's2':
                h2, w2, c4 = h // 2, wd // 2, 4 * cin
                dz = rf._bn_bwd(gc, S['y'], S['mi'], 1, b)
                dw2 = rf._wgrad(S['xs'], dz, 1, h2, w2, c4, cout, rf.TAPS2)
                with _hip._SideStream(dw2, dw):
                    rf._s2d_weight_grad(dw2, w, 1)
                if need_dx:
                    wpd = pack.get(('s2',), w, lambda: rf._s2d_weight(w, 1), True)
                    gs = rf._dgrad(dz, wpd, 1, h2, w2, c4, cout, rf.TAPS2)
                    dx = rf._d2s(gs, 1, 1, h, wd, cin).view(1, h, wd, cin)
            else:
                s = S['s']
                gt = torch.empty_like(S['y'])
                X.check(X.lib.mvx_d2s_bn_apply_frames(X.ptr(gt), None, X.ptr(gc), 1, h, wd, s, cout, cout, 0, 1, X.stream()),
                        'mvx_d2s_bn_apply_frames')
                dz0 = rf._bn_bwd(gt.view(-1, cout), S['y'].view(-1, cout), S['mi'], 1, b)
                dz = rf._retag(dz0.view(S['y'].shape), dz0)
                dw_all = _hip.linear_wgrad(S['xr'], dz)                           # (s*s*cout, cin)
                dw.add_(dw_all.view(s, s, cout, cin).permute(3, 2, 0, 1))
                if need_dx:
                    gx, _ = _hip.linear_forward(dz, S['w_all'].t().contiguous(), None, relu=False, want_stats=False,
                                                label='linear_dgrad', split=_hip.row_split('dgrad'))
                    dx = gx.view(1, h, wd, cin)
            _hip.join_side_stream(g.device)                      # the weight gradients were produced on the side stream
        ctx.S = None
        return (dx[0].permute(2, 0, 1).unsqueeze(0) if dx is not None else None), dw, db, None, None, None
