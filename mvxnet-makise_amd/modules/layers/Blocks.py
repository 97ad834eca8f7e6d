"""Building blocks with the reference's names and sub-module attributes (``fc`` / ``conv`` /
``deconv`` / ``bn``), computing on hand-written HIP kernels.

Reference: modules/layers/Blocks.py -- every block is  affine -> ReLU -> BatchNorm  with
batch statistics in train and eval, biased variance, eps = cfg.eps, no affine, no buffers
(Blocks.py:10,16; config.yml:19-20).  State-dict keys are unchanged (``fc.weight`` ...), so
reference checkpoints load.
"""
import torch
from torch import nn
from torch.nn import functional as F

import modules.config as cfg
from modules import _hip


def _as_rows(x):
    """(..., C) -> contiguous 2-D (R, C) view."""
    c = x.shape[-1]
    return x.reshape(-1, c) if x.is_contiguous() else x.contiguous().reshape(-1, c)


class FCNFunction(torch.autograd.Function):
    """rows -> BN(ReLU(rows W^T + b)).  ``row_w``: optional multiplicity of each row (a compact
    row standing for several identical rows of the reference's dense tensor)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, row_w, count, foreign=True):
        # ``foreign``: x is not the output of one of this library's BatchNorms (its range is unknown): _hip.foreign_split
        w2 = w.reshape(w.shape[0], -1)
        sp = _hip.row_split('fusion_%dx%d' % tuple(w2.shape))
        y, mi = _hip.linear_forward(x, w2, b, relu=True, want_stats=True, row_w=row_w, finalize=(count, eps), split=sp,
                                    foreign=foreign)
        out = _hip.bn_apply(y, mi)
        ctx.save_for_backward(x, w, y, mi, row_w)
        ctx.count = count
        ctx.foreign = foreign
        ctx.params = (w, b)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, y, mi, row_w = ctx.saved_tensors
        w2 = w.reshape(w.shape[0], -1)
        sw, sb = _hip.sink_of(ctx.params[0]), _hip.bias_sink_of(ctx.params[1])
        dz, db = _hip.bn_relu_backward(g.contiguous(), y, mi, ctx.count, True, row_w=row_w, dbias_out=sb)
        db = _hip.accumulate_grad(ctx.params[1], db)
        dw = _hip.linear_wgrad(x, dz, accumulate_into=sw, x_foreign=ctx.foreign)
        dw = dw.reshape(w.shape) if dw is not None else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _hip.rows_dgrad(dz, w2)
        return dx, dw, db, None, None, None, None


def fcn_rows(x2d, weight, bias, row_w=None, count=None, foreign=True):
    count = x2d.shape[0] if count is None else count
    return FCNFunction.apply(x2d, weight, bias, cfg.eps, row_w, float(count), foreign)


class FCN(nn.Module):
    """Linear -> ReLU -> BN over (batch, h, w) per channel (reference Blocks.py:5-18)."""

    def __init__(self, cin, cout):
        super().__init__()
        _hip.require_plain_batchnorm()
        self.fc = nn.Linear(cin, cout)
        self.bn = nn.BatchNorm2d(cout, eps=cfg.eps, affine=cfg.bnaffine, track_running_stats=cfg.bntrack)

    def forward(self, x):
        # input (batch, h, w, c) -> output (batch, h, w, cout)
        out = fcn_rows(_as_rows(x), self.fc.weight, self.fc.bias)
        return out.reshape(x.shape[:-1] + (out.shape[-1],))


def conv_split_math():
    """Number of bf16 pieces per operand of the dense convolutions: 0 -> exact-f32 MFMA (csrc/conv3d.hip); 2 -> hi/lo split
    ("bf16x3": three bf16 MFMAs per product, ~2e-5 per product); 3 -> hi/mid/lo ("bf16x6": six MFMAs, fp32-grade)
    (csrc/conv3d_split.hip).  Selected by ``convmath`` in config.yml / cfg.config (default f32)."""
    return _hip.split_pieces()


class PackedWeights:
    """Kernel-layout copies of a module's conv weight, refreshed when the parameter changes (in-place
    optimizer update -> ``_version``; ``.to()`` / reload -> ``data_ptr``).  The frames of a step share
    one packing.  Owned by the module, so the parameter outlives every cached copy."""

    def __init__(self, get_weight):
        self._get = get_weight
        self._cache = {}

    def __call__(self, for_dgrad, split):
        w = self._get()
        tag = (w._version, w.data_ptr())
        hit = self._cache.get((for_dgrad, split))
        if hit is None or hit[0] != tag:
            hit = (tag, _hip.conv3d_pack(w.detach(), for_dgrad, split=split))
            self._cache[(for_dgrad, split)] = hit
        return hit[1]


def _pack(packer, w, for_dgrad, split):
    return packer(for_dgrad, split) if packer is not None else _hip.conv3d_pack(w, for_dgrad, split=split)


RESTRICT_LAYER2 = True      # also restrict the backward of a layer whose input is itself restricted (conv2 in CML)
# The restricted backward hands the producer an input gradient that is only valid on part of the grid (plus closed-form
# sums): correct only if the producer's output has NO other consumer.  modules/tape.py knows the chain and switches it
# on; through the nn.Module / autograd interface it stays off (dense input gradients), so a model that reuses a CML
# activation elsewhere still gets correct gradients.
RESTRICTED_BACKWARD = False


def conv_background_on():
    """config.yml ``convbackground``: skip the voxel-free background of the CML activations (exact rewrite, csrc/activity.hip)."""
    return bool(cfg.config.get('convbackground', True))


class CRB3dFunction(torch.autograd.Function):
    """channels-last (D,H,W,Cin) -> BN(ReLU(conv3d)) (Dout,H,W,Cout), one frame.

    ``bg_in`` (a _hip.Background of x, or None) switches the background rewrite on: forward fills voxel-free
    tiles with the per-plane constant, wgrad sums over the other tiles and adds the constant's share in
    closed form; ``aux['bg']`` receives the Background of the result for the next block."""

    @staticmethod
    def forward(ctx, x, w, b, sd, pd, eps, packer=None, bg_in=None, aux=None):
        cout = w.shape[0]
        split = conv_split_math()
        wpk = _pack(packer, w, False, split)
        ctx.packer = packer
        if bg_in is not None:
            din, H, W, _ = x.shape
            bg_pre = _hip.conv3d_background(w, bg_in.c, din, sd, pd)
            out_mask, out_hflag, out_tflag = _hip.activity_dilate(bg_in.mask, False, din, H, W, sd, pd, mark_border=True,
                                                                  want_tile_flags=True)
            y, mi = _hip.conv3d_forward_bg(x, wpk, b, cout, sd, pd, bg_in, out_mask, bg_pre, finalize_eps=eps, split=split)
            count = y.numel() // cout
        else:
            y, stats = _hip.conv3d_forward(x, wpk, b, cout, sd, pd, relu=True, want_stats=True, split=split)
            count = y.numel() // cout
            mi = _hip.bn_finalize(stats, count, eps)
        out = _hip.bn_apply(y, mi)
        ctx.bg_out = None
        if bg_in is not None and aux is not None:
            c_out, y_out = _hip.bn_background(bg_pre, b, mi, y.shape[0], cout, want_y=True)
            restricted = RESTRICTED_BACKWARD and RESTRICT_LAYER2 and bg_in.back is not None and bg_in.tflag is not None
            # tiles of THIS layer's output gradient that its own restricted backward touches: the halos of the input
            # tiles its dgrad is restricted to (and of its wgrad steps), plus every tile holding a non-background site
            bflag = _hip.tile_dilate_flags(bg_in.tflag, out_tflag, x.shape[0], x.shape[1], x.shape[2], sd, pd) if restricted else None
            aux['bg'] = _hip.Background(c_out, out_mask, out_hflag, tflag=out_tflag, y_bg=y_out,
                                        back={} if restricted else None, bflag=bflag)
            ctx.bg_out = aux['bg']
        ctx.save_for_backward(x, w, y, mi)
        ctx.geom = (sd, pd, count, split)
        ctx.params = (w, b)
        ctx.bg_in = bg_in
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, y, mi = ctx.saved_tensors
        sd, pd, count, split = ctx.geom
        bg_in, bg_out = ctx.bg_in, ctx.bg_out
        tap_sums = None
        if bg_out is not None and bg_out.back and 'plane_grad_sums' in bg_out.back:
            # the consumer produced g on bg_out.bflag tiles only and left the per-plane sums of the rest (closed form)
            dz, db, inact = _hip.bn_relu_backward_tiles(g.contiguous(), y, mi, bg_out, bg_out.back.pop('plane_grad_sums'),
                                                        dbias_out=_hip.bias_sink_of(ctx.params[1]), want_inactive_sums=True)
            tap_sums = _hip.plane_tap_sums(dz, bg_out.bflag, inact)
        else:
            dz, db = _hip.bn_relu_backward(g.contiguous(), y, mi, count, True, dbias_out=_hip.bias_sink_of(ctx.params[1]))
        db = _hip.accumulate_grad(ctx.params[1], db)
        if bg_in is not None and tap_sums is None:
            tap_sums = _hip.plane_tap_sums(dz)
        if bg_in is not None:
            dw = _hip.conv3d_wgrad_bg(x, dz, sd, pd, bg_in, tap_sums, accumulate_into=_hip.sink_of(ctx.params[0]), split=split)
        else:
            dw = _hip.conv3d_wgrad(x, dz, sd, pd, split=split, accumulate_into=_hip.sink_of(ctx.params[0]))
        dx = None
        if ctx.needs_input_grad[0]:
            wpd = _pack(ctx.packer, w, True, split)
            if bg_in is not None and bg_in.back is not None and bg_in.bflag is not None:
                # the producer only needs the gradient on its bflag tiles plus per-plane sums (closed form)
                dx = _hip.conv3d_dgrad_tiles(dz, wpd, x.shape[0], x.shape[3], sd, pd, bg_in.bflag, split=split)
                bg_in.back['plane_grad_sums'] = _hip.conv3d_input_grad_sums(w, tap_sums, x.shape[0], sd, pd)
            else:
                dx = _hip.conv3d_dgrad(dz, wpd, x.shape[0], x.shape[3], sd, pd, split=split)
        return dx, dw, db, None, None, None, None, None, None


class VoxelGemmCRB3dFunction(torch.autograd.Function):
    """reindex + first CRB3d through the [V x 27*Cout] factorisation (csrc/sparseconv.hip): one row
    GEMM + one index-grid gather per pass, no dense input grid.  Equal to the dense evaluation up to
    fp32 summation order."""

    @staticmethod
    def forward(ctx, feat, coords, w, b, dhw, sd, pd, eps, aux=None):
        cout, cin = w.shape[0], w.shape[1]
        feat = feat.contiguous()
        w_all = w.permute(2, 3, 4, 0, 1).reshape(27 * cout, cin).contiguous()
        P, _ = _hip.linear_forward(feat, w_all, None, relu=False, want_stats=False, split=_hip.row_split('conv1'))
        idx_grid, status = _hip.index_grid(coords, dhw)
        y, stats = _hip.sparse_conv_output(P, idx_grid, dhw, b, cout, sd, pd, relu=True, want_stats=True)
        count = y.numel() // cout
        mi = _hip.bn_finalize(stats, count, eps)
        out = _hip.bn_apply(y, mi)
        if aux is not None:
            # voxel-free sites hold ReLU(bias) exactly, also at the image border (zero grid, zero padding)
            mask, hflag, tflag = _hip.activity_dilate(idx_grid, True, dhw[0], dhw[1], dhw[2], sd, pd, mark_border=False,
                                                      want_tile_flags=True)
            c1, y1 = _hip.bn_background(None, b, mi, y.shape[0], cout, want_y=True)
            aux['bg'] = _hip.Background(c1, mask, hflag, tflag=tflag, y_bg=y1, back={} if RESTRICTED_BACKWARD else None)
        ctx.bg = aux['bg'] if aux is not None else None
        ctx.save_for_backward(feat, coords, w_all, y, mi)
        ctx.geom = (dhw[0], sd, pd, count, tuple(w.shape))
        ctx.params = (w, b)
        return out

    @staticmethod
    def backward(ctx, g):
        feat, coords, w_all, y, mi = ctx.saved_tensors
        din, sd, pd, count, wshape = ctx.geom
        cout, cin = wshape[0], wshape[1]
        bg = ctx.bg
        if bg is not None and bg.back and 'plane_grad_sums' in bg.back:
            # g is only valid next to the voxels (conv3d_dgrad_tiles); dz is only read there (gather below)
            dz, db = _hip.bn_relu_backward_tiles(g.contiguous(), y, mi, bg, bg.back.pop('plane_grad_sums'),
                                                 dbias_out=_hip.bias_sink_of(ctx.params[1]))
        else:
            dz, db = _hip.bn_relu_backward(g.contiguous(), y, mi, count, True, dbias_out=_hip.bias_sink_of(ctx.params[1]))
        db = _hip.accumulate_grad(ctx.params[1], db)
        G = _hip.sparse_conv_gather_dz(dz, coords, din, sd, pd)
        dw_all = _hip.linear_wgrad(feat, G)                                   # (27*cout, cin)
        dw = _hip.accumulate_grad(ctx.params[0], dw_all.reshape(3, 3, 3, cout, cin).permute(3, 4, 0, 1, 2))
        dfeat = None
        if ctx.needs_input_grad[0]:
            dfeat = _hip.rows_dgrad(G, w_all)
        return dfeat, None, dw, db, None, None, None, None, None


def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


class CRB3d(nn.Module):
    """Conv3d -> ReLU -> BN3d (reference Blocks.py:20-29) on the MFMA implicit-GEMM kernel.

    Supports what CML uses: kernel 3, stride (s,1,1), padding (p,1,1), batch 1.  The input may be
    logical NCDHW in any memory format; internally the data is channels-last and the result is
    returned as a logical NCDHW view of channels-last storage (no copy)."""

    def __init__(self, cin, cout, k, s, p):
        super().__init__()
        _hip.require_plain_batchnorm()
        self.conv = nn.Conv3d(cin, cout, k, s, p)
        self.bn = nn.BatchNorm3d(cout, eps=cfg.eps, affine=cfg.bnaffine, track_running_stats=cfg.bntrack)
        k3, s3, p3 = _triple(k), _triple(s), _triple(p)
        if k3 != (3, 3, 3) or s3[1:] != (1, 1) or p3[1:] != (1, 1) or s3[0] not in (1, 2) or p3[0] not in (0, 1):
            raise NotImplementedError('CRB3d HIP kernel: kernel 3, stride (s,1,1), padding (p,1,1) only')
        self._sd, self._pd = s3[0], p3[0]
        object.__setattr__(self, '_packer', PackedWeights(lambda: self.conv.weight))

    def forward_voxels(self, feat, coords, dhw):
        """Fused reindex + this block on sparse voxel rows: the [V x 27*Cout] factorisation (exact, see
        VoxelGemmCRB3dFunction); the dense grid is never built."""
        aux = {} if conv_background_on() else None
        out = VoxelGemmCRB3dFunction.apply(feat, coords, self.conv.weight, self.conv.bias, tuple(dhw), self._sd,
                                           self._pd, cfg.eps, aux)
        res = _hip.mark_lib(out.permute(3, 0, 1, 2).unsqueeze(0))
        if aux:
            res._mvx_background = aux['bg']      # read by the next CRB3d (plain attribute: the module API is unchanged)
        return res

    def forward(self, x):
        if x.shape[0] != 1:
            raise NotImplementedError('batch size 1 only (reference VoxelNet.py:19)')
        # squeeze, not x[0]: the backward of a select allocates zeros and copies the whole gradient
        xc = x.squeeze(0).permute(1, 2, 3, 0).contiguous()  # no-op when already channels-last
        bg_in = getattr(x, '_mvx_background', None) if conv_background_on() else None
        aux = {} if bg_in is not None else None
        with _hip.foreign_input_math(x):         # fp16x3 on a tensor of unknown range: this call runs in bf16x6
            out = CRB3dFunction.apply(xc, self.conv.weight, self.conv.bias, self._sd, self._pd, cfg.eps, self._packer, bg_in, aux)
        res = _hip.mark_lib(out.permute(3, 0, 1, 2).unsqueeze(0))
        if aux and 'bg' in aux:
            res._mvx_background = aux['bg']
        return res


# RPN.forward_torch -- the torch / MIOpen comparison path of the tests -- sets this while it runs: the 2-D blocks then stay on
# the torch modules unless config `crb2d_hip` is 'force' (tests of the per-block HIP path through that entry)
_IN_FORWARD_TORCH = [False]


def _block2d_on(x, kind):
    """Does a stand-alone CRB2d / DeCRB2d call with this input run on the HIP kernels (modules/layers/Block2d.py)?  Yes by
    default (config `crb2d_hip: true`) for float32 CUDA tensors of batch 1 whose hyper-parameters the kernels cover."""
    mode = cfg.config.get('crb2d_hip', True)
    if _IN_FORWARD_TORCH[0] and mode != 'force':
        return False
    if not (mode and kind is not None and x.is_cuda and x.dim() == 4 and x.shape[0] == 1 and x.dtype == torch.float32):
        return False
    return kind != 's2' or (x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)


def _nchw(x):
    """Hand a channels_last tensor back to stock PyTorch-ROCm in its fast NCHW layout."""
    return x.contiguous(memory_format=torch.contiguous_format)


_WARNED = set()


def _torch_dispatch(x, name, kind):
    """A CRB2d / DeCRB2d call is about to run on the torch modules (MIOpen).  That is never silent: inside RPN.forward_torch (the
    comparison path of the tests, chosen by the caller) and with `crb2d_hip: false` (chosen in config.yml) it is what was asked
    for; a CPU tensor raises like every other module of this package (there is no CPU path); any other reason -- a batch, a
    dtype, a kernel / channel count the MFMA tiles do not cover, an odd map under a stride-2 layer -- is named ONCE in a
    RuntimeWarning."""
    if _IN_FORWARD_TORCH[0] or not cfg.config.get('crb2d_hip', True):
        return
    if not x.is_cuda:
        raise _hip.X.MvxHipError('%s: CPU tensor -- this package has no CPU path (the HIP library runs on the GPU only)' % name)
    why = ('hyper-parameters outside the HIP kernels (3x3 / stride 1 or 2, kernel = stride deconvolutions, channels in multiples of 32)'
           if kind is None else 'batch size %d' % x.shape[0] if x.dim() == 4 and x.shape[0] != 1 else
           'dtype %s' % x.dtype if x.dtype != torch.float32 else 'map size %s under a stride-2 layer' % (tuple(x.shape[-2:]),))
    if (name, why) not in _WARNED:
        _WARNED.add((name, why))
        import warnings
        warnings.warn('%s runs on torch / MIOpen, not on the HIP kernels: %s' % (name, why), RuntimeWarning, stacklevel=3)


class _TorchBN2d(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = c

    def forward(self, x):
        return F.batch_norm(x, None, None, None, None, True, 0.0, cfg.eps)


class CRB2d(nn.Module):
    """Conv2d -> ReLU -> BN2d (reference Blocks.py:31-40).  1x1 kernels (the fusion MLP,
    imhead/Pipe.py:89,91) run on the HIP row-GEMM.  3x3 kernels belong to the RPN, whose forward runs as one node on the HIP
    kernels (voxelnet/Pipe.py RPNFunction); called on its own, a 3x3 block (stride 1 or 2) is ONE autograd node on the same
    kernels (modules/layers/Block2d.py; config ``crb2d_hip``, default true).  The torch / MIOpen form remains as the comparison
    path of the tests (RPN.forward_torch) and, announced by a RuntimeWarning (_torch_dispatch), for batches and channel counts
    the MFMA tiles do not cover; a CPU tensor raises."""

    def __init__(self, cin, cout, k, s, p):
        super().__init__()
        _hip.require_plain_batchnorm()
        self.conv = nn.Conv2d(cin, cout, k, s, p)
        self.bn = nn.BatchNorm2d(cout, eps=cfg.eps, affine=cfg.bnaffine, track_running_stats=cfg.bntrack)
        self._pointwise = (k == 1 and s == 1 and p == 0)
        from modules.layers import Block2d
        self._kind = Block2d.kind_of('conv', k, s, p, cin, cout)
        self._stride = s
        object.__setattr__(self, '_pack2d', Block2d._Pack())

    def forward(self, x):
        if self._pointwise and x.is_cuda:
            # (b, c, h, w) -> rows (b*h*w, c); a permuted channels-last input costs no copy
            rows = x.permute(0, 2, 3, 1)
            out = fcn_rows(_as_rows(rows), self.conv.weight, self.conv.bias)
            return out.reshape(rows.shape[:-1] + (out.shape[-1],)).permute(0, 3, 1, 2)
        if _block2d_on(x, self._kind):
            from modules.layers.Block2d import Block2dFunction
            return Block2dFunction.apply(x, self.conv.weight, self.conv.bias, self._kind, self._stride, self._pack2d)
        _torch_dispatch(x, 'CRB2d', self._kind)
        return F.batch_norm(F.relu(self.conv(_nchw(x))), None, None, None, None, True, 0.0, cfg.eps)


class DeCRB2d(nn.Module):
    """ConvTranspose2d -> ReLU -> BN2d (reference Blocks.py:42-51).  Inside the RPN the default RPN.forward reads this module's
    parameters and runs modules/rpn_frames.py; called on its own it is ONE autograd node on the same kernels (3x3 / stride 1 as
    a convolution with the flipped kernel, kernel = stride as a row GEMM + pixel shuffle: modules/layers/Block2d.py).  The torch
    / MIOpen form remains as the comparison path (RPN.forward_torch) and for shapes the kernels do not cover."""

    def __init__(self, cin, cout, k, s, p):
        super().__init__()
        _hip.require_plain_batchnorm()
        self.deconv = nn.ConvTranspose2d(cin, cout, k, s, p)
        self.bn = nn.BatchNorm2d(cout, eps=cfg.eps, affine=cfg.bnaffine, track_running_stats=cfg.bntrack)
        from modules.layers import Block2d
        self._kind = Block2d.kind_of('deconv', k, s, p, cin, cout)
        self._stride = s
        object.__setattr__(self, '_pack2d', Block2d._Pack())

    def forward(self, x):
        if _block2d_on(x, self._kind):
            from modules.layers.Block2d import Block2dFunction
            out = Block2dFunction.apply(x, self.deconv.weight, self.deconv.bias, self._kind, self._stride, self._pack2d)
            return _nchw(out) if _IN_FORWARD_TORCH[0] else out    # inside RPN.forward_torch: joins the other (NCHW) branches
        _torch_dispatch(x, 'DeCRB2d', self._kind)
        return F.batch_norm(F.relu(self.deconv(_nchw(x))), None, None, None, None, True, 0.0, cfg.eps)
