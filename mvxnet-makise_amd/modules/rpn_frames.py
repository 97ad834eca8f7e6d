"""Region proposal network on frame sets: RPN.forward and its backward (modules/voxelnet/Pipe.py:45-75) for all frames
of a step, on the HIP kernels -- no MIOpen, no autograd.

    blk1 = CRB2d(128,128,3,2,1) + 3 x CRB2d(128,128,3,1,1)        -> x1 (H/2 x W/2)
    blk2 = CRB2d(128,128,3,2,1) + 5 x CRB2d(128,128,3,1,1)        -> x2 (H/4 x W/4)
    blk3 = CRB2d(128,256,3,2,1) + 5 x CRB2d(256,256,3,1,1)        -> x3 (H/8 x W/8)
    up   = cat(DeCRB2d(128,256,3,1,1)(x1), DeCRB2d(128,256,2,2,0)(x2), DeCRB2d(256,256,4,4,0)(x3))
    score, reg = sigmoid(Conv2d(768,2,1)(up)), Conv2d(768,14,1)(up)

every CRB2d / DeCRB2d = conv -> ReLU -> BatchNorm with per-frame batch statistics (Blocks.py:31-51).  How each piece runs:
  * 3x3 stride-1 blocks (and deconv1, a stride-1 transposed convolution = a convolution with the flipped, transposed
    kernel): the MFMA gather / wgrad kernels of csrc/conv3d.hip, one plane per frame (mvx_conv2d_*_frames);
  * the three stride-2 blocks: the same kernels with a 2x2 tap window on the space-to-depth image of the input (the weight
    is rearranged, include/mvx_hip.h); the first one reads the CML output planes directly, so the (1,128,H,W) reshape of
    VoxelNet.py:36 is never materialised on this path;
  * deconv2 / deconv3 (kernel = stride): row GEMMs (csrc/linear.hip) whose output is normalised and pixel-shuffled into
    its channel slice of the 768-channel concat buffer by one pass (csrc/rpn.hip);
  * the two heads: one row GEMM with 16 output columns.
Maps are channels-last [F][h][w][c].  Parameter gradients are ADDED into the existing .grad buffers, weight-gradient
kernels run on the side stream.
"""
import torch

import modules.config as cfg
from modules import _hip
from modules import Extension as X

R = _hip.STATS_REPLICAS
TAPS2 = 16          # MVX_FLAG_TAPS2


ROW_WGRAD_SIDE = __import__('os').environ.get('MVX_RPN_ROW_WGRAD_SIDE', '1') != '0'   # heads / deconv2 / deconv3 weight gradients on the side stream (A/B: 0)


def _split():
    """convmath: bf16x3 / bf16x6 -> the 3x3 convolutions run on the split-MFMA kernels (csrc/conv3d_split.hip): the number of
    bf16 pieces per operand (0 = exact-f32 kernels)."""
    return _hip.split_pieces()


_GRAD_TARGETS = None        # id(parameter) -> tensor the gradient is ADDED into instead of .grad (the autograd wrapper)


class grad_targets:
    """``with grad_targets({id(p): buffer})``: rpn_backward adds every parameter gradient into the given buffers instead of
    the parameters' .grad (modules/voxelnet/Pipe.py RPNFunction hands them to the autograd engine)."""

    def __init__(self, mapping):
        self.mapping = mapping

    def __enter__(self):
        global _GRAD_TARGETS
        self.old, _GRAD_TARGETS = _GRAD_TARGETS, self.mapping
        return self

    def __exit__(self, *exc):
        global _GRAD_TARGETS
        _GRAD_TARGETS = self.old
        return False


def _grad_of(p):
    if _GRAD_TARGETS is not None:
        return _GRAD_TARGETS[id(p)]
    if p.grad is None or not p.grad.is_contiguous():
        raise X.MvxHipError('the frame-set path adds gradients into existing contiguous .grad buffers (GradBucket)')
    return p.grad


_S2D_INDEX = {}


def _s2d_index(cout, cin, planes, dev):
    """Gather index / mask that rearrange a stride-2 kernel W (cout, cin, 3, 3) into the 2x2-window kernel over the
    space-to-depth channels: W2[co][(p, d, c)][ta][tb] = W[co][c*planes + d][a][b] with a -> (ta, pr): 0 -> (0,1), 1 -> (1,0),
    2 -> (1,1) (b likewise), zeros elsewhere.  cin = planes * C: with planes > 1 the reference channel index is c*planes + d
    (VoxelNet.py:36), the space-to-depth image holds the planes side by side (d-major)."""
    key = (cout, cin, planes, str(dev))
    hit = _S2D_INDEX.get(key)
    if hit is None:
        C = cin // planes
        amap = {0: (0, 1), 1: (1, 0), 2: (1, 1)}
        idx = torch.zeros((4 * cin, 3, 3), dtype=torch.int64)
        mask = torch.zeros((4 * cin, 3, 3), dtype=torch.float32)
        ch = torch.arange(cin)
        d, c = ch // C, ch % C                       # s2d channel (within a parity block) = d*C + c
        ref = c * planes + d                         # the reference's channel
        for a in range(3):
            for b in range(3):
                (ta, pr), (tb, pc) = amap[a], amap[b]
                p = pr * 2 + pc
                idx[p * cin + ch, ta, tb] = ref * 9 + a * 3 + b
                mask[p * cin + ch, ta, tb] = 1.0
        hit = (idx.reshape(-1).to(dev), mask.reshape(-1).to(dev), mask.reshape(-1).bool().nonzero().reshape(-1).to(dev))
        _S2D_INDEX[key] = hit
    return hit


def _s2d_weight(w, planes):
    cout, cin = w.shape[0], w.shape[1]
    idx, mask, _ = _s2d_index(cout, cin, planes, w.device)
    return (w.reshape(cout, cin * 9)[:, idx] * mask).view(cout, 4 * cin, 3, 3)


def _s2d_weight_grad(dw2, w, planes):
    """ADD the gradient of the rearranged kernel back into the layout of ``w``'s gradient."""
    cout, cin = w.shape[0], w.shape[1]
    idx, _, valid = _s2d_index(cout, cin, planes, w.device)
    _grad_of(w).view(cout, cin * 9).index_add_(1, idx[valid], dw2.reshape(cout, -1)[:, valid])


class _Packs:
    """Kernel-layout copies of the (possibly rearranged) convolution weights of one RPN, refreshed when a parameter changes."""

    def __init__(self):
        self.cache = {}

    def get(self, key, w, make, for_dgrad):
        split = _split()
        tag = (w._version, w.data_ptr())
        hit = self.cache.get((key, for_dgrad, split))
        if hit is None or hit[0] != tag:
            w2 = make().detach()
            if split:                      # bf16x3: the 2-D kernel as the middle depth slice of a 3-D one, hi/lo split pack
                w3 = torch.zeros(w2.shape[:2] + (3, 3, 3), dtype=torch.float32, device=w2.device)
                w3[:, :, 1] = w2
                hit = (tag, _hip.conv3d_pack(w3, for_dgrad, split=split))
            else:
                hit = (tag, _hip.conv3d_pack(w2, for_dgrad))
            self.cache[(key, for_dgrad, split)] = hit
        return hit[1]


def _packs_of(rpn):
    pk = rpn.__dict__.get('_mvx_packs')
    if pk is None:
        pk = _Packs()
        object.__setattr__(rpn, '_mvx_packs', pk)
    return pk


def _desc(F):
    return X.FramesDesc.make([0] * (F + 1), [0] * (F + 1), 1)


def _conv(x, wpk, bias, F, h, w, cin, cout, flags, eps):
    """y = ReLU(conv(x) + b) with per-frame statistics finalised in the launch -> (y, mean_inv (F,2,cout))."""
    dev = x.device
    y = torch.empty((F, h, w, cout), dtype=torch.float32, device=dev)
    stats, fz = _hip._acc_f64((F, R, 2, cout), dev)
    fin = _hip._fin_slot(dev, fz)
    if fin is None:
        fin = torch.zeros((1,), dtype=torch.float64, device=dev)
    mi = torch.empty((F, 2, cout), dtype=torch.float32, device=dev)
    if _split():                               # split arithmetic (bf16x3 / bf16x6), same window / structural-zero skipping
        with _hip._Timed('rpn_conv', 2.0 * F * h * w * cin * cout * (2.25 if flags & TAPS2 else 9) if _hip.KERNEL_TIMERS is not None else 0):
            X.check(X.lib.mvx_conv2d_forward_split_frames(X.ptr(x), X.ptr(wpk), X.ptr(bias), X.ptr(y), X.ptr(stats), h, w, cin, cout,
                                                          _hip.FLAG_RELU | fz | flags | _hip.split_flags(_split()), F, X.stream()), 'mvx_conv2d_forward_split_frames')
        X.check(X.lib.mvx_bn_finalize_frames(X.ptr(stats), float(h * w), float(eps), X.ptr(mi), cout, F, X.stream()),
                'mvx_bn_finalize_frames')
        return y, mi
    nt = 2.25 if flags & TAPS2 else 9          # space-to-depth form: 9 of the 16 (window tap, parity) blocks are executed
    with _hip._Timed('rpn_conv', 2.0 * F * h * w * cin * cout * nt if _hip.KERNEL_TIMERS is not None else 0):
        X.check(X.lib.mvx_conv2d_forward_frames(X.ptr(x), X.ptr(wpk), X.ptr(bias), X.ptr(y), X.ptr(stats), h, w, cin, cout,
                                                _hip.FLAG_RELU | fz | flags, X.ptr(fin), float(eps), X.ptr(mi),
                                                X.ptr(_hip._work_counter(dev)), F, X.stream()), 'mvx_conv2d_forward_frames')
    return y, mi


def _bn_apply(y, mi, F):
    C = mi.shape[-1]
    out = torch.empty_like(y)
    X.check(X.lib.mvx_bn_apply_frames(X.ptr(y), X.ptr(mi), X.ptr(out), y.numel() // C, C, _desc(F).ref(), X.ROWS_GRID, X.stream()),
            'mvx_bn_apply_frames')
    return out


def _bn_bwd(g, y, mi, F, bias):
    C = mi.shape[-1]
    dz = torch.empty_like(y)
    scratch, fz = _hip._acc_f64((X.lib.mvx_bn_backward_scratch_bytes_frames(C, F) // 8,), y.device)
    amax = _hip.new_amax(y.device)
    X.check(X.lib.mvx_bn_relu_backward_frames(X.ptr(g), X.ptr(y), X.ptr(mi), 1.0, X.ptr(dz), X.ptr(_grad_of(bias)), X.ptr(scratch),
                                              None, y.numel() // C, C, _hip.FLAG_ACCUMULATE | fz, _desc(F).ref(), X.ROWS_GRID,
                                              X.ptr(amax), X.stream()), 'mvx_bn_relu_backward_frames')
    return _hip.tag_amax(dz, amax)                      # max |dz|: the range the fp16x3 kernels scale dz by


def _retag(t, like):
    """A view of a tagged gradient keeps the tag."""
    return _hip.tag_amax(t, _hip.amax_of(like))


def _dgrad(dz, wpd, F, h, w, cin, cout, flags):
    dx = torch.empty((F, h, w, cin), dtype=torch.float32, device=dz.device)
    if _split():
        with _hip._Timed('rpn_conv', 2.0 * F * h * w * cin * cout * (2.25 if flags & TAPS2 else 9) if _hip.KERNEL_TIMERS is not None else 0):
            _hip.bind_amax(_split(), dz)
            X.check(X.lib.mvx_conv2d_dgrad_split_frames(X.ptr(dz), X.ptr(wpd), X.ptr(dx), h, w, cin, cout, flags | _hip.split_flags(_split()), F, X.stream()),
                    'mvx_conv2d_dgrad_split_frames')
        return dx
    nt = 2.25 if flags & TAPS2 else 9
    with _hip._Timed('rpn_conv', 2.0 * F * h * w * cin * cout * nt if _hip.KERNEL_TIMERS is not None else 0):
        X.check(X.lib.mvx_conv2d_dgrad_frames(X.ptr(dz), X.ptr(wpd), X.ptr(dx), h, w, cin, cout, flags,
                                              X.ptr(_hip._work_counter(dz.device)), F, X.stream()), 'mvx_conv2d_dgrad_frames')
    return dx


def _wgrad(x, dz, F, h, w, cin, cout, flags, into=None):
    """dW (cout, cin, 3, 3) over all frames, on the side stream; ADDED into ``into`` or returned."""
    dev = x.device
    # split arithmetic: the same entry point and workgroup decomposition (conv3d_wgrad4s, csrc/conv3d.hip) -- MVX_FLAG_SPLIT[3]
    sp = _hip.split_flags(_split(), True)
    nbytes = X.lib.mvx_conv2d_wgrad_workspace_bytes_frames(h, w, cin, cout, F)
    dw = into if into is not None else torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=dev)
    nt = 2.25 if flags & TAPS2 else 9
    with _hip._SideStream(x, dz, dw), _hip._Timed('rpn_wgrad', 2.0 * F * h * w * cin * cout * nt if _hip.KERNEL_TIMERS is not None else 0):
        ws = _hip.workspace(nbytes, dev, 'rpn_wgrad_side')
        _hip.bind_amax(_split(), None, dz)
        X.check(X.lib.mvx_conv2d_wgrad_frames(X.ptr(x), X.ptr(dz), X.ptr(dw), h, w, cin, cout,
                                              flags | sp | (_hip.FLAG_ACCUMULATE if into is not None else 0), X.ptr(ws), ws.numel(), F,
                                              X.stream()), 'mvx_conv2d_wgrad_frames')
    return dw


def _s2d(x, F, planes, h, w, C):
    out = torch.empty((F, h // 2, w // 2, 4 * planes * C), dtype=torch.float32, device=x.device)
    X.check(X.lib.mvx_space_to_depth_frames(X.ptr(x), X.ptr(out), F, planes, h, w, C, 0, X.stream()), 'mvx_space_to_depth_frames')
    return out


def _d2s(xs, F, planes, h, w, C):
    out = torch.empty((F * planes, h, w, C), dtype=torch.float32, device=xs.device)
    X.check(X.lib.mvx_space_to_depth_frames(X.ptr(xs), X.ptr(out), F, planes, h, w, C, 1, X.stream()), 'mvx_space_to_depth_frames')
    return out


def rpn_forward(rpn, x_cl, F, planes, H, W, Cp):
    """x_cl: the normalised CML output, channels-last planes [F*planes][H][W][Cp] (BEV channel c*planes + d).
    Returns (heads (F*H/2*W/2, 16) = [cls logits (2) | reg (14)] per BEV cell, saved state)."""
    dev = x_cl.device
    eps = cfg.eps
    _hip.require_plain_batchnorm()
    pk = _packs_of(rpn)
    S = {'F': F, 'geom': (planes, H, W, Cp), 'blocks': []}
    x, h, w, C, pl = x_cl, H, W, Cp, planes
    for bi, blk in enumerate((rpn.blk1, rpn.blk2, rpn.blk3)):
        layers = []
        for li, m in enumerate(blk):
            wt, b = m.conv.weight, m.conv.bias
            cout = wt.shape[0]
            if li == 0:                                   # stride 2: 2x2 window on the space-to-depth image
                xs = _s2d(x, F, pl, h, w, C)
                h, w = h // 2, w // 2
                cin = 4 * pl * C
                wpk = pk.get(('s2', bi), wt, lambda wt=wt, pl=pl: _s2d_weight(wt, pl), False)
                y, mi = _conv(xs, wpk, b, F, h, w, cin, cout, TAPS2, eps)
                layers.append(dict(kind='s2', x=xs, y=y, mi=mi, m=m, cin=cin, cout=cout, h=h, w=w, planes=pl, cfull=C, bi=bi))
                pl = 1
            else:
                cin = C
                wpk = pk.get(('s1', bi, li), wt, lambda wt=wt: wt, False)
                y, mi = _conv(x, wpk, b, F, h, w, cin, cout, 0, eps)
                layers.append(dict(kind='s1', x=x, y=y, mi=mi, m=m, cin=cin, cout=cout, h=h, w=w, bi=bi, li=li))
            x = _bn_apply(y, mi, F)
            C = cout
        S['blocks'].append(dict(layers=layers, out=x, h=h, w=w, C=C))
    (x1, h1, w1), (x2, h2, w2), (x3, h3, w3) = [(b['out'], b['h'], b['w']) for b in S['blocks']]
    rows = F * h1 * w1
    up = torch.empty((rows, 768), dtype=torch.float32, device=dev)
    # deconv1: stride-1 transposed convolution = convolution with the flipped kernel, channel axes swapped
    d1 = rpn.deconv1
    wc = pk.get(('d1',), d1.deconv.weight, lambda: d1.deconv.weight.flip(2, 3).transpose(0, 1).contiguous(), False)
    y, mi = _conv(x1, wc, d1.deconv.bias, F, h1, w1, 128, 256, 0, eps)
    X.check(X.lib.mvx_bn_apply_strided_frames(X.ptr(y), X.ptr(mi), X.ptr(up), rows, 256, 768, 0, F, 0, X.stream()),
            'mvx_bn_apply_strided_frames')
    S['d1'] = dict(x=x1, y=y, mi=mi)
    # deconv2 / deconv3: kernel = stride -> row GEMM + normalising pixel shuffle into the concat slice
    S['dk'] = []
    for xk, hk, wk, m, s, off in ((x2, h2, w2, rpn.deconv2, 2, 256), (x3, h3, w3, rpn.deconv3, 4, 512)):
        wt, b = m.deconv.weight, m.deconv.bias                    # (Cin, Cout, s, s)
        cin, cout = wt.shape[0], wt.shape[1]
        w_all = wt.permute(2, 3, 1, 0).reshape(s * s * cout, cin).contiguous()     # row (i*s+j)*Cout + co
        xr = xk.view(F * hk * wk, cin)
        t, _ = _hip.linear_forward(xr, w_all, b.repeat(s * s), relu=True, want_stats=False, split=_hip.row_split('rpn'))
        stats = torch.empty((F, R, 2, cout), dtype=torch.float64, device=dev)
        X.check(X.lib.mvx_row_stats_frames(X.ptr(t), X.ptr(stats), t.numel() // cout, cout, F, X.stream()), 'mvx_row_stats_frames')
        mi = torch.empty((F, 2, cout), dtype=torch.float32, device=dev)
        X.check(X.lib.mvx_bn_finalize_frames(X.ptr(stats), float(hk * wk * s * s), float(eps), X.ptr(mi), cout, F, X.stream()),
                'mvx_bn_finalize_frames')
        X.check(X.lib.mvx_d2s_bn_apply_frames(X.ptr(t), X.ptr(mi), X.ptr(up), F, hk, wk, s, cout, 768, off, 0, X.stream()),
                'mvx_d2s_bn_apply_frames')
        S['dk'].append(dict(x=xr, t=t, mi=mi, m=m, s=s, off=off, h=hk, w=wk, w_all=w_all, cin=cin, cout=cout))
    # heads (Pipe.py:64-65,74): one GEMM, 16 columns
    w_heads = torch.cat([rpn.cls.weight.view(2, 768), rpn.reg.weight.view(14, 768)]).contiguous()
    b_heads = torch.cat([rpn.cls.bias, rpn.reg.bias])
    heads, _ = _hip.linear_forward(up, w_heads, b_heads, relu=False, want_stats=False)
    S.update(up=up, w_heads=w_heads, h1=h1, w1=w1)
    return heads, S


def split_heads(heads, F, h1, w1):
    """(score (F,2,h1,w1) = sigmoid(cls), reg (F,14,h1,w1)) as views of the channels-last head output (no copies of reg)."""
    v = heads.view(F, h1, w1, 16)
    return torch.sigmoid(v[..., :2]).permute(0, 3, 1, 2), v[..., 2:].permute(0, 3, 1, 2)


_MUTATE = {}        # TESTS ONLY (tests/test_rpn_gpu.py): name -> factor applied to one term of the RPN backward, to prove that the
                    # full-size gradient check would notice a 1 % error in it.  Empty in every product run.


def _mut(name, t):
    return _hip.mutate(_MUTATE, name, t) if _MUTATE else t


def rpn_backward(rpn, S, d_heads):
    """Backward from dL/d(heads) (F*h1*w1, 16); returns dL/d(x_cl) in the layout of rpn_forward's input.
    Needs _hip.ASYNC_WGRAD semantics: weight gradients are produced on the side stream (join before the optimizer)."""
    F = S['F']
    planes, H, W, Cp = S['geom']
    pk = _packs_of(rpn)
    cap = S.get('capture')          # tests: list receiving (layer key, upstream gradient, input gradient) per conv layer
    up, w_heads, h1, w1 = S['up'], S['w_heads'], S['h1'], S['w1']
    dev = up.device
    rows = up.shape[0]
    d_heads = d_heads.contiguous()
    if _split() == 4 and _hip.amax_of(d_heads) is None:
        _hip.tensor_amax(d_heads)        # fp16x3: the loss gradient's range (9 MB: 5 us), else its weight gradient runs in bf16x6
    # heads
    # (the row-GEMM weight gradients of the heads and of deconv2 / deconv3 run on the side stream like the convolutions': they
    # were 0.41 ms of main-queue time per --mode full step)
    with (_hip._SideStream(up, d_heads) if ROW_WGRAD_SIDE else _hip._Inline()):
        dwh = _hip.linear_wgrad(up, d_heads)                               # (16, 768)
    with _hip._SideStream(dwh, d_heads):
        _grad_of(rpn.cls.weight).view(2, 768).add_(dwh[:2])
        _grad_of(rpn.reg.weight).view(14, 768).add_(dwh[2:])
        db = d_heads.sum(0)
        _grad_of(rpn.cls.bias).add_(db[:2])
        _grad_of(rpn.reg.bias).add_(db[2:])
    g_up, _ = _hip.linear_forward(d_heads, w_heads, None, relu=False, want_stats=False, w_transposed=True)      # (rows, 768)
    g_up = _mut('heads_dgrad', g_up)
    # deconv2 / deconv3
    g_in = {}
    for rec in S['dk']:
        m, s, off, hk, wk, cin, cout = rec['m'], rec['s'], rec['off'], rec['h'], rec['w'], rec['cin'], rec['cout']
        gt = torch.empty_like(rec['t'])
        X.check(X.lib.mvx_d2s_bn_apply_frames(X.ptr(gt), None, X.ptr(g_up), F, hk, wk, s, cout, 768, off, 1, X.stream()),
                'mvx_d2s_bn_apply_frames')
        dz0 = _bn_bwd(gt.view(-1, cout), rec['t'].view(-1, cout), rec['mi'], F, m.deconv.bias)
        dz = _retag(dz0.view(rec['t'].shape), dz0)
        with (_hip._SideStream(rec['x'], dz) if ROW_WGRAD_SIDE else _hip._Inline()):
            dw_all = _hip.linear_wgrad(rec['x'], dz)                       # (s*s*cout, cin)
        with _hip._SideStream(dw_all):
            _grad_of(m.deconv.weight).add_(dw_all.view(s, s, cout, cin).permute(3, 2, 0, 1))
        # input gradient with the weight as a row-major [cin][s*s*cout] matrix (both operands read along k)
        gx, _ = _hip.linear_forward(dz, rec['w_all'].t().contiguous(), None, relu=False, want_stats=False, label='linear_dgrad',
                                    split=_hip.row_split('dgrad'))
        g_in[s] = _mut('deconv%d_dgrad' % s, gx).view(F, hk, wk, cin)
    # deconv1
    d1 = rpn.deconv1
    g1 = torch.empty((rows, 256), dtype=torch.float32, device=dev)
    X.check(X.lib.mvx_bn_apply_strided_frames(X.ptr(g1), None, X.ptr(g_up), rows, 256, 768, 0, F, 1, X.stream()),
            'mvx_bn_apply_strided_frames')
    r1 = S['d1']
    dz = _bn_bwd(g1.view(F, h1, w1, 256), r1['y'], r1['mi'], F, d1.deconv.bias)
    dwc = _wgrad(r1['x'], dz, F, h1, w1, 128, 256, 0)                      # gradient of the flipped / transposed kernel
    with _hip._SideStream(dwc):
        _grad_of(d1.deconv.weight).add_(dwc.transpose(0, 1).flip(2, 3))
    wcd = pk.get(('d1',), d1.deconv.weight, lambda: d1.deconv.weight.flip(2, 3).transpose(0, 1).contiguous(), True)
    g_x1 = _mut('deconv1_dgrad', _dgrad(dz, wcd, F, h1, w1, 128, 256, 0))
    # blocks, last first; x2 and x1 collect the gradients of both their consumers
    g_next = {2: g_in[4], 1: g_in[2], 0: g_x1}
    g = g_next[2]
    for bi in (2, 1, 0):
        if bi < 2:
            g = g_next[bi] + g                      # deconv path + the next block's stride-2 layer
        sums = None                                 # BatchNorm-backward sums of the layer about to be processed, when its producer made them
        layers = S['blocks'][bi]['layers']
        for rec in reversed(layers):
            m = rec['m']
            wt = m.conv.weight
            h, w, cin, cout = rec['h'], rec['w'], rec['cin'], rec['cout']
            g_up_layer = g
            dz = _mut('bn_bwd_b%d' % bi, _bn_bwd(g, rec['y'], rec['mi'], F, m.conv.bias))
            if rec['kind'] == 's1':
                _wgrad(rec['x'], dz, F, h, w, cin, cout, 0, into=_grad_of(wt))
                wpd = pk.get(('s1', rec['bi'], rec['li']), wt, lambda wt=wt: wt, True)
                g = _mut('s1_dgrad_b%d' % bi, _dgrad(dz, wpd, F, h, w, cin, cout, 0))
            else:
                pl, Cf = rec['planes'], rec['cfull']
                dw2 = _wgrad(rec['x'], dz, F, h, w, cin, cout, TAPS2)
                with _hip._SideStream(dw2):
                    _s2d_weight_grad(dw2, wt, pl)
                wpd = pk.get(('s2', rec['bi']), wt, lambda wt=wt, pl=pl: _s2d_weight(wt, pl), True)
                gs = _mut('s2_dgrad_b%d' % bi, _dgrad(dz, wpd, F, h, w, cin, cout, TAPS2))             # gradient of the space-to-depth image
                g = _d2s(gs, F, pl, 2 * h, 2 * w, Cf)
                if pl == 1:
                    g = g.view(F, 2 * h, 2 * w, Cf)
            if cap is not None:
                cap.append(((rec['bi'], rec.get('li', 0)), g_up_layer.clone(), g.clone()))
    return g                                       # [F*planes][H][W][Cp]
