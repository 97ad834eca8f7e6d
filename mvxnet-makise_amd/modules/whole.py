"""MVXNet.forward as ONE autograd node: the frame-set executor (modules/frames.py + modules/rpn_frames.py) on a set of one
frame behind the reference's interface (MVXNet.py:21-27, train.py:131 / :161).

The per-module path (modules/voxelnet, modules/imhead, modules/layers under the autograd engine) costs the host ~13 ms per
frame -- one Python autograd node with saved-tensor bookkeeping per layer, ~400 launches -- against ~9 ms of GPU work, so
the drop-in interface was host-bound.  Here forward enqueues the same kernels the benchmarked step uses (fusion sampling +
MLP on compact rows, VFE stack, reindex + conv1 on the voxel rows, CML with the background rewrite, RPN) without any graph
node in between, and backward runs the mirrored chain (restricted CML backward, weight gradients on the side stream).
Parameter gradients are handed to the autograd engine like any node's (or added straight into the .grad views of a
GradBucket when the training pipeline switched that on, _hip.GRAD_SINK)."""
import torch

import modules.config as cfg
from modules import _hip
from modules import Extension as X


def supported(model, voxels, idx):
    """The single-node path needs what the frame-set kernels need: a contiguous f32 (1,N,T,9) voxel tensor with N > 0 on the
    GPU, the background rewrite on, exact-f32 or bf16x3 arithmetic, maps whose sides divide by 8 (three stride-2 layers)."""
    from modules.layers.Blocks import conv_background_on
    if not (torch.is_tensor(voxels) and voxels.is_cuda and voxels.dtype == torch.float32 and voxels.dim() == 4):
        return False
    if voxels.shape[0] != 1 or voxels.shape[1] == 0 or voxels.shape[3] != 9 or not voxels.is_contiguous():
        return False
    if not (torch.is_tensor(idx) and idx.is_cuda and idx.dtype == torch.int64 and idx.shape == (voxels.shape[1], 4)):
        return False
    if not conv_background_on() or not bool(cfg.config.get('rpn_hip', True)) or not model.backbone.sparse_first_layer:
        return False
    return cfg.voxelshape[0] % 8 == 0 and cfg.voxelshape[1] % 8 == 0


class WholeModelFunction(torch.autograd.Function):
    """(voxels (1,N,T,9), idx (N,4), FPN maps) -> heads (H/2 * W/2, 16) = [cls logits | reg]."""

    @staticmethod
    def forward(ctx, voxels, idx, model, imgs, imsize_hw, *params):
        from modules import frames as fr
        from modules import rpn_frames as rf
        dev = voxels.device
        n, t = voxels.shape[1], voxels.shape[2]
        fs = getattr(voxels, '_mvx_fs', None)        # built with the voxels by pipeline.voxelize_batch(with_maps=True)
        if fs is None or fs.voxels.data_ptr() != voxels.data_ptr() or fs.Vt != n or fs.T != t or fs.desc is None:
            fs = fr.FrameSet(voxels[0], idx.contiguous(), [0, n], t)
            real_off = fs.enqueue_map().tolist()     # zeroes the padded rows in place (imhead/Pipe.py:54-59); one host read
            fs.finish_map(real_off)
        model.prepack()
        statuses = []
        _hip.arena_begin(dev, doubles=1 << 21)
        try:
            feat, saved = fr.rows_forward(model, fs, [imgs], imsize_hw, statuses)
            fr.cml_forward(model, fs, feat, saved, statuses, want_bev=False)
            heads, rs = rf.rpn_forward(model.backbone.rpn, saved.x3, 1, saved.D3, saved.H, saved.W, saved.C3)
        finally:
            _hip.arena_end()
        ctx.model, ctx.saved, ctx.rs, ctx.params = model, saved, rs, params
        model.__dict__.setdefault('_mvx_status', []).extend(statuses)      # checked by whole.forward / take_status, never here
        return heads

    @staticmethod
    def backward(ctx, g_heads):
        from modules import frames as fr
        from modules import rpn_frames as rf
        model, saved, rs, params = ctx.model, ctx.saved, ctx.rs, ctx.params
        dev = g_heads.device
        direct = _hip.GRAD_SINK and all(p.grad is not None and p.grad.is_contiguous() for p in params)
        targets, views = None, None
        if not direct:
            flat = torch.zeros((sum(p.numel() for p in params),), dtype=torch.float32, device=dev)
            views, off = [], 0
            for p in params:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            targets = {id(p): v for p, v in zip(params, views)}
        old_async, _hip.ASYNC_WGRAD = _hip.ASYNC_WGRAD, True
        _hip.arena_begin(dev, doubles=1 << 21)
        try:
            with fr.grad_targets(targets), rf.grad_targets(targets):
                g_cl = rf.rpn_backward(model.backbone.rpn, rs, g_heads.contiguous())
                fr.rows_backward(model, saved, fr.cml_backward(model, saved, None, g_cl=g_cl))
        finally:
            _hip.ASYNC_WGRAD = old_async
            _hip.arena_end()
            _hip.join_side_stream(dev)
            if views is not None:
                # not the direct sink: autograd adds the returned views into .grad on this stream AFTER the join, so the tail
                # pair no longer bounds the gradient writes
                _hip.drop_tail(dev)
        ctx.saved = ctx.rs = None
        return (None, None, None, None, None) + (tuple(views) if views is not None else (None,) * len(params))


def _check(statuses):
    """The reference asserts that every projected point lies inside the feature map (imhead/Pipe.py:71, a device sync)."""
    if statuses and int(torch.stack([t.reshape(()) for t in statuses]).max()) & 1:
        raise AssertionError('projected point outside the feature map')


def take_status(model):
    """The device status words of the forwards run since the last check (a training loop reads them with whatever it reads
    anyway, e.g. once per logging interval: a host read drains the stream, so none is made inside forward / backward)."""
    return model.__dict__.pop('_mvx_status', [])


MAX_PENDING_STATUS = 512          # forwards whose status words may stay unread (then one read is made)


def _imsize_hw(imsize):
    """[height, width] as Python floats.  A device tensor is read ONCE per (tensor, version): the reference hands the same
    ``imsize`` tensor to every forward (train.py:131), and a read per call would drain the stream per frame."""
    if not torch.is_tensor(imsize):
        return [float(imsize[0]), float(imsize[1])]
    hit = imsize.__dict__.get('_mvx_hw')
    if hit is None or hit[0] != imsize._version:
        hit = (imsize._version, [float(v) for v in imsize.tolist()])
        imsize.__dict__['_mvx_hw'] = hit
    return hit[1]


def _all_params(model):
    """model.parameters() as a list kept on the model (walking the module tree costs ~0.3 ms per call); made again when any
    module of the tree was added, removed or REPLACED (the signature is the identity of every module's sub-module table and
    of the sub-modules themselves, four levels deep = every module of MVXNet: replacing backbone.rpn.cls keeps the counts but
    not the identities; ~10 us)."""
    sig = []
    level = [model]
    for _ in range(4):              # model -> head / backbone -> svfe, cml, rpn, fusion ... -> their blocks -> conv / bn / fc
        nxt = []
        for m in level:
            for c in m._modules.values():
                if c is not None:
                    sig.append(id(c))
                    nxt.append(c)
        level = nxt
    sig = tuple(sig)
    hit = model.__dict__.get('_mvx_params')
    if hit is None or hit[0] != sig:
        hit = (sig, list(model.parameters()))
        model.__dict__['_mvx_params'] = hit
    return hit[1]


def forward(model, voxels, imgs, idx, imsize):
    """MVXNet.forward on the single node: (score (1,2,H/2,W/2), reg (1,14,H/2,W/2)).  The data-dependent status words of the
    sampling / scatter kernels stand for the reference's assert (imhead/Pipe.py:71); they are collected on the model and
    read without stalling the training stream (take_status), at once for a no-grad call."""
    hw = _imsize_hw(imsize)
    params = [p for p in _all_params(model) if p.requires_grad]
    heads = WholeModelFunction.apply(voxels, idx, model, imgs, hw, *params)
    # no backward will follow (inference): the reference's assert, now; training: at most every MAX_PENDING_STATUS words
    if not heads.requires_grad or len(model.__dict__.get('_mvx_status', ())) > MAX_PENDING_STATUS:
        _check(take_status(model))
    h1, w1 = cfg.voxelshape[0] // 2, cfg.voxelshape[1] // 2
    v = heads.view(1, h1, w1, 16)
    return torch.sigmoid(v[..., :2]).permute(0, 3, 1, 2), v[..., 2:].permute(0, 3, 1, 2)
