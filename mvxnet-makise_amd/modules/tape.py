"""Straight-line execution of the hot path: the same layer code (the ``forward`` / ``backward`` static methods
of the autograd Functions in modules/layers, modules/voxelnet, modules/imhead) without the autograd engine.

MVXNet.middle on compact rows is a pure chain -- every node has one differentiable input, the previous
node's output -- so its backward is the reversed list of the nodes.  Calling the static methods directly
with a plain context object removes what the engine costs per node on the host (graph node, context,
saved-variable wrappers, a thread hop per backward node): about half of the enqueue time of a frame.
Parameter gradients go straight into the ``.grad`` buffers (modules._hip.GRAD_SINK), exactly as on the
autograd path; tests/test_voxelnet_gpu.py compares the two paths."""
import torch

import modules.config as cfg
from modules import _hip
from modules.layers.Blocks import (CRB3dFunction, FCNFunction, VoxelGemmCRB3dFunction, conv_background_on)
from modules.voxelnet.Pipe import CompactInputFunction, FCNMaxFunction, VFEFunction
from modules.imhead.Pipe import _channels_last_levels


class Node:
    """Stands for the autograd context of one Function call."""

    def __init__(self, fn, needs_input_grad):
        self.fn = fn
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass


def _call(tape, fn, needs_first, *args):
    node = Node(fn, (needs_first,) + (False,) * (len(args) - 1))
    out = fn.forward(node, *args)
    tape.append(node)
    return out


def middle_forward(model, voxels, fpn_levels, idx, imsize, prepared, status_sink):
    """Forward of MVXNet.middle (compact rows) for one frame; returns (mid (1,128,H,W), tape)."""
    head, bb = model.head, model.backbone
    tape = []
    v = voxels.squeeze(0) if voxels.dim() == 4 else voxels[0]
    n, t, c = v.shape
    rows = n * t
    vox2d = v.view(rows, c)
    row_map, rows_sel, nr = prepared
    feats = head.extractor(fpn_levels)
    levels = _channels_last_levels(feats, 0)
    width = levels[0].shape[2] * len(levels)
    compact = torch.empty((nr + 1, width), dtype=torch.float32, device=v.device)
    compact[nr].zero_()
    status = _hip.feature_sample(vox2d, levels, (float(imsize[0]), float(imsize[1])), cfg.eps, compact, row_map, rows_sel=rows_sel, n_real=nr)
    status_sink.append(status)
    if _hip.split_pieces() == 4:
        _hip.tensor_amax(compact)                       # fp16x3: the range of the image features (frames.sample_rows does the same)
    row_w = torch.ones((nr + 1,), dtype=torch.float32, device=v.device)
    row_w[nr] = float(rows - nr)
    cr = _hip.CompactRows(row_map, rows_sel, nr, n, t)
    # fusion MLP (imhead/Pipe.py:84-104): the sampled image features carry no gradient
    x = compact
    for i, (w, b) in enumerate(head.fusion._layers()):
        x = _call(tape, FCNFunction, i > 0, x, w, b, cfg.eps, row_w, float(rows), i == 0)
    # concat with the 7 geometric channels (MVXNet.py:26), VFE stack (voxelnet/Pipe.py:5-29), FCN + max (VoxelNet.py:27-33)
    x = _call(tape, CompactInputFunction, True, x, vox2d, cr)
    for vfe in (bb.svfe.vfe1, bb.svfe.vfe2):
        x = _call(tape, VFEFunction, True, x, vfe.fcn.fc.weight, vfe.fcn.fc.bias, cr.V, cr.T, cfg.eps, cr)
    x = _call(tape, FCNMaxFunction, True, x, bb.fcn.fc.weight, bb.fcn.fc.bias, cr.V, cr.T, cfg.eps, cr)
    # reindex + CML (VoxelNet.py:16-22, voxelnet/Pipe.py:31-43), channels-last throughout
    d, h, w_ = cfg.voxelshape[2], cfg.voxelshape[0], cfg.voxelshape[1]
    c1, c2, c3 = bb.cml.conv1, bb.cml.conv2, bb.cml.conv3
    aux = {} if conv_background_on() else None
    x = _call(tape, VoxelGemmCRB3dFunction, True, x, idx.contiguous(), c1.conv.weight, c1.conv.bias, (d, h, w_), c1._sd,
              c1._pd, cfg.eps, aux)
    bg = aux['bg'] if aux else None
    for m in (c2, c3):
        aux = {} if (bg is not None and m is not c3) else None      # nobody reads the background of the last layer
        x = _call(tape, CRB3dFunction, True, x, m.conv.weight, m.conv.bias, m._sd, m._pd, cfg.eps, m._packer, bg, aux)
        bg = aux.get('bg') if aux else None
    return _hip.cl_to_bev(x)[None], tape


def middle_backward(tape, grad_mid, depth):
    """Backward of the chain; parameter gradients are accumulated by the kernels (GRAD_SINK)."""
    assert _hip.GRAD_SINK, 'the tape path adds parameter gradients in the kernels: run it inside the training pipeline'
    g = _hip.bev_to_cl(grad_mid[0].contiguous(), depth)
    for node in reversed(tape):
        res = node.fn.backward(node, g)
        g = res[0]
        node.saved_tensors = ()
        if g is None:
            break
    del tape[:]


def middle_train(model, voxels, fpn_levels, idx, imsize, prepared, status_sink, grad_mid):
    """One frame: forward + backward of the hot path without autograd; returns the middle map."""
    from modules.layers import Blocks
    old, Blocks.RESTRICTED_BACKWARD = Blocks.RESTRICTED_BACKWARD, True     # the chain is known here: one consumer per node
    try:
        with torch.no_grad():
            mid, tape = middle_forward(model, voxels, fpn_levels, idx, imsize, prepared, status_sink)
            depth = mid.shape[1] // 64
            middle_backward(tape, grad_mid, depth)
    finally:
        Blocks.RESTRICTED_BACKWARD = old
    return mid
