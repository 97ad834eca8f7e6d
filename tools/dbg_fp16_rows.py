"""Developer tool: fp16x3 row-GEMM gradients on a gradient tensor with an outlier row (the fusion MLP's shared padded row)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
import torch
from modules import _hip
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
rows, K = 20000, 128
for N in (16, 128):
    for outlier in (1.0, 1e3, 1e6):
        for scale in (1.0, 1e-5):
            x = torch.randn((rows, K), generator=g).to(dev)
            dz = (torch.randn((rows, N), generator=g) * scale).to(dev)
            dz[-1] *= outlier
            w = (torch.randn((K, N), generator=g) * 0.05).to(dev)          # dgrad: dx = dz w^T with weight (K, N)
            ref_w = dz.double().t() @ x.double()
            ref_x = dz.double() @ w.double().t()
            out = {}
            for mode, sp in (('f32', 0), ('bf16x6', 3), ('fp16x3 untagged', 4), ('fp16x3 tagged', 4)):
                d = dz.clone()
                if mode.endswith(' tagged'):
                    _hip.tensor_amax(d)
                dw = _hip.linear_wgrad(x, d, split=sp)
                dx, _ = _hip.linear_forward(d, w, None, relu=False, want_stats=False, split=sp)
                rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
                # the typical rows alone (the outlier dominates the max norm)
                relx = float((dx.double()[:-1] - ref_x[:-1]).abs().max() / ref_x[:-1].abs().max())
                out[mode] = 'dw %.1e dx(all) %.1e dx(typical rows) %.1e' % (rel(dw, ref_w), rel(dx, ref_x), relx)
            print('N', N, 'outlier', outlier, 'scale', scale)
            for k, v in out.items():
                print('   %-16s %s' % (k, v))
