"""RPN + loss stand-in (next scope row, stock PyTorch-ROCm / MIOpen) on the full-size BEV map -- developer tool."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
import modules.config as cfg
from modules.voxelnet.Pipe import RPN
dev = torch.device('cuda')
torch.manual_seed(0)
rpn = RPN().to(dev)
x = torch.randn((1, 128, cfg.voxelshape[0], cfg.voxelshape[1]), device=dev, requires_grad=True)
for hip, cl in ((True, False), (False, False), (False, True)):
    cfg.config['rpn_hip'] = hip
    if cl:
        rpn = rpn.to(memory_format=torch.channels_last)
        xx = x.detach().to(memory_format=torch.channels_last).requires_grad_(True)
    else:
        xx = x
    def step():
        s, r = rpn(xx)
        (s.sum() + r.sum()).backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    print('RPN fwd+bwd (3x3 stride-1 blocks on %s, channels_last input=%s): %.2f ms per frame' % ('HIP MFMA kernels' if hip else 'MIOpen', cl, (time.perf_counter() - t0) / 5 * 1e3))
