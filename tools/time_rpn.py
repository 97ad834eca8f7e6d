"""RPN forward + backward at the full 352x400 map: this library's kernels over a frame set (modules/rpn_frames.py) against
the torch modules on MIOpen, frames/s and ms per frame (developer tool; writes profiles/<tag>_rpn_timing.json).
usage: python tools/time_rpn.py [frames] [tag]"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
TAG = sys.argv[2] if len(sys.argv) > 2 else None
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip, parallel  # noqa: E402
from modules import rpn_frames as rf  # noqa: E402
from modules.voxelnet.Pipe import RPN  # noqa: E402

dev = torch.device('cuda')
torch.manual_seed(0)
rpn = RPN().to(dev)
bucket = parallel.GradBucket(list(rpn.parameters()))
H, W = 352, 400
x_cl = torch.randn((F * 2, H, W, 64), device=dev)
d_heads = torch.randn((F * (H // 2) * (W // 2), 16), device=dev) * 1e-2
_hip.ASYNC_WGRAD = True
_hip.GRAD_SINK = True


def hip_step():
    _hip.arena_begin(dev, doubles=1 << 21)
    with torch.no_grad():
        heads, S = rf.rpn_forward(rpn, x_cl, F, 2, H, W, 64)
        g = rf.rpn_backward(rpn, S, d_heads)
    _hip.arena_end()
    _hip.join_side_stream()
    return g


mid = x_cl.view(F, 2, H, W, 64).permute(0, 4, 1, 2, 3).reshape(F, 128, H, W).contiguous()


def module_step():
    for f in range(F):
        leaf = mid[f:f + 1].clone().requires_grad_(True)
        score, reg = rpn(leaf)
        logits = torch.log(score / (1 - score))
        out = torch.cat([logits, reg], 1)[0].permute(1, 2, 0).reshape(-1, 16)
        (out * d_heads.view(F, -1, 16)[f]).sum().backward()


def clock(fn, n=10, warm=3):
    for _ in range(warm):
        bucket.zero()
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        bucket.zero()
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {'frames': F, 'map': [H, W]}
out['hip_ms_per_frame'] = clock(hip_step) / F
_hip.KERNEL_TIMERS = {}
bucket.zero()
hip_step()
torch.cuda.synchronize()
ev = _hip.KERNEL_TIMERS.get('rpn_conv', [])
_hip.KERNEL_TIMERS = None
ms = sum(s.elapsed_time(e) for s, e, _ in ev)
out['rpn_conv'] = {'launches': len(ev), 'ms_per_frame': ms / F, 'counted_tflops': sum(f for _, _, f in ev) / (ms * 1e-3) / 1e12}
_hip.ASYNC_WGRAD = False            # everything on one stream: no sharing, the serial sum of the kernel times
out['hip_one_stream_ms_per_frame'] = clock(hip_step) / F
_hip.ASYNC_WGRAD = True
out['miopen_ms_per_frame'] = clock(module_step, n=5, warm=2) / F
print(json.dumps(out))
if TAG:
    with open(os.path.join(REPO, 'profiles', '%s_rpn_timing.json' % TAG), 'w') as fh:
        json.dump(out, fh, indent=1)
