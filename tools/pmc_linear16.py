"""fcn1-shaped row GEMM (80,000 x 768 -> 768) in fp16x3 for `rocprofv3 --pmc` passes -- developer tool."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip
dev = torch.device('cuda')
R, K, N = 80000, 768, 768
x = torch.randn((R, K), device=dev)
w = torch.randn((N, K), device=dev) * 0.03
b = torch.zeros(N, device=dev)
for _ in range(4):
    _hip.linear_forward(x, w, b, relu=True, want_stats=True, split=4)
torch.cuda.synchronize()
