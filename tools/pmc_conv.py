"""One conv2-shaped forward + dgrad + wgrad launch set, for `rocprofv3 --pmc ...` passes -- developer tool."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip
dev = torch.device('cuda')
H, W = 352, 400
cin, cout, din, sd, pd = 64, 64, 5, 1, 0
dout = _hip.conv_out_depth(din, sd, pd)
x = torch.randn((din, H, W, cin), device=dev)
w = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.02
b = torch.zeros(cout, device=dev)
dz = torch.randn((dout, H, W, cout), device=dev)
split = {'bf16x3': 2, 'bf16x6': 3, 'fp16x3': 4}.get(sys.argv[1] if len(sys.argv) > 1 else 'f32', 0)
wpk, wpd = _hip.conv3d_pack(w, False, split=split), _hip.conv3d_pack(w, True, split=split)
for _ in range(3):
    _hip.conv3d_forward(x, wpk, b, cout, sd, pd, split=split)
    _hip.conv3d_dgrad(dz, wpd, din, cin, sd, pd, split=split)
    _hip.conv3d_wgrad(x, dz, sd, pd, split=split)
torch.cuda.synchronize()
