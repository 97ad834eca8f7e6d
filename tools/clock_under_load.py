"""Runs one conv kernel in a loop for ~8 s and samples sclk / power from rocm-smi meanwhile -- developer tool.
usage: clock_under_load.py [f32|bf16x3|idle]"""
import os, sys, subprocess, threading, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip
mode = sys.argv[1] if len(sys.argv) > 1 else 'f32'
dev = torch.device('cuda')
H, W = 352, 400
x = torch.randn((5, H, W, 64), device=dev)
w = torch.randn((64, 64, 3, 3, 3), device=dev) * 0.02
b = torch.zeros(64, device=dev)
split = mode == 'bf16x3'
wpk = _hip.conv3d_pack(w, False, split=split)
samples = []
stop = False

def sampler():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=10).stdout
            keep = [l.strip() for l in out.splitlines() if 'sclk' in l or 'Power' in l or 'mclk' in l]
            samples.append(' | '.join(keep))
        except Exception as e:
            samples.append('smi failed: %r' % e)
        time.sleep(1.0)

t = threading.Thread(target=sampler)
t.start()
t0 = time.time()
n = 0
while time.time() - t0 < 8:
    if mode != 'idle':
        for _ in range(50):
            _hip.conv3d_forward(x, wpk, b, 64, 1, 0, split=split)
        n += 50
    torch.cuda.synchronize()
stop = True
t.join()
print('launches', n, 'avg ms', (time.time() - t0) * 1e3 / max(n, 1))
for s in samples:
    print(s)
