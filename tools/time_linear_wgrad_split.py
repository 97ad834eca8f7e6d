import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip
dev='cuda'
def clock(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for R,K,N in ((79700,768,768),(79700,768,128),(79700,128,128),(100000,128,128),(140800,128,1024),(140800,256,4096)):
    x=torch.randn(R,K,device=dev); dz=torch.randn(R,N,device=dev)
    a=clock(lambda: _hip.linear_wgrad(x,dz,split=False)); b=clock(lambda: _hip.linear_wgrad(x,dz,split=True))
    fl=2.0*R*K*N
    print('%6d x %4d x %4d  f32 %.3f ms (%.0f TF)   bf16x3 %.3f ms (%.0f TF algorithmic, %.2f of the bf16 peak executed)'%(R,K,N,a,fl/a/1e9,b,fl/b/1e9,3*fl/b/1e9/2500))
