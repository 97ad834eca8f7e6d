import torch, time
dev='cuda'
R,K,N=80000,768,768
x=torch.randn(R,K,device=dev); w=torch.randn(N,K,device=dev)*0.03; dz=torch.randn(R,N,device=dev)
def clock(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
fl=2.0*R*K*N
for name,fn in (('fwd x@W^T', lambda: torch.matmul(x,w.t())), ('dgrad dz@W', lambda: torch.matmul(dz,w)), ('wgrad dz^T@x', lambda: torch.matmul(dz.t(),x))):
    t=clock(fn); print('%-14s %.3f ms  %.1f TFLOP/s (%.2f of 157.3)'%(name,t,fl/t/1e9,fl/t/1e9/157.3))
for R2,K2,N2 in ((80000,768,128),(80000,128,768),(100000,128,128)):
    a=torch.randn(R2,K2,device=dev); b=torch.randn(N2,K2,device=dev)
    t=clock(lambda: torch.matmul(a,b.t())); f2=2.0*R2*K2*N2
    print('fwd %dx%dx%d %.3f ms %.1f TF'%(R2,K2,N2,t,f2/t/1e9))
