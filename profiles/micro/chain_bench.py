import sys; sys.path.insert(0, '.')
import torch, magic_amd
from magic_amd.host import ops as O
DEV='cuda'; dt=torch.bfloat16
M,H=3840,128
def mk(n,k): return (torch.randn(n,k,device=DEV)*0.05).to(dt), torch.zeros(n,device=DEV)
L=6
Wqkv=[mk(3*H,H) for _ in range(L)]; Wo=[mk(H,H) for _ in range(L)]; W1=[mk(4*H,H) for _ in range(L)]; W2=[mk(H,4*H) for _ in range(L)]
x0=torch.randn(M,H,device=DEV).to(dt)
bufs=dict(qkv=torch.empty(M,3*H,device=DEV,dtype=dt), o=torch.empty(M,H,device=DEV,dtype=dt), f1=torch.empty(M,4*H,device=DEV,dtype=dt), f2=torch.empty(M,H,device=DEV,dtype=dt))
def chain(fresh):
    x=x0; n=0
    for l in range(L):
        qkv = torch.empty(M,3*H,device=DEV,dtype=dt) if fresh else bufs['qkv']
        O.linear_fwd(x,Wqkv[l][0],Wqkv[l][1],M,out=qkv)
        o = torch.empty(M,H,device=DEV,dtype=dt) if fresh else bufs['o']
        O.linear_fwd(qkv,Wo[l][0],Wo[l][1],M,out=o,lda=3*H,residual=x)
        f1 = torch.empty(M,4*H,device=DEV,dtype=dt) if fresh else bufs['f1']
        O.linear_fwd(o,W1[l][0],W1[l][1],M,out=f1,epilogue=1)
        f2 = torch.empty(M,H,device=DEV,dtype=dt) if fresh else bufs['f2']
        O.linear_fwd(f1,W2[l][0],W2[l][1],M,out=f2,residual=o)
        x=f2; n+=4
    return n
def same():
    for _ in range(24): O.linear_fwd(x0,Wo[0][0],Wo[0][1],M,out=bufs['o'])
    return 24
def timeg(fn, *a):
    fn(*a); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): n=fn(*a)
    g.replay(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/50/n*1e3
print("graph same-kernel x24 (txt o 3840x128x128): %.2f us/kernel" % timeg(same))
print("graph dependent chain, reused buffers: %.2f us/kernel" % timeg(chain, False))
print("graph dependent chain, fresh buffers : %.2f us/kernel" % timeg(chain, True))
# LN in chain
g_=torch.ones(H,device=DEV); b_=torch.zeros(H,device=DEV); r=torch.empty(M,device=DEV)
def lnchain():
    x=x0
    for i in range(24):
        y=torch.empty(M,H,device=DEV,dtype=dt); O.ln_fwd(M,H,y,in0=x,gamma=g_,beta=b_,rstd=r); x=y
    return 24
print("graph LN chain: %.2f us/kernel" % timeg(lnchain))
