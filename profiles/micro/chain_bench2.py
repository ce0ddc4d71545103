import sys; sys.path.insert(0, '.')
import torch, magic_amd
from magic_amd.host import ops as O
DEV='cuda'; dt=torch.bfloat16
M,H=3840,128
def mk(n,k): return (torch.randn(n,k,device=DEV)*0.05).to(dt), torch.zeros(n,device=DEV)
L=6
Wo=[mk(H,H) for _ in range(24)]
x0=torch.randn(M,H,device=DEV).to(dt)
bufs=[torch.empty(M,H,device=DEV,dtype=dt) for _ in range(25)]
def run(dep, samew, big_between=False, junk=None):
    x=x0; n=0
    for i in range(24):
        W=Wo[0] if samew else Wo[i]
        out=bufs[i+1]
        O.linear_fwd(x if dep else x0, W[0], W[1], M, out=out)
        if dep: x=out
        n+=1
    return n
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
print("independent, same W : %.2f" % timeg(run, False, True))
print("independent, diff W : %.2f" % timeg(run, False, False))
print("dependent,   same W : %.2f" % timeg(run, True, True))
print("dependent,   diff W : %.2f" % timeg(run, True, False))
# dependent with residual epilogue + different shapes
W1=[mk(4*H,H) for _ in range(12)]; W2=[mk(H,4*H) for _ in range(12)]
f1=[torch.empty(M,4*H,device=DEV,dtype=dt) for _ in range(12)]
def ffn():
    x=x0
    for i in range(12):
        O.linear_fwd(x,W1[i][0],W1[i][1],M,out=f1[i],epilogue=1)
        O.linear_fwd(f1[i],W2[i][0],W2[i][1],M,out=bufs[i],residual=x)
        x=bufs[i]
    return 24
print("ffn chain (gelu, residual): %.2f" % timeg(ffn))
def ffn_nores():
    x=x0
    for i in range(12):
        O.linear_fwd(x,W1[i][0],W1[i][1],M,out=f1[i])
        O.linear_fwd(f1[i],W2[i][0],W2[i][1],M,out=bufs[i])
        x=bufs[i]
    return 24
print("ffn chain (plain): %.2f" % timeg(ffn_nores))
