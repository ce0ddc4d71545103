import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
import torch, magic_amd
from types import SimpleNamespace
from magic_amd.host import ops as O
dev = "cuda"
def lin(N, K): return SimpleNamespace(W=(torch.randn(N, K, device=dev) * 0.05).bfloat16(), b=torch.zeros(N, device=dev), N=N, K=K)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for H, M in ((128, 3840), (128, 9720), (256, 3840), (256, 9720)):
    I = 4 * H
    dt = torch.bfloat16
    ctx, x = torch.randn(M, H, device=dev).to(dt), torch.randn(M, H, device=dev).to(dt)
    lo, l1, l2, l4 = lin(H, H), lin(I, H), lin(H, I), lin(3 * H, H)
    g1 = torch.ones(H, device=dev); b1 = torch.zeros(H, device=dev)
    a, z, g, out, qn = (torch.empty(M, n_, dtype=dt, device=dev) for n_ in (H, I, I, H, 3 * H))
    r1, r2 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    st4 = [O.rb_ln(lo, x, g1, b1, 1e-12, a, r1), O.rb_act(l1, g, pre=z), O.rb_ln(l2, None, g1, b1, 1e-12, out, r2, res_stage=0), O.rb_lin(l4, qn)]
    st2 = [O.rb_ln(lo, x, g1, b1, 1e-12, a, r1), O.rb_lin(lo, out)]
    t4 = timeit(lambda: O.rowblock_fwd(ctx, M, st4))
    t2 = timeit(lambda: O.rowblock_fwd(ctx, M, st2))
    def sep():
        O.linear_ln(ctx, lo.W, lo.b, M, x, g1, b1, 1e-12, a, r1)
        O.linear_fwd(a, l1.W, l1.b, M, epilogue=1, pre=z, out=g)
        if O.linear_ln_ok(H, I): O.linear_ln(g, l2.W, l2.b, M, a, g1, b1, 1e-12, out, r2)
        else:
            fo = O.linear_fwd(g, l2.W, l2.b, M, residual=a); O.ln_fwd(M, H, out, in0=fo, gamma=g1, beta=b1, eps=1e-12, rstd=r2)
        O.linear_fwd(out, l4.W, l4.b, M, out=qn)
    ts = timeit(sep)
    def sep2():
        O.linear_ln(ctx, lo.W, lo.b, M, x, g1, b1, 1e-12, a, r1)
        O.linear_fwd(a, lo.W, lo.b, M, out=out)
    ts2 = timeit(sep2)
    print(f"H={H} M={M}: rowblock 4-stage {t4:.1f} us | separate (4-5 launches) {ts:.1f} us || rowblock 2-stage {t2:.1f} us | separate 2 launches {ts2:.1f} us")
