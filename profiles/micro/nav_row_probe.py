"""navigator-step row kernels at MAGIC-L width in isolation (graph-replayed, back to back = warm instruction cache) and interleaved with a GEMM
(as in the step's chain): LayerNorm forward / backward at M = 624, H = 768 with and without the parameter gradients."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import ops as O  # noqa: E402
from gemm_tile_sweep import timed  # noqa: E402

H = 768
bf = torch.bfloat16
for M in (624, 1024, 8192):
    dy = torch.randn(M, H, device="cuda", dtype=bf)
    y = torch.randn(M, H, device="cuda", dtype=bf)
    x = torch.randn(M, H, device="cuda", dtype=bf)
    R = torch.randn(M, H, device="cuda", dtype=bf)
    gamma, beta, rstd = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda"), torch.ones(M, device="cuda")
    dx = torch.empty(M, H, device="cuda", dtype=bf)
    out = torch.empty(M, H, device="cuda", dtype=bf)
    dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    W = torch.randn(H, H, device="cuda", dtype=bf) * 0.05
    b = torch.zeros(H, device="cuda")
    g_out = torch.empty(M, H, device="cuda", dtype=bf)
    t_plain = timed(lambda: O.ln_bwd(M, H, dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dx=dx))
    t_atom = timed(lambda: O.ln_bwd(M, H, dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dx=dx, dgamma=dg, dbeta=db))
    O.DEFER["active"] = True
    t_part = timed(lambda: O.ln_bwd(M, H, dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dx=dx, dgamma=dg, dbeta=db))
    O.PART_JOBS.clear()
    t_gemm = timed(lambda: O.linear_fwd(x, W, b, M, out=g_out))

    def both():
        O.linear_fwd(x, W, b, M, out=g_out)
        O.ln_bwd(M, H, dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dx=dx, dgamma=dg, dbeta=db)
    t_both = timed(both)
    O.PART_JOBS.clear()
    O.DEFER["active"] = False
    rs = torch.empty(M, device="cuda")
    t_fwd = timed(lambda: O.ln_fwd(M, H, out, in0=x, in1=R, gamma=gamma, beta=beta, rstd=rs))
    print(f"M={M:5d}: ln_bwd no-pgrad {t_plain:6.2f}  atomics {t_atom:6.2f}  partial rows {t_part:6.2f} | gemm {t_gemm:6.2f}  gemm+ln_bwd(partial) {t_both:6.2f} | ln_fwd {t_fwd:6.2f} us", flush=True)

# ---- in-situ form: a dependent chain over 16 distinct buffer / weight sets (nothing is L2-warm, every launch consumes its predecessor's output)
M, NS = 624, 16
xs = [torch.randn(M, H, device="cuda", dtype=bf) for _ in range(NS)]
ys = [torch.randn(M, H, device="cuda", dtype=bf) for _ in range(NS)]
dys = [torch.empty(M, H, device="cuda", dtype=bf) for _ in range(NS)]
dxs = [torch.empty(M, H, device="cuda", dtype=bf) for _ in range(NS)]
Ws = [torch.randn(H, H, device="cuda", dtype=bf) * 0.05 for _ in range(NS)]
dgs = [(torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")) for _ in range(NS)]
outs = [torch.empty(M, H, device="cuda", dtype=bf) for _ in range(NS)]
rss = [torch.empty(M, device="cuda") for _ in range(NS)]
gamma, beta, rstd = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda"), torch.ones(M, device="cuda")
b = torch.zeros(H, device="cuda")


def chain(kind):
    src = xs[0]
    for i in range(NS):
        O.linear_fwd(src, Ws[i], b, M, out=dys[i])
        if kind == "gemm":
            src = dys[i]
        elif kind == "lnb":
            O.ln_bwd(M, H, dys[i], y=ys[i], gamma=gamma, beta=beta, rstd=rstd, dx=dxs[i], dgamma=dgs[i][0], dbeta=dgs[i][1])
            src = dxs[i]
        elif kind == "lnb_nopg":
            O.ln_bwd(M, H, dys[i], y=ys[i], gamma=gamma, beta=beta, rstd=rstd, dx=dxs[i])
            src = dxs[i]
        elif kind == "lnf":
            O.ln_fwd(M, H, outs[i], in0=dys[i], in1=ys[i], gamma=gamma, beta=beta, rstd=rss[i])
            src = outs[i]


O.DEFER["active"] = True
res = {}
for kind in ("gemm", "lnb", "lnb_nopg", "lnf"):
    res[kind] = timed(lambda: chain(kind), reps=4) / NS
    O.PART_JOBS.clear()
O.DEFER["active"] = False
print(f"dependent chain, cold operands (per pair): gemm {res['gemm']:.2f} us; + ln_bwd(partial) {res['lnb'] - res['gemm']:.2f}; + ln_bwd(no pgrad) {res['lnb_nopg'] - res['gemm']:.2f}; "
      f"+ ln_fwd {res['lnf'] - res['gemm']:.2f} us", flush=True)
