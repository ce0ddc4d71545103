"""linear_lnbwd (input-gradient GEMM + LayerNorm backward, H = 128) and linear_ln launch times, with and without the gamma/beta atomics
and the residual operand.  Measured on MI355X: 14.3 us at M = 3840, K = 512 = ~6 us fixed + 0.5 us per K-step + 2.8 us of same-address
gamma/beta atomics (6 us at M >= 8192) + 1.1 us residual.  A 4-K-group variant (as gemm_kg_kernel) was tried and measured 1-2 us SLOWER
(the K loop is not what bounds these launches) -- not kept."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import ops as O  # noqa: E402
from gemm_tile_sweep import timed  # noqa: E402

H = 128
for M, K in ((3840, 512), (3840, 384), (1776, 512), (1776, 384), (816, 512), (8192, 512), (10440, 512)):
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    W = torch.randn(K, H, device="cuda", dtype=torch.bfloat16) * 0.1
    R = torch.randn(M, H, device="cuda", dtype=torch.bfloat16)
    y = torch.randn(M, H, device="cuda", dtype=torch.bfloat16)
    gamma, beta, rstd = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda"), torch.ones(M, device="cuda")
    dx = torch.empty(M, H, device="cuda", dtype=torch.bfloat16)
    dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    t = timed(lambda: O.linear_lnbwd(x, W, M, R, y, gamma, beta, rstd, dx, dg, db))
    t_na = timed(lambda: O.linear_lnbwd(x, W, M, R, y, gamma, beta, rstd, dx, None, None))
    t_nr = timed(lambda: O.linear_lnbwd(x, W, M, None, y, gamma, beta, rstd, dx, None, None))
    out = torch.empty(M, H, device="cuda", dtype=torch.bfloat16)
    rs = torch.empty(M, device="cuda")
    xf = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    Wf = torch.randn(H, K, device="cuda", dtype=torch.bfloat16) * 0.1
    bf = torch.zeros(H, device="cuda")
    t2 = timed(lambda: O.linear_ln(xf, Wf, bf, M, R, gamma, beta, 1e-12, out, rs))
    print(f"M={M:6d} K={K:4d}  linear_lnbwd {t:6.2f} us (no gamma/beta atomics {t_na:6.2f}, and no residual {t_nr:6.2f})   linear_ln {t2:6.2f} us", flush=True)
