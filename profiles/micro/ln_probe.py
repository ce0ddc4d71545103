"""LayerNorm forward / backward launch times at the navigator-step shape (M ~ 600 rows, H = 768) and the headline shapes (H = 128):
with and without the gamma/beta gradient atomics and the dropout masks.

Measured (isolated, back-to-back launches): ln_bwd 10.1 us at M = 3840, H = 128 vs 4.4 us without the gamma/beta gradients; 17.3 vs 5.8 us
at M = 10440.  Two follow-ups were built on this and NOT kept: (1) a scratch-row mode (every workgroup stores its column sums, one grouped
reduction at the end of backward; both LayerNorm-backward kernels, full parity green) left the replayed training step at 2.91 ms -- inside
the step these atomics are evidently not on the critical chain the way the isolated launches suggest; (2) a smaller grid cap for H <= 256
(192 workgroups) helped only in this isolated measurement."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import ops as O  # noqa: E402
from gemm_tile_sweep import timed  # noqa: E402

seed = torch.tensor([3, 4], dtype=torch.int32, device="cuda")
for M, H in ((592, 768), (8000, 768), (3840, 128), (10440, 128), (1776, 128)):
    dy = torch.randn(M, H, device="cuda", dtype=torch.bfloat16)
    y = torch.randn(M, H, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(M, H, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(M, H, device="cuda", dtype=torch.bfloat16)
    gamma, beta, rstd = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda"), torch.ones(M, device="cuda")
    dx = torch.empty(M, H, device="cuda", dtype=torch.bfloat16)
    dxm = torch.empty(M, H, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, H, device="cuda", dtype=torch.bfloat16)
    dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    t_b = timed(lambda: O.ln_bwd(M, H, dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dx=dx, dgamma=dg, dbeta=db))
    t_bn = timed(lambda: O.ln_bwd(M, H, dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dx=dx))
    t_bd = timed(lambda: O.ln_bwd(M, H, dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dx=dx, dgamma=dg, dbeta=db, drop_dx=(seed, 0.1, 77), dxm=dxm))
    t_f = timed(lambda: O.ln_fwd(M, H, out, in0=x, in1=r, gamma=gamma, beta=beta, rstd=rstd))
    print(f"M={M:6d} H={H:4d}  ln_bwd {t_b:6.2f} us (no gamma/beta grads {t_bn:6.2f}, with dropout mask {t_bd:6.2f})   ln_fwd {t_f:6.2f} us", flush=True)
