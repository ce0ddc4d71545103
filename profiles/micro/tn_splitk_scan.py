"""weight-gradient GEMM dW[N,K] += dY[M,N]^T X[M,K] at MAGIC-L widths: us per launch against split-K, 64x64 tile vs wide tile, vs torch"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import lib as L  # noqa: E402
from magic_amd.host import ops as O  # noqa: E402
from gemm_tile_sweep import timed  # noqa: E402

for M, N, K in ((8192, 768, 768), (8192, 2304, 768), (8192, 3072, 768), (8192, 768, 3072), (3840, 768, 768), (10440, 768, 768), (3840, 3072, 768)):
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    dW = torch.zeros(N, K, device="cuda")
    db = torch.zeros(N, device="cuda")
    row = []
    for mode in (0, 2):
        L.call("magic_gemm_set_big", mode)
        for sk in (1, 2, 4, 8, 16):
            t = timed(lambda: O.gemm(2, dy, x, dW, N, K, M, N, K, K, splitk=sk, accumulate=True, bias_grad=db))
            row.append(f"{t:6.1f}")
    L.call("magic_gemm_set_big", 0)
    tt = timed(lambda: torch.matmul(dy.t(), x))
    print(f"M={M:6d} dW {N:5d}x{K:5d} | 64-tile sk1,2,4,8,16: {' '.join(row[:5])} | wide: {' '.join(row[5:])} | torch {tt:6.1f}", flush=True)
