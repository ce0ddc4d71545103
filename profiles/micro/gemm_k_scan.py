"""time of one GEMM launch against K at fixed M, N (both tile modes in one process): separates the fixed cost of a launch from
the cost per 64-deep K-step.  python profiles/micro/gemm_k_scan.py [M N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import lib as L  # noqa: E402
from magic_amd.host import ops as O  # noqa: E402
from gemm_tile_sweep import timed  # noqa: E402

M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 2048)
print(f"M={M} N={N}   us per launch: K | NT64 NT128 | NN64 NN128 | TN64 TN128 (contraction over K) | torch NT")
for K in (64, 128, 256, 512, 1024, 2048, 4096):
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    W = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    Wn = torch.randn(K, N, device="cuda", dtype=torch.bfloat16) * 0.05
    xt = torch.randn(K, M, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out32 = torch.zeros(M, N, device="cuda")
    r = []
    for mode in (0, 2):
        L.call("magic_gemm_set_big", mode)
        nt = timed(lambda: O.gemm(0, x, W, out, M, N, K, K, K, N))
        nn = timed(lambda: O.gemm(1, x, Wn, out, M, N, K, K, N, N))
        tn = timed(lambda: O.gemm(2, xt, Wn, out32, M, N, K, M, N, N, splitk=1, accumulate=True))
        r.append((nt, nn, tn))
    L.call("magic_gemm_set_big", 0)
    t = timed(lambda: torch.matmul(x, W.t(), out=out))
    print(f"{K:5d} | {r[0][0]:7.1f} {r[1][0]:7.1f} | {r[0][1]:7.1f} {r[1][1]:7.1f} | {r[0][2]:7.1f} {r[1][2]:7.1f} | {t:7.1f}", flush=True)
