"""the navigator step's GEMM shapes at MAGIC-L width (a few hundred rows x 768 / 2304 / 3072): csrc/gemm.hip (forward NT, input-gradient NN) against
torch.matmul (hipBLASLt), graph-replayed back to back."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import ops as O  # noqa: E402
from gemm_tile_sweep import timed  # noqa: E402

SHAPES = [(624, 768, 768), (624, 2304, 768), (624, 3072, 768), (624, 768, 3072), (624, 1536, 768), (592, 768, 768), (1024, 768, 768), (1024, 3072, 768),
          (1024, 768, 3072), (256, 768, 768), (16, 768, 768), (8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 2304, 768)]
for M, N, K in SHAPES:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    W = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    Wt = W.t().contiguous()
    b = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    r = dict(M=M, N=N, K=K)
    r["nt_us"] = round(timed(lambda: O.linear_fwd(x, W, b, M, out=out)), 2)
    r["nn_us"] = round(timed(lambda: O.linear_dx(dy, W, M, out=dx)), 2)
    r["torch_nt_us"] = round(timed(lambda: torch.matmul(x, W.t(), out=out)), 2)
    r["torch_nn_us"] = round(timed(lambda: torch.matmul(dy, W, out=dx)), 2)
    r["nt_tflops"] = round(2.0 * M * N * K / r["nt_us"] / 1e6, 1)
    print(json.dumps(r), flush=True)
