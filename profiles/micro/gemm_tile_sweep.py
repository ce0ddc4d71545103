"""64x64 vs 128x128 block tile of csrc/gemm.hip vs torch.matmul (hipBLASLt) on the linear-layer shapes of MAGIC-S/M/L, graph-replayed
back to back.  Run once per tile mode (the choice is read from the environment at first use):
    MAGIC_GEMM_BIG=0 python profiles/micro/gemm_tile_sweep.py ; MAGIC_GEMM_BIG=2 python profiles/micro/gemm_tile_sweep.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import ops as O  # noqa: E402

SHAPES = [  # (M, N, K)
    (3840, 256, 256), (3840, 768, 256), (3840, 1024, 256), (3840, 256, 1024), (10440, 256, 768), (10440, 768, 256), (10440, 1024, 256),
    (600, 768, 768), (600, 2304, 768), (600, 3072, 768), (600, 768, 3072), (8192, 768, 768), (8192, 2304, 768), (8192, 3072, 768),
    (8192, 768, 3072), (8192, 1536, 768), (3840, 384, 384), (3840, 1536, 384), (3840, 384, 1536), (2048, 512, 512), (1100, 3072, 768)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    mode = os.environ.get("MAGIC_GEMM_BIG", "1")
    rows = []
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        W = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
        b = torch.zeros(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
        dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
        dW = torch.zeros(N, K, device="cuda")
        db = torch.zeros(N, device="cuda")
        r = dict(M=M, N=N, K=K, mode=mode)
        r["nt_us"] = round(timed(lambda: O.linear_fwd(x, W, b, M, out=out)), 2)
        r["nn_us"] = round(timed(lambda: O.linear_dx(dy, W, M, out=dx)), 2)
        r["tn_us"] = round(timed(lambda: O.linear_dw(dy, x, dW, db, M)), 2)
        r["torch_nt_us"] = round(timed(lambda: torch.matmul(x, W.t(), out=out)), 2)
        r["torch_tn_us"] = round(timed(lambda: torch.matmul(dy.t(), x)), 2)
        r["nt_tflops"] = round(2.0 * M * N * K / r["nt_us"] / 1e6, 1)
        print(json.dumps(r), flush=True)
        rows.append(r)


if __name__ == "__main__":
    main()
