"""Host cost of one eager launch through the Python -> ctypes -> HIP path (the eager pretraining step and the navigator loop are
host-bound): wall time per call of a tiny GEMM / LayerNorm issued back to back without synchronising, and a cProfile breakdown."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import lib as L  # noqa: E402
from magic_amd.host import ops as O  # noqa: E402

dev = "cuda"
M, N, K = 256, 128, 128
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
W = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
b = torch.zeros(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)


def loop(n):
    for _ in range(n):
        O.linear_fwd(x, W, b, M, out=out)


def raw(n):
    fn = L.load().magic_gemm
    args = (1, 0, 1, 1, M, N, K, x.data_ptr(), K, 0, 0, W.data_ptr(), K, 0, 0, out.data_ptr(), N, 0, 0, 0, 0, b.data_ptr(), 0, None, 0, None, 0,
            None, 0, 1.0, 1, None, L.stream())
    for _ in range(n):
        fn(*args)


def alloc(n):
    for _ in range(n):
        torch.empty(M, N, device=dev, dtype=torch.bfloat16)


for name, f in (("ops.linear_fwd", loop), ("raw ctypes magic_gemm", raw), ("torch.empty", alloc), ("L.stream()", lambda n: [L.stream() for _ in range(n)]),
                ("torch.matmul", lambda n: [torch.matmul(x, W.t(), out=out) for _ in range(n)])):
    f(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f(3000)
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name:28s} {dt / 3000 * 1e6:7.2f} us per call (host, async)")
cProfile.run("loop(3000)", "/tmp/hl.prof")
pstats.Stats("/tmp/hl.prof").sort_stats("tottime").print_stats(12)
