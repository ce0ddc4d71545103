import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
import torch, magic_amd
from magic_amd.host import ops as O
dev = "cuda"
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def graph_time(fn, n=100):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in ((3840, 384, 128), (3840, 512, 128), (3840, 128, 512), (9720, 384, 128), (9720, 128, 768), (3840, 768, 256), (3840, 1024, 256), (360, 50272, 128)):
    x = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); b = torch.zeros(N, device=dev)
    bb = b.bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_mine = graph_time(lambda: O.linear_fwd(x, W, b, M, out=y))
    t_torch = graph_time(lambda: torch.nn.functional.linear(x, W, bb))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: ours {t_mine:.1f} us ({fl/t_mine/1e6:.0f} TF/s) | torch/hipBLASLt {t_torch:.1f} us ({fl/t_torch/1e6:.0f} TF/s)")
