"""magic_cfp_loss launch time (B = 48, H = 128, bf16), losses only vs losses + gradients, graph-replayed back to back"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import ops as O  # noqa: E402
from gemm_tile_sweep import timed  # noqa: E402

B, H = 48, 128
a = [torch.randn(B, H, device="cuda").bfloat16() for _ in range(3)]
txt = torch.randn(B, H, device="cuda").bfloat16()
rows = torch.empty(6, B, device="cuda")
d_a = [torch.empty(B, H, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
d_txt = torch.empty(B, H, device="cuda", dtype=torch.bfloat16)
O.cfp_loss(B, H, a, txt, 0.7, 0.01, rows, d_a=d_a, d_txt=d_txt)
print(f"losses only        : {timed(lambda: O.cfp_loss(B, H, a, txt, 0.7, 0.01, rows)):6.2f} us")
print(f"losses + gradients : {timed(lambda: O.cfp_loss(B, H, a, txt, 0.7, 0.01, rows, d_a=d_a, d_txt=d_txt)):6.2f} us")
