"""What the embedding-LayerNorm backwards of the headline step are made of (isolated back-to-back launches): the text embedding's backward
(M = 3840 rows, word / position / token-type table gradients) and the panorama embedding's (M ~ 10.7 k rows, nav-type + token-type)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import ops as O  # noqa: E402
from gemm_tile_sweep import timed  # noqa: E402

dev = "cuda"
H, B, L, V = 128, 48, 80, 50265
M = B * L
g = torch.Generator().manual_seed(0)
lens = torch.randint(20, 81, (B,), generator=g)
ids = torch.randint(4, V, (B, L), generator=g)
ids[torch.arange(L)[None, :] >= lens[:, None]] = 0
ids = ids.reshape(-1).int().to(dev)
dy = torch.randn(M, H, device=dev, dtype=torch.bfloat16)
y = torch.randn(M, H, device=dev, dtype=torch.bfloat16)
gamma, beta, rstd = torch.ones(H, device=dev), torch.zeros(H, device=dev), torch.ones(M, device=dev)
dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
dword, dpos, dtt = torch.zeros(V, H, device=dev), torch.zeros(130, H, device=dev), torch.zeros(2, H, device=dev)
word, pos, tt = (ids, 0, 0, dword, 0), (None, L, 2, dpos, 0), (None, 0, 0, dtt, 0)
for name, tabs in (("word+pos+type", (word, pos, tt)), ("no tables", (None, None, None)), ("word only", (word, None, None)), ("pos only", (pos, None, None)),
                   ("type only", (tt, None, None)), ("pos+type", (pos, tt, None))):
    t = timed(lambda: O.ln_bwd(M, H, dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dx=None, dgamma=dg, dbeta=db, dtabs=tabs))
    print(f"text emb bwd M={M}: {name:14s} {t:6.2f} us", flush=True)
ids_nopad = torch.randint(4, V, (M,), generator=g).int().to(dev)
t = timed(lambda: O.ln_bwd(M, H, dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dx=None, dgamma=dg, dbeta=db, dtabs=((ids_nopad, 0, 0, dword, 0), None, None)))
print(f"text emb bwd M={M}: word only, no repeated id {t:6.2f} us", flush=True)

Mp = 290 * 37
dyp = torch.randn(Mp, H, device=dev, dtype=torch.bfloat16)
yp = torch.randn(Mp, H, device=dev, dtype=torch.bfloat16)
rp = torch.ones(Mp, device=dev)
nav = torch.randint(0, 3, (Mp,), generator=g).int().to(dev)
dnav = torch.zeros(3, H, device=dev)
dx = torch.empty(Mp, H, device=dev, dtype=torch.bfloat16)
for name, tabs in (("nav+type", ((nav, 0, 0, dnav, 1), tt, None)), ("no tables", (None, None, None))):
    t = timed(lambda: O.ln_bwd(Mp, H, dyp, y=yp, gamma=gamma, beta=beta, rstd=rp, dx=dx, dgamma=dg, dbeta=db, dtabs=tabs))
    print(f"pano emb bwd M={Mp}: {name:14s} {t:6.2f} us", flush=True)
t = timed(lambda: O.ln_bwd(Mp, H, dyp, y=yp, gamma=gamma, beta=beta, rstd=rp, dx=dx))
print(f"pano emb bwd M={Mp}: no tables, no gamma/beta grads {t:6.2f} us", flush=True)
t = timed(lambda: O.ln_bwd(M, H, dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dx=None, dgamma=dg, dbeta=db, dtabs=(word, pos, tt), hot0=0))
print(f"text emb bwd M={M}: word+pos+type, padding row reduced per workgroup (hot0=0) {t:6.2f} us", flush=True)
for Mk, Kin in ((Mp, 7), (48 * 37, 14), (48 * 30, 7)):
    xk = torch.randn(Mk, Kin, device=dev)
    dyk = torch.randn(Mk, H, device=dev, dtype=torch.bfloat16)
    yk = torch.randn(Mk, H, device=dev, dtype=torch.bfloat16)
    rk = torch.ones(Mk, device=dev)
    dW, dbk = torch.zeros(H, Kin, device=dev), torch.zeros(H, device=dev)
    t = timed(lambda: O.smallk_ln_bwd(Mk, H, Kin, xk, dyk, yk, gamma, beta, rk, dW, dbk, dg, db))
    print(f"smallk_ln_bwd M={Mk} K={Kin}: {t:6.2f} us", flush=True)
