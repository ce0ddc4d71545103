"""GPU parity tests of every C-ABI kernel against plain fp32 torch math / the oracle (-m gpu).

fp32 mode (exact-f32 MFMA) is held to tight tolerances; bf16 mode is compared with the same fp32
reference evaluated on bf16-rounded inputs, to a bf16-sized tolerance written next to each check."""
import math

import pytest
import torch
import torch.nn.functional as F

import magic_amd  # noqa: F401
from magic_amd.host import lib as L
from magic_amd.host import ops as O
from oracle import makd_ref as M
from oracle import optim_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def tol(dtype):
    return dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (sum(shape) * 7919 + len(shape)))
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def check(got, want, name, **kw):
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    ok = torch.allclose(got, want, **kw)
    assert ok, f"{name}: max|err|={err:.3e} (ref max {ref:.3e}) tol={kw}"


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(100, 72, 64), (64, 64, 128), (333, 128, 768), (48, 50, 40)])
def test_gemm_nt_bias_epilogues(dtype, M, N, K):
    x, W = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, scale=0.1)
    b = rnd(N, scale=0.5)
    res = rnd(M, N, dtype=dtype)
    ref = x.float() @ W.float().t() + b
    out = O.linear_fwd(x, W, b, M)
    check(out, ref, "nt+bias", **tol(dtype))
    pre = torch.empty(M, N, dtype=dtype, device=DEV)
    out = O.linear_fwd(x, W, b, M, epilogue=1, residual=res, pre=pre)
    check(pre, ref, "pre-activation", **tol(dtype))
    check(out, F.gelu(ref) + res.float(), "gelu+residual", **tol(dtype))
    out = O.linear_fwd(x, W, b, M, epilogue=2)
    check(out, F.relu(ref), "relu", **tol(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_nn_dx_with_dgelu_and_residual(dtype):
    M, N, K = 150, 256, 128          # dx[M,K] = dy[M,N] @ W[N,K]
    dy, W = rnd(M, N, dtype=dtype), rnd(N, K, dtype=dtype, scale=0.1)
    z = rnd(M, K, dtype=dtype)
    r = rnd(M, K, dtype=dtype)
    ref = dy.float() @ W.float()
    check(O.linear_dx(dy, W, M), ref, "nn", **tol(dtype))
    zz = z.float().requires_grad_(True)
    F.gelu(zz).backward(torch.ones_like(zz))
    check(O.linear_dx(dy, W, M, epilogue=3, aux=z, residual=r), ref * zz.grad + r.float(), "nn+dgelu+res", **tol(dtype))
    y = rnd(M, K, dtype=dtype)
    check(O.linear_dx(dy, W, M, epilogue=4, aux=y), ref * (y.float() > 0), "nn+drelu", **tol(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1000, 128, 128), (77, 72, 40), (3840, 128, 512)])
def test_gemm_tn_dw_splitk_and_bias_grad(dtype, M, N, K):
    dy, x = rnd(M, N, dtype=dtype), rnd(M, K, dtype=dtype)
    dW = torch.zeros(N, K, device=DEV)
    db = torch.zeros(N, device=DEV)
    O.linear_dw(dy, x, dW, db, M)
    O.linear_dw(dy, x, dW, db, M)                       # accumulates
    t = dict(rtol=1e-4, atol=1e-3) if dtype == torch.float32 else dict(rtol=2e-2, atol=5e-2 * math.sqrt(M / 64))
    check(dW, 2 * dy.float().t() @ x.float(), "dW", **t)
    check(db, 2 * dy.float().sum(0), "db", **t)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_batched_attention_shapes(dtype):
    B, nh, N, d, H = 3, 2, 37, 64, 128
    ldp = (N + 7) // 8 * 8
    qkv = rnd(B * N, 3 * H, dtype=dtype)
    S = torch.zeros(B, nh, N, ldp, device=DEV)
    O.gemm(0, qkv, qkv[:, H:], S, N, N, d, 3 * H, 3 * H, ldp, batch=B * nh, nh=nh,
           sA=(N * 3 * H, d), sB=(N * 3 * H, d), sC=(nh * N * ldp, N * ldp))
    q = qkv[:, :H].float().view(B, N, nh, d).transpose(1, 2)
    k = qkv[:, H:2 * H].float().view(B, N, nh, d).transpose(1, 2)
    v = qkv[:, 2 * H:].float().view(B, N, nh, d).transpose(1, 2)
    t = tol(dtype) if dtype == torch.float32 else dict(rtol=2e-2, atol=1e-1)
    check(S[..., :N], q @ k.transpose(-1, -2), "QK^T", **t)
    assert (S[..., N:] == 0).all()
    Pm = torch.zeros(B, nh, N, ldp, device=DEV, dtype=dtype)
    Pm[..., :N] = torch.softmax(S[..., :N] / 8, -1).to(dtype)
    ctx = torch.empty(B * N, H, dtype=dtype, device=DEV)
    O.gemm(1, Pm, qkv[:, 2 * H:], ctx, N, d, N, ldp, 3 * H, H, batch=B * nh, nh=nh,
           sA=(nh * N * ldp, N * ldp), sB=(N * 3 * H, d), sC=(N * H, d))
    ref = (Pm[..., :N].float() @ v).transpose(1, 2).reshape(B * N, H)
    check(ctx, ref, "PV", **tol(dtype))
    # dV = P^T dO (TN, direct store)
    dO = rnd(B * N, H, dtype=dtype)
    dqkv = torch.zeros(B * N, 3 * H, dtype=dtype, device=DEV)
    O.gemm(2, Pm, dO, dqkv[:, 2 * H:], N, d, N, ldp, H, 3 * H, batch=B * nh, nh=nh,
           sA=(nh * N * ldp, N * ldp), sB=(N * H, d), sC=(N * 3 * H, d))
    dOh = dO.float().view(B, N, nh, d).transpose(1, 2)
    ref = (Pm[..., :N].float().transpose(-1, -2) @ dOh).transpose(1, 2).reshape(B * N, H)
    check(dqkv[:, 2 * H:], ref, "dV", **tol(dtype))


# wide-tile path (csrc/gemm.hip gemm_wide_kernel: 128x128, LDS-DMA staging, swizzled LDS images); forced on for every eligible shape here
# (by default only large forward / input-gradient problems take it)
@pytest.fixture
def big_tiles():
    from magic_amd.host import lib as L
    L.call("magic_gemm_set_big", 2)
    yield
    L.call("magic_gemm_set_big", 1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_big_tile_all_layouts_and_epilogues(dtype, big_tiles):
    M, N, K = 1301, 1208, 328                     # 11 x 10 tiles, ragged edges in every dimension, K not a tile multiple
    x, W = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, scale=0.1)
    b, res = rnd(N, scale=0.5), rnd(M, N, dtype=dtype)
    ref = x.float() @ W.float().t() + b
    t = tol(dtype) if dtype == torch.float32 else dict(rtol=2e-2, atol=6e-2)
    check(O.linear_fwd(x, W, b, M), ref, "big nt+bias", **t)
    pre = torch.empty(M, N, dtype=dtype, device=DEV)
    out = O.linear_fwd(x, W, b, M, epilogue=1, residual=res, pre=pre)
    check(pre, ref, "big pre-activation", **t)
    check(out, F.gelu(ref) + res.float(), "big gelu+residual", **t)
    # NN: dx[M,K2] = dy[M,N] @ W2[N,K2] with dgelu and residual
    K2 = 1160
    dy, W2 = rnd(M, N, dtype=dtype, scale=0.3), rnd(N, K2, dtype=dtype, scale=0.1)
    z, r = rnd(M, K2, dtype=dtype), rnd(M, K2, dtype=dtype)
    refx = dy.float() @ W2.float()
    tn = tol(dtype) if dtype == torch.float32 else dict(rtol=2e-2, atol=1.5e-1)
    check(O.linear_dx(dy, W2, M), refx, "big nn", **tn)
    zz = z.float().requires_grad_(True)
    F.gelu(zz).backward(torch.ones_like(zz))
    check(O.linear_dx(dy, W2, M, epilogue=3, aux=z, residual=r), refx * zz.grad + r.float(), "big nn+dgelu+res", **tn)
    # TN: dW[N,K2] += dy^T x2, bias grad, accumulate twice (split-K chosen by the host heuristic)
    x2 = rnd(M, K2, dtype=dtype)
    dW, db = torch.zeros(N, K2, device=DEV), torch.zeros(N, device=DEV)
    O.linear_dw(dy, x2, dW, db, M)
    O.linear_dw(dy, x2, dW, db, M)
    tw = dict(rtol=1e-4, atol=2e-3) if dtype == torch.float32 else dict(rtol=2e-2, atol=5e-2 * math.sqrt(M / 64))
    check(dW, 2 * dy.float().t() @ x2.float(), "big dW", **tw)
    check(db, 2 * dy.float().sum(0), "big db", **tw)


# ------------------------------------------------------------------------------------ LayerNorm family
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H", [128, 256])
def test_ln_fwd_bwd_with_tables(dtype, H):
    M, L_, V = 150, 30, 50
    in0 = rnd(M, H, dtype=dtype)
    word, pos, typ, nav = rnd(V, H, dtype=dtype), rnd(L_ + 2, H, dtype=dtype), rnd(1, H, dtype=dtype), rnd(3, H, dtype=dtype)
    ids = torch.randint(0, V, (M,), device=DEV, dtype=torch.int32)
    navi = torch.randint(0, 3, (M,), device=DEV, dtype=torch.int32)
    gamma, beta = (1 + 0.1 * rnd(H)), 0.1 * rnd(H, seed=5)
    dy = rnd(M, H, dtype=dtype, seed=9)
    for case in ("text", "image", "sum"):
        if case == "text":
            tabs = ((word, ids, 0, 0), (pos, None, L_, 2), (typ, None, 0, 0))
            leaves = [t.float().clone().requires_grad_(True) for t in (word, pos, typ)]
            rows = torch.arange(M, device=DEV) % L_ + 2
            x = leaves[0][ids.long()] + leaves[1][rows] + leaves[2][0]
            dense = None
        elif case == "image":
            tabs = ((nav, navi, 0, 0), (typ, None, 0, 0), None)
            leaves = [t.float().clone().requires_grad_(True) for t in (nav, typ)]
            dense = in0.float().clone().requires_grad_(True)
            x = dense + leaves[0][navi.long()] + leaves[1][0]
        else:
            tabs = ((word, ids, 0, 0), None, None)
            leaves = [word.float().clone().requires_grad_(True)]
            dense = in0.float().clone().requires_grad_(True)
            x = dense + leaves[0][ids.long()]
        do_ln = case != "sum"
        g32, b32 = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        ref = F.layer_norm(x, (H,), g32, b32, 1e-12) if do_ln else x
        ref.backward(dy.float())
        out = torch.empty(M, H, dtype=dtype, device=DEV)
        rstd = torch.empty(M, device=DEV)
        O.ln_fwd(M, H, out, in0=in0 if dense is not None else None, tabs=tabs, gamma=gamma, beta=beta, rstd=rstd, do_ln=do_ln)
        check(out, ref, f"ln_fwd[{case}]", **tol(dtype))
        dx = torch.empty(M, H, dtype=dtype, device=DEV)
        dg, dbt = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
        dtabs, dbufs = [], []
        for t in tabs:
            if t is None:
                dtabs.append(None)
                continue
            buf = torch.zeros(t[0].shape, device=DEV)
            dbufs.append(buf)
            dtabs.append((t[1], t[2], t[3], buf, 1 if t[0] is nav else 0))
        O.ln_bwd(M, H, dy, y=out, gamma=gamma, beta=beta, rstd=rstd, dx=dx, dgamma=dg, dbeta=dbt, dtabs=tuple(dtabs), do_ln=do_ln)
        t2 = tol(dtype) if dtype == torch.float32 else dict(rtol=5e-2, atol=8e-2)
        tp = dict(rtol=1e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=0.5)
        if dense is not None:
            check(dx, dense.grad, f"ln_bwd dx[{case}]", **t2)
        if do_ln:
            check(dg, g32.grad, f"dgamma[{case}]", **tp)
            check(dbt, b32.grad, f"dbeta[{case}]", **tp)
        for buf, leaf in zip(dbufs, leaves):
            check(buf, leaf.grad, f"dtable[{case}]", **tp)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Kin", [7, 14])
def test_smallk_ln_fwd_bwd(dtype, Kin):
    M, H = 100, 128
    x = rnd(M, Kin)
    W, b = rnd(H, Kin, scale=0.3).requires_grad_(True), rnd(H, scale=0.1).requires_grad_(True)
    gamma, beta = (1 + 0.1 * rnd(H, seed=1)).requires_grad_(True), (0.1 * rnd(H, seed=2)).requires_grad_(True)
    ref = F.layer_norm(x @ W.t() + b, (H,), gamma, beta, 1e-12)
    dy = rnd(M, H, dtype=dtype, seed=4)
    ref.backward(dy.float())
    out, rstd = torch.empty(M, H, dtype=dtype, device=DEV), torch.empty(M, device=DEV)
    O.smallk_ln_fwd(M, H, Kin, x, W.detach(), b.detach(), gamma.detach(), beta.detach(), 1e-12, out, rstd)
    check(out, ref, "smallk fwd", **tol(dtype))
    dW, db, dg, dbt = torch.zeros(H, Kin, device=DEV), torch.zeros(H, device=DEV), torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    O.smallk_ln_bwd(M, H, Kin, x, dy, out, gamma.detach(), beta.detach(), rstd, dW, db, dg, dbt)
    tp = dict(rtol=1e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=0.3)
    for got, want, n in ((dW, W.grad, "dW"), (db, b.grad, "db"), (dg, gamma.grad, "dgamma"), (dbt, beta.grad, "dbeta")):
        check(got, want, f"smallk {n}", **tp)


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_fwd_bwd_mask_and_sprel(dtype):
    B, nh, Nq, Nk = 3, 2, 19, 19
    ldp = 24
    S = torch.zeros(B, nh, Nq, ldp, device=DEV)
    S[..., :Nk] = rnd(B, nh, Nq, Nk, scale=3.0)
    kmask = torch.ones(B, Nk, dtype=torch.uint8, device=DEV)
    kmask[1, 11:] = 0
    kmask[2, 1] = 0
    dist = rnd(B, Nq, Nk, seed=3).abs() * 5
    sw, sb = torch.tensor([0.3], device=DEV, requires_grad=True), torch.tensor([-0.2], device=DEV, requires_grad=True)
    s_in = S[..., :Nk].clone().requires_grad_(True)
    logits = s_in * 0.125 + (1 - kmask.float())[:, None, None, :] * -10000.0 + (sw * dist + sb)[:, None]
    ref = torch.softmax(logits, -1)
    Pm = torch.empty(B, nh, Nq, ldp, dtype=dtype, device=DEV)
    O.softmax_fwd(S, Pm, B, nh, Nq, Nk, ldp, 0.125, kmask=kmask, dist=dist, sprel_w=sw.detach(), sprel_b=sb.detach())
    check(Pm[..., :Nk], ref, "softmax fwd", **(dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=4e-3)))
    assert (Pm[..., Nk:] == 0).all()
    dP = torch.zeros(B, nh, Nq, ldp, device=DEV)
    dP[..., :Nk] = rnd(B, nh, Nq, Nk, seed=8)
    (Pm[..., :Nk].float().detach() * 0 + ref).backward(dP[..., :Nk])
    dS = torch.empty(B, nh, Nq, ldp, dtype=dtype, device=DEV)
    dsw, dsb = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    O.softmax_bwd(Pm, dP, dS, B, nh, Nq, Nk, ldp, 0.125, dist=dist, dsprel_w=dsw, dsprel_b=dsb)
    t = dict(rtol=1e-4, atol=1e-6) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-3)
    check(dS[..., :Nk], s_in.grad, "softmax bwd", **t)
    assert (dS[..., Nk:] == 0).all()
    t = dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=5e-2)
    check(dsw, sw.grad, "d sprel w", **t)
    check(dsb, sb.grad, "d sprel b", **t)


@pytest.mark.parametrize("dtype", DTYPES)
def test_head_mean(dtype):
    B, nh, inner = 5, 4, 36 * 40
    Pm = rnd(B, nh, inner, dtype=dtype)
    out = torch.empty(B, inner, device=DEV)
    O.head_mean_fwd(Pm, out, B, nh, inner)
    check(out, Pm.float().mean(1), "head mean", rtol=1e-5, atol=1e-5)
    g = rnd(B, inner, seed=2)
    dP = torch.ones(B, nh, inner, device=DEV)
    O.head_mean_bwd(g, dP, B, nh, inner, accumulate=True)
    check(dP, 1 + (g / nh)[:, None].expand(B, nh, inner), "head mean bwd", rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", DTYPES)
def test_lndot_fwd_bwd(dtype):
    M, H = 90, 128
    Y = F.relu(rnd(M, H)).to(dtype)
    gamma, beta = (1 + 0.1 * rnd(H, seed=1)).requires_grad_(True), (0.1 * rnd(H, seed=2)).requires_grad_(True)
    w2, b2 = rnd(H, scale=0.2, seed=3).requires_grad_(True), torch.tensor([0.3], device=DEV, requires_grad=True)
    z = rnd(M, H, seed=12)
    z = torch.where(Y.float() > 0, Y.float(), -z.abs() - 0.1).requires_grad_(True)     # pre-relu with relu(z) == Y
    ref = F.layer_norm(F.relu(z), (H,), gamma, beta, 1e-12) @ w2 + b2
    dl = rnd(M, seed=6)
    ref.backward(dl)
    logit = torch.empty(M, device=DEV)
    O.lndot_fwd(Y, M, H, gamma.detach(), beta.detach(), 1e-12, w2.detach(), b2.detach(), logit)
    check(logit, ref, "lndot fwd", **(dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)))
    dZ = torch.empty(M, H, dtype=dtype, device=DEV)
    dg, dbt, dw2, db2 = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV), torch.zeros(H, device=DEV), torch.zeros(1, device=DEV)
    O.lndot_bwd(Y, M, H, gamma.detach(), beta.detach(), 1e-12, w2.detach(), dl, dZ, dg, dbt, dw2, db2)
    O.flush_part_jobs()        # (inside a backward pass these parameter gradients go through partial rows: the flush adds them up -- a no-op outside one)
    t = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    check(dZ, z.grad, "lndot dZ", **t)
    tp = dict(rtol=1e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=0.2)
    for got, want, n in ((dg, gamma.grad, "dgamma"), (dbt, beta.grad, "dbeta"), (dw2, w2.grad, "dw2"), (db2, b2.grad, "db2")):
        check(got, want, f"lndot {n}", **tp)


# ------------------------------------------------------------------------------------------ losses
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N", [17, 1000, 50265, 2051])       # >= 2048 in bf16: the 16-byte-vector, one-statistics-pass kernel (MLM vocabulary)
def test_ce_rows_matches_torch(dtype, N):
    Mr = 9
    ld = (N + 7) // 8 * 8
    x = torch.zeros(Mr, ld, device=DEV, dtype=dtype)
    x[:, :N] = rnd(Mr, N, scale=2.0).to(dtype)
    x[0, 3] = float("-inf")
    x[2, 0] = float("-inf")
    labels = torch.randint(4, N, (Mr,), device=DEV, dtype=torch.int32)
    labels[4] = -100
    xs = x[:, :N].float().clone().requires_grad_(True)
    per = F.cross_entropy(xs, labels.long(), reduction="none", ignore_index=-100)
    roww = torch.rand(Mr, device=DEV)
    (per * roww).sum().mul(0.37).backward()
    loss_row, w_out = torch.empty(Mr, device=DEV), torch.empty(Mr, device=DEV)
    d = torch.full((Mr, ld), 7.0, device=DEV, dtype=dtype)
    O.ce_rows(x, Mr, N, ld, labels, coef=0.37, row_w=roww, loss_row=loss_row, dlogits=d, ldd=ld, w_out=w_out, w_rate=0.7)
    t = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    check(loss_row, per, "ce loss", **t)
    check(w_out, M.exponential_decay(per.detach(), 0.7), "mktd weights", **t)
    check(d[:, :N], xs.grad, "ce grad", **(dict(rtol=1e-4, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-3)))
    assert (d[:, N:] == 0).all()
    # in-place gradient (MLM path) gives the same result
    x2 = x.clone()
    O.ce_rows(x2, Mr, N, ld, labels, coef=0.37, row_w=roww, dlogits=x2, ldd=ld)
    check(x2[:, :N], d[:, :N], "ce in-place", rtol=0, atol=0)


def test_kd_rows_matches_oracle():
    B, K = 7, 13
    s, t = rnd(B, K, scale=2.0), rnd(B, K, scale=2.0, seed=3)
    for b in range(B):
        s[b, K - 1 - (b % 3)] = float("-inf")
        t[b, K - 1 - (b % 3)] = float("-inf")
    w = torch.rand(B, device=DEV)
    for weighted in (False, True):
        for lt in ("sum", "mean"):
            ss = s.clone().requires_grad_(True)
            ref = M.kd_loss(ss, t, 2.0, w if weighted else None, lt)
            (ref * 0.6).backward()
            norm = 1.0 if lt == "sum" else (1.0 / B if weighted else 1.0 / (B * K))
            loss_row, ds = torch.empty(B, device=DEV), torch.empty(B, K, device=DEV)
            O.kd_rows(s, t, B, K, K, 2.0, w=w if weighted else None, norm=norm, coef=0.6, loss_row=loss_row, ds=ds)
            check(loss_row.sum(), ref.detach(), f"kd {lt} w={weighted}", rtol=1e-5, atol=1e-6)
            g = torch.nan_to_num(ss.grad, nan=0.0)
            check(ds, g, f"kd grad {lt} w={weighted}", rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("dtype", DTYPES)
def test_mse_matches_oracle_with_head_slicing(dtype):
    B, hs, ht, Nq, ldp = 4, 2, 4, 10, 16
    s, t = rnd(B, hs, Nq, ldp, dtype=dtype), rnd(B, ht, Nq, ldp, dtype=dtype, seed=4)
    w = torch.rand(B, device=DEV)
    hmin = 2
    for lt in ("sum", "mean"):
        for weighted in (False, True):
            ss = s.float().clone().requires_grad_(True)
            ref = M.mse_loss(ss[:, :hmin], t.float()[:, :hmin], w if weighted else None, lt)
            (0.8 * ref).backward()
            inner = hmin * Nq * ldp
            norm = 1.0 if lt == "sum" else 1.0 / (B * inner)
            loss = torch.zeros(1, device=DEV)
            ds = torch.zeros(B, hs, Nq, ldp, device=DEV)
            O.mse(s, t, B, inner, hs * Nq * ldp, ht * Nq * ldp, w=w if weighted else None, norm=norm, coef=0.8, loss=loss, ds=ds,
                  g_stride=hs * Nq * ldp)
            check(loss[0], ref.detach(), f"mse {lt} w={weighted}", rtol=1e-4, atol=1e-5)
            check(ds, ss.grad, f"mse grad {lt} w={weighted}", rtol=1e-4, atol=1e-6)
    # feature flavour: grad in compute dtype
    a, b = rnd(B * 6, 32, dtype=dtype), rnd(B * 6, 32, dtype=dtype, seed=2)
    aa = a.float().clone().requires_grad_(True)
    ref = M.mse_loss(aa.view(B, 6, 32), b.float().view(B, 6, 32), w, flavour="pretrain")
    ref.backward()
    loss = torch.zeros(1, device=DEV)
    ds = torch.empty(B * 6, 32, dtype=dtype, device=DEV)
    O.mse(a, b, B, 6 * 32, 6 * 32, 6 * 32, w=w, norm=1.0 / a.numel(), coef=1.0, loss=loss, ds=ds, g_stride=6 * 32)
    check(loss[0], ref.detach(), "mse feat", rtol=1e-4, atol=1e-6)
    check(ds, aa.grad, "mse feat grad", **(dict(rtol=1e-4, atol=1e-7) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-4)))


# ------------------------------------------------------------------------------------- graph ops
@pytest.mark.parametrize("dtype", DTYPES)
def test_csr_gather_and_transpose(dtype):
    n_src, n_out, H = 40, 11, 128
    src = rnd(n_src, H, dtype=dtype)
    ptr = torch.tensor([0, 0, 1, 4, 4, 6, 7, 9, 9, 10, 12, 13], dtype=torch.int32, device=DEV)
    idx = torch.tensor([5, 1, 2, 3, 7, 9, 11, 30, 31, 39, 0, 4, 6], dtype=torch.int32, device=DEV)
    w = torch.rand(13, device=DEV)
    out = torch.full((n_out, H), 3.0, dtype=dtype, device=DEV)
    O.csr_gather(src, ptr, idx, w, out, n_out, H)
    A = torch.zeros(n_out, n_src, device=DEV)
    for n in range(n_out):
        for e in range(int(ptr[n]), int(ptr[n + 1])):
            A[n, int(idx[e])] += w[e]
    check(out, A @ src.float(), "csr gather", **tol(dtype))
    out2 = out.clone()
    O.csr_gather(src, ptr, idx, w, out2, n_out, H, accumulate=True)
    check(out2, 2 * (A @ src.float()), "csr accumulate", **tol(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_pano_fuse_fwd_bwd(dtype):
    N, V, H = 6, 36, 128
    x = rnd(N, V, H, dtype=dtype)
    lens = torch.tensor([36, 36, 20, 36, 5, 36], dtype=torch.int32, device=DEV)
    wf, bf = rnd(H, scale=0.2).requires_grad_(True), torch.tensor([0.1], device=DEV, requires_grad=True)
    xx = x.float().clone().requires_grad_(True)
    mask = torch.arange(V, device=DEV)[None] < lens[:, None]
    sc = xx @ wf + bf + (~mask).float() * -10000.0
    p = torch.softmax(sc, -1)
    ref = (p[..., None] * xx).sum(1)
    df = rnd(N, H, dtype=dtype, seed=3)
    ref.backward(df.float())
    fused, probs = torch.empty(N, H, dtype=dtype, device=DEV), torch.empty(N, V, device=DEV)
    nh, ldp = 2, 40
    P = torch.softmax(rnd(N, nh, V, ldp, seed=9).float(), -1).to(dtype)
    pmean = torch.empty(N, V, ldp, device=DEV)
    O.pano_fuse_fwd(x, lens, wf.detach(), bf.detach(), fused, probs, N, V, H, P=P, nh=nh, inner=V * ldp, pmean=pmean)     # + the attention map's head mean
    check(pmean, P.float().mean(1), "head mean riding in the pano fuse launch", rtol=1e-6, atol=1e-7)
    check(fused, ref, "pano fuse fwd", **tol(dtype))
    check(probs, p, "pano fuse probs", **(dict(rtol=1e-4, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-3)))
    dx = torch.zeros(N, V, H, dtype=dtype, device=DEV)
    dwf, dbf = torch.zeros(H, device=DEV), torch.zeros(1, device=DEV)
    O.pano_fuse_bwd(x, probs, wf.detach(), df, dx, dwf, dbf, N, V, H)
    O.flush_part_jobs()
    t = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=2e-2)
    check(dx, xx.grad, "pano fuse dx", **t)
    tp = dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=0.1)
    check(dwf, wf.grad, "pano fuse dwf", **tp)
    check(dbf, bf.grad, "pano fuse dbf", **tp)


def test_sap_fuse_fwd_bwd_against_oracle_fusion():
    from magic_amd.host import synth
    from magic_amd.host.plan import build_plan
    from oracle.model_ref import fuse_logits
    batch = synth.make_batch("sap", batch_size=6, seed=5, min_len=5, max_len=9, min_steps=2, max_steps=5)
    plan = build_plan(batch, "sap", DEV)
    B, K = batch["gmap_step_ids"].shape
    Vp = 37
    g_raw = rnd(B, K).requires_grad_(True)
    l_raw = rnd(B, Vp, seed=2).requires_grad_(True)
    fuse_raw = rnd(B, seed=3).requires_grad_(True)
    fw = torch.sigmoid(fuse_raw)[:, None]
    gmask = plan["gmask"].bool()
    lmask = plan["lmask"].bool()
    gl = (g_raw * fw).masked_fill(~gmask, float("-inf"))
    ll = (l_raw * (1 - fw)).masked_fill(~lmask, float("-inf"))
    cpu_batch = batch
    fl = fuse_logits(gl.cpu(), ll.cpu(), cpu_batch).to(DEV)
    ga = batch["global_act_labels"].to(DEV)
    la = batch["local_act_labels"].to(DEV)
    loss = F.cross_entropy(gl, ga) + F.cross_entropy(ll, la, ignore_index=-100) * 0.7 + F.cross_entropy(fl, ga) * 1.3
    loss.backward()
    o_gl, o_ll, o_fl = torch.empty(B, K, device=DEV), torch.empty(B, Vp, device=DEV), torch.empty(B, K, device=DEV)
    O.sap_fuse_fwd(B, K, Vp, g_raw.detach(), l_raw.detach(), fuse_raw.detach(), plan["gmask"], plan["lmask"], plan["fsrc"],
                   plan["bwmask"], True, o_gl, o_ll, o_fl)
    for got, want, n in ((o_gl, gl, "gl"), (o_ll, ll, "ll"), (o_fl, fl, "fl")):
        assert torch.equal(torch.isinf(got), torch.isinf(want)), n
        check(torch.nan_to_num(got, neginf=0.0), torch.nan_to_num(want.detach(), neginf=0.0), n, rtol=1e-5, atol=1e-6)
    dgl, dll, dfl = torch.empty(B, K, device=DEV), torch.empty(B, Vp, device=DEV), torch.empty(B, K, device=DEV)
    ga32, la32 = ga.int(), la.int()
    O.ce_rows(o_gl, B, K, K, ga32, coef=1.0 / B, dlogits=dgl, ldd=K)
    nl = max(int((la != -100).sum()), 1)
    O.ce_rows(o_ll, B, Vp, Vp, la32, coef=0.7 / nl, dlogits=dll, ldd=Vp)
    O.ce_rows(o_fl, B, K, K, ga32, coef=1.3 / B, dlogits=dfl, ldd=K)
    dg, dl, df = torch.empty(B, K, device=DEV), torch.empty(B, Vp, device=DEV), torch.empty(B, device=DEV)
    O.sap_fuse_bwd(B, K, Vp, g_raw.detach(), l_raw.detach(), fuse_raw.detach(), plan["gmask"], plan["lmask"], plan["fsrc"],
                   plan["bwmask"], True, dgl, dll, dfl, dg, dl, df)
    check(dg, g_raw.grad, "d g_raw", rtol=1e-4, atol=1e-6)
    check(dl, l_raw.grad, "d l_raw", rtol=1e-4, atol=1e-6)
    check(df, fuse_raw.grad, "d fuse_raw", rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------- optimizer
def test_adamw_clip_and_shadow_match_oracle():
    n = 10007
    p0, g0 = rnd(n), rnd(n, seed=3) * 3
    params, grads = [p0.cpu().clone()], [g0.cpu().clone()]
    state = optim_ref.adamw_init(params)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for step in range(1, 4):
        gr = [g.clone() * step for g in grads]
        optim_ref.clip_grad_norm(gr, 5.0)
        optim_ref.adamw_step(params, gr, state, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01)
        ss = torch.zeros(1, device=DEV)
        gdev = (g0 * step).contiguous()
        O.sumsq(gdev, ss)
        check(ss[0], (gdev.double() ** 2).sum().float(), "sumsq", rtol=1e-5, atol=0)
        step_size = 1e-3 * math.sqrt(1 - 0.98 ** step) / (1 - 0.9 ** step)
        O.adamw(n, p, gdev, m, v, shadow, 1e-3, 0.9, 0.98, 1e-6, 0.01, step_size, ss, 5.0, 1.0)
        check(p, params[0], f"adamw step {step}", rtol=1e-5, atol=1e-6)
    assert torch.equal(shadow, p.to(torch.bfloat16))


@pytest.mark.parametrize("bad", [float("inf"), float("nan")])
def test_adamw_skips_a_step_whose_gradient_norm_is_not_finite(bad):
    """fp16 storage runs under a STATIC gradient scale: an overflowed activation gradient makes the norm inf / NaN.  The AdamW kernel then
    skips the update the way amp.GradScaler.step does (train_r2r_magic.py:370-371): weights, moments and shadow untouched, the consumed
    gradient zeroed, the skip counted -- and the next finite step proceeds from the untouched state."""
    n = 4099
    p0, g0 = rnd(n), rnd(n, seed=3)
    p, m, v = p0.clone(), rnd(n, seed=4).abs() * 0.01, rnd(n, seed=5).abs() * 0.01
    m0, v0 = m.clone(), v.clone()
    shadow = p.to(torch.float16)
    sh0 = shadow.clone()
    over = torch.zeros(1, dtype=torch.int32, device=DEV)
    g = g0.clone()
    g[17] = bad
    ss = torch.zeros(1, device=DEV)
    O.sumsq(g, ss)
    assert not torch.isfinite(ss).item()
    O.adamw(n, p, g, m, v, shadow, 1e-3, 0.9, 0.98, 1e-6, 0.01, 1e-3, ss, 5.0, 1.0, zero_grad=True, overflow=over)
    torch.cuda.synchronize()
    assert torch.equal(p, p0) and torch.equal(m, m0) and torch.equal(v, v0) and torch.equal(shadow, sh0)
    assert int(over.item()) == 1 and float(g.abs().max()) == 0.0
    g = g0.clone()
    ss.zero_()
    O.sumsq(g, ss)
    O.adamw(n, p, g, m, v, shadow, 1e-3, 0.9, 0.98, 1e-6, 0.01, 1e-3, ss, 5.0, 1.0, zero_grad=True, overflow=over)
    torch.cuda.synchronize()
    assert int(over.item()) == 1 and torch.isfinite(p).all() and not torch.equal(p, p0) and torch.equal(shadow, p.to(torch.float16))


def test_cast_roundtrip_and_add():
    x = rnd(1003)
    y = O.cast_to(x, torch.bfloat16)
    assert torch.equal(y, x.to(torch.bfloat16))
    z = O.cast_to(y, torch.float32)
    assert torch.equal(z, y.float())
    a, b = rnd(77), rnd(77, seed=2)
    want = a + b
    O.add_(a, b)
    check(a, want, "add", rtol=0, atol=0)
    dy, zz = rnd(50, 8), rnd(50, 8, seed=9)
    zr = zz.clone().requires_grad_(True)
    F.gelu(zr).backward(dy)
    check(O.dact(dy, zz, 1), zr.grad, "dgelu", rtol=1e-5, atol=1e-6)


def test_gemm_default_tile_rule_large_forward_and_long_reduction_weight_gradient():
    """shapes the DEFAULT rules send to the wide tile (>= 192 tiles of 128x128, K >= 512: forward and input gradient) and to the
    long-reduction split-K branch of the weight gradient (ops._splitk with more 64x64 tiles than CUs)"""
    dtype = torch.bfloat16
    M, N, K = 2176, 1536, 520                      # 17 x 12 = 204 wide tiles, K-tail of 8
    x, W, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, scale=0.1), rnd(N, scale=0.5)
    ref = x.float() @ W.float().t() + b
    check(O.linear_fwd(x, W, b, M), ref, "default nt", rtol=2e-2, atol=8e-2)
    dy = rnd(M, N, dtype=dtype, scale=0.3)
    check(O.linear_dx(dy, W, M), dy.float() @ W.float(), "default nn", rtol=2e-2, atol=2e-1)
    dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
    assert O._splitk((N // 64) * ((K + 63) // 64), M) == 2
    O.linear_dw(dy, x, dW, db, M)
    check(dW, dy.float().t() @ x.float(), "default dW", rtol=2e-2, atol=5e-2 * math.sqrt(M / 64))
    check(db, dy.float().sum(0), "default db", rtol=2e-2, atol=5e-2 * math.sqrt(M / 64))


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_k_group_kernel_few_tiles_long_k(dtype):
    """few output tiles and a long K -> gemm_kg_kernel (4 K-groups per workgroup, partial sums added through LDS): forward with every
    epilogue operand, input gradient through the transposed operand image, ragged M / N and a K tail"""
    M, N, K = 601, 776, 3080                       # 10 x 13 = 130 tiles of 64x64, 49 K-tiles (the last one 8 deep)
    x, W = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, scale=0.05)
    b, res = rnd(N, scale=0.5), rnd(M, N, dtype=dtype)
    ref = x.float() @ W.float().t() + b
    t = tol(dtype) if dtype == torch.float32 else dict(rtol=2e-2, atol=1e-1)
    check(O.linear_fwd(x, W, b, M), ref, "kg nt+bias", **t)
    pre = torch.empty(M, N, dtype=dtype, device=DEV)
    out = O.linear_fwd(x, W, b, M, epilogue=1, residual=res, pre=pre)
    check(pre, ref, "kg pre-activation", **t)
    check(out, F.gelu(ref) + res.float(), "kg gelu+residual", **t)
    K2 = 768                                        # dx[M,K2] = dy[M,N2] @ W2[N2,K2], contraction over N2 = 3080
    N2 = 3080
    dy, W2 = rnd(M, N2, dtype=dtype, scale=0.3), rnd(N2, K2, dtype=dtype, scale=0.05)
    z, r = rnd(M, K2, dtype=dtype), rnd(M, K2, dtype=dtype)
    refx = dy.float() @ W2.float()
    check(O.linear_dx(dy, W2, M), refx, "kg nn", **t)
    zz = z.float().requires_grad_(True)
    F.gelu(zz).backward(torch.ones_like(zz))
    check(O.linear_dx(dy, W2, M, epilogue=3, aux=z, residual=r), refx * zz.grad + r.float(), "kg nn+dgelu+res", **t)
    # K between 768 and one round of 4 tiles short of the pipeline depth
    x3, W3 = rnd(M, 768, dtype=dtype), rnd(N, 768, dtype=dtype, scale=0.05)
    check(O.linear_fwd(x3, W3, None, M), x3.float() @ W3.float().t(), "kg nt K=768", **t)


def test_gemm_wide_tile_random_shapes_and_batched_strides(big_tiles):
    """the LDS-DMA wide tile against fp32 torch on seeded random shapes (every extent a multiple of 8, tile edges in all three
    dimensions, K down to one 8-element chunk) and on a batched (batch, head)-strided problem"""
    g = torch.Generator().manual_seed(7)
    dtype = torch.bfloat16
    shapes = [(128, 128, 8), (136, 264, 72), (1024, 128, 64), (129 * 8, 131 * 8, 65 * 8)]
    for _ in range(6):
        M, N, K = (int(torch.randint(16, 200, (1,), generator=g)) * 8 for _ in range(3))
        shapes.append((M, N, K))
    for M, N, K in shapes:
        x, W = rnd(M, K, dtype=dtype, seed=M + K), rnd(N, K, dtype=dtype, scale=0.1, seed=N)
        atol = 2e-2 * math.sqrt(K / 8) + 2e-2
        check(O.linear_fwd(x, W, None, M), x.float() @ W.float().t(), f"wide nt {M}x{N}x{K}", rtol=2e-2, atol=atol)
        dy = rnd(M, N, dtype=dtype, scale=0.3, seed=M)
        check(O.linear_dx(dy, W, M), dy.float() @ W.float(), f"wide nn {M}x{N}x{K}", rtol=2e-2, atol=2e-2 * math.sqrt(N / 8) + 2e-2)
        dW = torch.zeros(N, K, device=DEV)
        O.linear_dw(dy, x, dW, None, M)
        check(dW, dy.float().t() @ x.float(), f"wide tn {M}x{N}x{K}", rtol=2e-2, atol=3e-2 * math.sqrt(M / 8) + 2e-2)
    # batched: C[b,h] = A[b,h] @ B[b,h]^T with (batch, head) strides, 3 x 2 problems of 256 x 136 x 64 inside padded buffers
    Bn, nh, Mq, Nk, D = 3, 2, 256, 136, 64
    A = rnd(Bn, Mq, nh * D + 8, dtype=dtype, seed=3)
    Bm = rnd(Bn, Nk, nh * D + 8, dtype=dtype, seed=4)
    Cc = torch.zeros(Bn, nh, Mq, Nk + 8, dtype=dtype, device=DEV)
    ld = nh * D + 8
    O.gemm(0, A, Bm, Cc, Mq, Nk, D, ld, ld, Nk + 8, batch=Bn * nh, nh=nh, sA=(Mq * ld, D), sB=(Nk * ld, D), sC=(nh * Mq * (Nk + 8), Mq * (Nk + 8)))
    for b in range(Bn):
        for h in range(nh):
            ref = A[b, :, h * D:(h + 1) * D].float() @ Bm[b, :, h * D:(h + 1) * D].float().t()
            check(Cc[b, h, :, :Nk], ref, f"wide batched {b},{h}", rtol=2e-2, atol=8e-2)
    assert float(Cc[..., Nk:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_deferred_weight_gradients_many_problems_per_launch(dtype):
    """magic_gemm_dw_grouped with compact descriptors: 40 deferred problems of mixed shapes (split-K or not, with and without a bias
    gradient, accumulating into non-zero buffers) flushed 32 + 8 per launch must equal the per-problem sums"""
    g = torch.Generator().manual_seed(5)
    probs, refs = [], []
    O.DEFER["queue"].clear()
    O.defer_dw(True)
    for i in range(40):
        M = int(torch.randint(3, 60, (1,), generator=g)) * 8 + (0 if i % 3 else 5)
        N = int(torch.randint(1, 6, (1,), generator=g)) * 64 if i % 4 else 24
        K = int(torch.randint(1, 6, (1,), generator=g)) * 64 if i % 5 else 40
        if i == 7:
            M, N, K = 3840, 128, 512                # long reduction: host picks split-K > 1
        dy, x = rnd(M, N, dtype=dtype, scale=0.3, seed=100 + i), rnd(M, K, dtype=dtype, seed=200 + i)
        dW = rnd(N, K, seed=300 + i).contiguous()
        db = rnd(N, seed=400 + i) if i % 2 else None
        refs.append((dW.clone() + dy.float().t() @ x.float(), None if db is None else db.clone() + dy.float().sum(0)))
        O.linear_dw(dy, x, dW, db, M)
        probs.append((dW, db, M))
    assert len(O.DEFER["queue"]) == 40
    O.flush_dw(group=32)
    torch.cuda.synchronize()
    for i, ((dW, db, M), (rW, rb)) in enumerate(zip(probs, refs)):
        t = dict(rtol=1e-4, atol=2e-3) if dtype == torch.float32 else dict(rtol=2e-2, atol=5e-2 * math.sqrt(M / 64) + 1e-2)
        check(dW, rW, f"dW[{i}]", **t)
        if db is not None:
            check(db, rb, f"db[{i}]", **t)


@pytest.mark.parametrize("dtype", DTYPES)
def test_weight_gradient_launch_is_bitwise_reproducible_and_sums_repeated_layers_in_a_fixed_order(dtype):
    """The deterministic seam of magic_gemm_dw_grouped (csrc/gemm.hip dw_seam): K-splits store partials, the last workgroup at a tile adds them in
    slot order.  (i) the same queue flushed five times from the same starting buffers gives BITWISE-identical dW and db (the atomic form does
    not); (ii) one dW queued several times in a launch (a Linear called at every navigator step, the weight-tied MLM decoder) is summed exactly
    once per problem; (iii) both forms agree with the fp32 reference; (iv) the arrival counters are back at zero after every launch."""
    g = torch.Generator().manual_seed(9)
    shapes = [(3840, 128, 512), (3840, 512, 128), (10440, 384, 128), (960, 128, 128), (48, 128, 256), (1776, 24, 40), (3840, 128, 128)]
    ops_, bufs = [], []
    for i, (M, N, K) in enumerate(shapes):
        dW, db = rnd(N, K, seed=300 + i).contiguous(), (rnd(N, seed=400 + i) if i % 2 == 0 else None)
        reps = 4 if i in (3, 6) else 1                    # the same Linear used four times: four problems, one dW
        terms = []
        for r in range(reps):
            dy, x = rnd(M, N, dtype=dtype, scale=0.3, seed=1000 + 10 * i + r), rnd(M, K, dtype=dtype, seed=2000 + 10 * i + r)
            terms.append((dy, x))
        ops_.append((terms, dW, db, M))
        bufs.append((dW.clone(), None if db is None else db.clone()))

    def run(det):
        outs = []
        for (terms, dW, db, M), (w0, b0) in zip(ops_, bufs):
            dW.copy_(w0)
            if db is not None:
                db.copy_(b0)
        O.DEFER["queue"].clear()
        O.defer_dw(True)
        for terms, dW, db, M in ops_:
            for dy, x in terms:
                O.linear_dw(dy, x, dW, db, M)
        prev = O.DW_DETERMINISTIC
        O.DW_DETERMINISTIC = det
        try:
            O.flush_dw()
        finally:
            O.DW_DETERMINISTIC = prev
        torch.cuda.synchronize()
        for terms, dW, db, M in ops_:
            outs.append((dW.clone(), None if db is None else db.clone()))
        return outs
    first = run(True)
    cnt = O.dw_counters(DEV)
    assert cnt is not None and int(cnt.abs().max()) == 0
    for _ in range(4):
        again = run(True)
        for (a, ab), (b, bb) in zip(first, again):
            assert torch.equal(a, b) and (ab is None or torch.equal(ab, bb))
        assert int(cnt.abs().max()) == 0
    atom = run(False)
    for i, ((terms, dW, db, M), (w0, b0)) in enumerate(zip(ops_, bufs)):
        rW = w0 + sum(dy.float().t() @ x.float() for dy, x in terms)
        t = dict(rtol=1e-4, atol=3e-3) if dtype == torch.float32 else dict(rtol=2e-2, atol=5e-2 * math.sqrt(M * len(terms) / 64) + 1e-2)
        check(first[i][0], rW, f"deterministic dW[{i}]", **t)
        check(atom[i][0], rW, f"atomic dW[{i}]", **t)
        if db is not None:
            rb = b0 + sum(dy.float().sum(0) for dy, x in terms)
            check(first[i][1], rb, f"deterministic db[{i}]", **t)
            check(atom[i][1], rb, f"atomic db[{i}]", **t)


@pytest.mark.parametrize("layout,M,N,K", [(0, 200, 384, 128), (1, 77, 128, 512), (2, 96, 132, 300), (0, 64, 64, 1000)])
def test_fp32_gemm_split_bf16_contraction(layout, M, N, K):
    """magic_set_f32_mfma(1): fp32 operands, a.b = a_hi b_hi + a_hi b_lo + a_lo b_hi on the bf16 matrix cores.  Against fp64: relative
    error ~1e-5 of sum |a||b| (exact fp32 MFMA: ~1e-7; single bf16 operands: ~4e-3)."""
    from magic_amd.host import lib as L
    g = torch.Generator().manual_seed(M + N + K)
    rnd = lambda *s_: torch.randn(*s_, generator=g).to(DEV)
    if layout == 0:
        A, B = rnd(M, K), rnd(N, K); ref = A.double() @ B.double().t(); lda, ldb = K, K
    elif layout == 1:
        A, B = rnd(M, K), rnd(K, N); ref = A.double() @ B.double(); lda, ldb = K, N
    else:
        A, B = rnd(K, M), rnd(K, N); ref = A.double().t() @ B.double(); lda, ldb = M, N
    scale = (A.abs().double().mean() * B.abs().double().mean() * K).item()
    errs = {}
    for mode in ("exact", "bf16x3"):
        prev = L.set_f32_mfma(mode)
        try:
            C = torch.empty(M, N, device=DEV)
            O.gemm(layout, A, B, C, M, N, K, lda, ldb, N)
            torch.cuda.synchronize()
        finally:
            L.set_f32_mfma(prev)
        errs[mode] = (C.double() - ref).abs().max().item() / scale
    assert errs["exact"] < 5e-6 and errs["bf16x3"] < 2e-4, errs
    assert errs["bf16x3"] > 0                      # the split path really ran (it is not bit-identical to the exact instruction)


@pytest.mark.parametrize("dtype,B,H", [(torch.float32, 48, 128), (torch.bfloat16, 48, 128), (torch.float32, 7, 256), (torch.bfloat16, 64, 256)])
def test_cfp_loss_kernel_matches_autograd(dtype, B, H):
    """magic_cfp_loss (three contrastive terms, forward + backward, one launch) against torch autograd of the reference arithmetic
    (train_r2r_magic.py:548-560: cross_entropy(sim, arange) + cross_entropy(sim.T, arange) on sim = a txt^T / temperature)"""
    import torch.nn.functional as F
    from magic_amd.host import ops as O
    g = torch.Generator().manual_seed(B * 1000 + H)
    a = [(torch.randn(B, H, generator=g) * 0.5).to(DEV).to(dtype).contiguous() for _ in range(3)]
    txt = (torch.randn(B, H, generator=g) * 0.5).to(DEV).to(dtype).contiguous()
    temp, coef = 0.07 * 10, 0.37 / B
    rows = torch.empty(6, B, device=DEV, dtype=torch.float32)
    d_a = [torch.empty(B, H, device=DEV, dtype=dtype) for _ in range(3)]
    d_txt = torch.empty(B, H, device=DEV, dtype=dtype)
    assert O.cfp_loss_ok(B, H)
    O.cfp_loss(B, H, a, txt, temp, coef, rows, d_a=d_a, d_txt=d_txt)
    rows2 = torch.empty_like(rows)
    O.cfp_loss(B, H, a, txt, temp, coef, rows2)                         # losses only
    torch.cuda.synchronize()
    ar = torch.arange(B, device=DEV)
    a64 = [x.double().requires_grad_(True) for x in a]
    t64 = txt.double().requires_grad_(True)
    tot, want_rows = 0.0, []
    for x in a64:
        sim = x @ t64.t() / temp
        l1, l2 = F.cross_entropy(sim, ar, reduction="none"), F.cross_entropy(sim.t(), ar, reduction="none")
        want_rows += [l1, l2]
        tot = tot + coef * (l1.sum() + l2.sum())
    tot.backward()
    want = torch.stack(want_rows).detach()
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert torch.allclose(rows.double(), want, rtol=1e-4, atol=1e-4), (rows.double() - want).abs().max().item()
    assert torch.equal(rows, rows2)
    for got, ref, nm in [(d_a[i], a64[i].grad, f"d_a{i}") for i in range(3)] + [(d_txt, t64.grad, "d_txt")]:
        err = (got.double() - ref).abs().max().item()
        assert err <= tol * ref.abs().max().item() + 1e-7, (nm, err, ref.abs().max().item())


def test_mse_multi_device_side_extents_match_a_masked_reference():
    """magic_mse_multi with valid_dev / valid_mod / norm_dev (shape-bucketed batches): only the batch's true extent enters the loss, the gradient
    outside it is zero, and the normaliser comes from device memory"""
    from magic_amd.host import ops as O
    g = torch.Generator().manual_seed(11)
    B, nh_s, nh_t, Nq, ld, Nq_true, Ht, rows, rows_true, Np, Np_true = 6, 2, 4, 24, 32, 19, 64, 24, 17, 10, 7
    s_att = torch.rand(B, nh_s, Nq, ld, generator=g).to(DEV).bfloat16()
    t_att = torch.rand(B, nh_t, Nq, ld, generator=g).to(DEV).bfloat16()
    s_emb = torch.randn(B, rows, Ht, generator=g).to(DEV).bfloat16()
    t_emb = torch.randn(B, rows, Ht, generator=g).to(DEV).bfloat16()
    s_out = torch.randn(Np, 5 * Ht, generator=g).to(DEV).bfloat16()
    t_out = torch.randn(Np, 5 * Ht, generator=g).to(DEV).bfloat16()
    loss = torch.zeros(3, device=DEV)
    d_att = torch.full((B, nh_s, Nq, ld), 7.0, device=DEV)                      # fp32 gradient, pre-filled: the kernel must overwrite all of it
    d_emb = torch.full((B, rows, Ht), 7.0, device=DEV).bfloat16()
    d_out = torch.full((Np, 5 * Ht), 7.0, device=DEV).bfloat16()
    hmin = 2
    vi = torch.tensor([B, Nq_true * ld, B, rows_true * Ht, Np_true, 5 * Ht], dtype=torch.int32, device=DEV)
    vf = torch.tensor([1.0 / (B * hmin * Nq_true * 20), 1.0 / (B * rows_true * Ht), 1.0 / (Np_true * 5 * Ht)], device=DEV)
    coef = 0.5
    O.mse_multi([
        dict(s=s_att, t=t_att, outer=B, inner=hmin * Nq * ld, s_stride=nh_s * Nq * ld, t_stride=nh_t * Nq * ld, norm=1.0, coef=coef, loss=loss[0:1], ds=d_att,
             g_stride=nh_s * Nq * ld, valid_dev=vi[0:2], norm_dev=vf[0:1], valid_mod=Nq * ld),
        dict(s=s_emb, t=t_emb, outer=B, inner=rows * Ht, s_stride=rows * Ht, t_stride=rows * Ht, norm=1.0, coef=coef, loss=loss[1:2], ds=d_emb,
             g_stride=rows * Ht, valid_dev=vi[2:4], norm_dev=vf[1:2]),
        dict(s=s_out, t=t_out, outer=Np, inner=5 * Ht, s_stride=5 * Ht, t_stride=5 * Ht, norm=1.0, coef=coef, loss=loss[2:3], ds=d_out,
             g_stride=5 * Ht, valid_dev=vi[4:6], norm_dev=vf[2:3])])
    torch.cuda.synchronize()
    # references
    da = (s_att.float()[:, :hmin] - t_att.float()[:, :hmin])
    da[:, :, Nq_true:] = 0
    assert abs(loss[0].item() - (da ** 2).sum().item() * vf[0].item()) < 1e-4 * max(1.0, loss[0].item())
    assert torch.allclose(d_att[:, :hmin], 2 * coef * vf[0] * da, rtol=1e-5, atol=1e-8) and (d_att[:, :, Nq_true:] == 0).all()
    de = (s_emb.float() - t_emb.float())
    de[:, rows_true:] = 0
    assert abs(loss[1].item() - (de ** 2).sum().item() * vf[1].item()) < 1e-4 * max(1.0, loss[1].item())
    assert torch.allclose(d_emb.float(), (2 * coef * vf[1] * de).bfloat16().float(), rtol=1e-2, atol=1e-6) and (d_emb[:, rows_true:] == 0).all()
    do = (s_out.float() - t_out.float())
    do[Np_true:] = 0
    assert abs(loss[2].item() - (do ** 2).sum().item() * vf[2].item()) < 1e-4 * max(1.0, loss[2].item())
    assert (d_out[Np_true:] == 0).all() and torch.allclose(d_out.float(), (2 * coef * vf[2] * do).bfloat16().float(), rtol=1e-2, atol=1e-6)


def test_csr_gather_multi_equals_consecutive_gathers():
    import numpy as np
    from magic_amd.host import ops as O
    from magic_amd.host.plan import csr_pair
    g = torch.Generator().manual_seed(5)
    H, n_src, n_out = 128, 90, 40
    src1 = torch.randn(n_src, H, generator=g).to(DEV).bfloat16()
    src2 = torch.randn(n_src, H, generator=g).to(DEV).bfloat16()
    rng = np.random.default_rng(3)
    def rand_csr(p):
        ent = [(o, int(s), float(rng.uniform(0.1, 1.0))) for o in range(n_out) for s in rng.choice(n_src, rng.integers(0, 4), replace=False) if rng.uniform() < p]
        f, _ = csr_pair(ent, n_out, n_src)
        return tuple(torch.as_tensor(a).to(DEV) for a in f)
    c1, c2, c3 = rand_csr(0.8), rand_csr(0.5), rand_csr(0.9)
    base = torch.randn(n_out, H, generator=g).to(DEV).bfloat16()
    ref_a = base.clone()
    O.csr_gather(src1, *c1, ref_a, n_out, H, accumulate=True)
    O.csr_gather(src2, *c2, ref_a, n_out, H, accumulate=True)
    ref_b = torch.empty(n_out, H, device=DEV, dtype=torch.bfloat16)
    O.csr_gather(src2, *c3, ref_b, n_out, H)
    out_a, out_b = base.clone(), torch.full((n_out, H), 3.0, device=DEV).bfloat16()
    O.csr_gather_multi(H, [dict(out=out_a, n_out=n_out, accumulate=True, src1=src1, csr1=c1, src2=src2, csr2=c2),
                           dict(out=out_b, n_out=n_out, src1=src2, csr1=c3)])
    torch.cuda.synchronize()
    assert torch.equal(out_a, ref_a) and torch.equal(out_b, ref_b)


@pytest.mark.parametrize("use_gate,hard,pred", [(True, True, True), (False, False, True), (True, True, False), (True, False, False)])
def test_sap_fuse_loss_equals_the_six_launches_it_replaces(use_gate, hard, pred):
    """magic_sap_fuse_loss = sap_fuse_fwd + 3 x ce_rows + teacher-sample weights (ce_rows w_out) + kd_rows (accumulated into dfl)"""
    from magic_amd.host import synth
    from magic_amd.host.plan import build_plan
    batch = synth.make_batch("sap", batch_size=7, seed=11, min_len=5, max_len=9, min_steps=2, max_steps=5)
    plan = build_plan(batch, "sap", DEV)
    B, K = batch["gmap_step_ids"].shape
    Vp = 37
    g_raw, l_raw, fuse_raw = rnd(B, K), rnd(B, Vp, seed=2), rnd(B, seed=3)
    t_fused = rnd(B, K, seed=4) * 2
    t_fused[~plan["gmask"].bool()] = float("-inf")
    ga, la = plan["global_act_labels"], plan["local_act_labels"]
    coef, T, rate, kcoef = 0.37 / B, 2.0, 0.7, 0.23
    kdev = torch.tensor([1.7], device=DEV)
    # the separate launches
    gl, ll, fl = torch.empty(B, K, device=DEV), torch.empty(B, Vp, device=DEV), torch.empty(B, K, device=DEV)
    O.sap_fuse_fwd(B, K, Vp, g_raw, l_raw, fuse_raw, plan["gmask"], plan["lmask"], plan["fsrc"], plan["bwmask"], use_gate, gl, ll, fl)
    rows = torch.empty(3, B, device=DEV)
    dgl, dll, dfl = torch.empty(B, K, device=DEV), torch.empty(B, Vp, device=DEV), torch.empty(B, K, device=DEV)
    O.ce_rows(gl, B, K, K, ga, coef=coef, loss_row=rows[0], dlogits=dgl, ldd=K)
    O.ce_rows(ll, B, Vp, Vp, la, coef=coef, loss_row=rows[1], dlogits=dll, ldd=Vp)
    O.ce_rows(fl, B, K, K, ga, coef=coef, loss_row=rows[2], dlogits=dfl, ldd=K)
    w = None
    if hard:
        w = torch.empty(B, device=DEV)
        O.ce_rows(t_fused, B, K, K, ga, w_out=w, w_rate=rate)
    kdr = None
    norm = 1.0 / B if hard else 1.0 / (B * K)
    if pred:
        kdr = torch.empty(B, device=DEV)
        O.kd_rows(fl, t_fused, B, K, K, T, w=w, norm=norm, coef=kcoef, coef_dev=kdev, loss_row=kdr, ds=dfl, accumulate=True)
    # the one launch
    gl2, ll2, fl2 = torch.empty(B, K, device=DEV), torch.empty(B, Vp, device=DEV), torch.empty(B, K, device=DEV)
    rows2 = torch.empty(3, B, device=DEV)
    dgl2, dll2, dfl2 = torch.empty(B, K, device=DEV), torch.empty(B, Vp, device=DEV), torch.empty(B, K, device=DEV)
    w2 = torch.empty(B, device=DEV) if hard else None
    kdr2 = torch.empty(B, device=DEV) if pred else None
    O.sap_fuse_loss(B, K, Vp, g_raw, l_raw, fuse_raw, plan["gmask"], plan["lmask"], plan["fsrc"], plan["bwmask"], use_gate, gl2, ll2, fl2, ga, la, coef, rows2,
                    dgl=dgl2, dll=dll2, dfl=dfl2, t_fused=t_fused if (hard or pred) else None, w_rate=rate, w_out=w2, T=T, kd_norm=norm, kd_coef=kcoef,
                    kd_coef_dev=kdev if pred else None, kd_rows=kdr2)
    torch.cuda.synchronize()
    for a, b_, nme in ((gl2, gl, "gl"), (ll2, ll, "ll"), (fl2, fl, "fl")):
        assert torch.equal(a, b_), nme                                     # same arithmetic, element for element
    tol_ = dict(rtol=2e-6, atol=1e-7)                                       # (the row reductions run in a different order: one wave instead of four)
    check(rows2, rows, "CE rows", **tol_)
    for a, b_, nme in ((dgl2, dgl, "dgl"), (dll2, dll, "dll"), (dfl2, dfl, "dfl (+ KD)")):
        check(a, b_, nme, rtol=2e-5, atol=1e-8)
    if hard:
        check(w2, w, "teacher-sample weights", **tol_)
    if pred:
        check(kdr2, kdr, "action-distillation rows", rtol=2e-5, atol=1e-8)
    # losses only (no gradient buffers): the evaluation form
    rows3 = torch.empty(3, B, device=DEV)
    O.sap_fuse_loss(B, K, Vp, g_raw, l_raw, fuse_raw, plan["gmask"], plan["lmask"], plan["fsrc"], plan["bwmask"], use_gate, gl2, ll2, fl2, ga, la, coef, rows3)
    check(rows3, rows, "CE rows without gradients", **tol_)


@pytest.mark.parametrize("H,M", [(768, 608), (384, 100), (768, 5000)])
def test_partial_row_parameter_gradients_equal_the_atomic_form(H, M):
    """round 5: inside a backward pass (weight-gradient queue active) the LayerNorm / position-embedding backwards of the wide models store
    their parameter-gradient sums per workgroup and one column-sum launch (magic_colsum_add_v) finishes them at the flush: same dx, same
    gradients (fp32 summation order) as the atomic form, and two runs of the partial form are bitwise equal."""
    import magic_amd.host.ops as O
    g = torch.Generator().manual_seed(H + M)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    dt = torch.bfloat16
    dy, y = rnd(M, H).to(dt), rnd(M, H).to(dt)
    gamma, beta, rstd = 1 + 0.1 * rnd(H), 0.1 * rnd(H), rnd(M).abs() + 0.5
    Kin = 7
    x = rnd(M, Kin)

    def run(partial):
        dx = torch.empty(M, H, dtype=dt, device=DEV)
        dg, db = torch.full((H,), 0.25, device=DEV), torch.full((H,), -0.5, device=DEV)
        dW, dbl, dg2, db2 = torch.full((H, Kin), 0.125, device=DEV), torch.zeros(H, device=DEV), torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
        O.defer_dw(partial)
        try:
            assert O.part_ok(H) == partial
            O.ln_bwd(M, H, dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dx=dx, dgamma=dg, dbeta=db)
            O.smallk_ln_bwd(M, H, Kin, x, dy, y, gamma, beta, rstd, dW, dbl, dg2, db2)
            O.smallk_ln_bwd_pair(H, [dict(M=M, Kin=Kin, x=x, dy=dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dW=dW, db=dbl, dgamma=dg2, dbeta=db2),
                                     dict(M=M // 3, Kin=Kin, x=x, dy=dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dW=dW, db=dbl, dgamma=dg2, dbeta=db2)])
            if partial:
                assert len(O.PART_JOBS) == 2 + 4 + 8
            O.flush_dw()
        finally:
            O.defer_dw(False)
        torch.cuda.synchronize()
        return [t.clone() for t in (dx, dg, db, dW, dbl, dg2, db2)]
    a, b, c = run(False), run(True), run(True)
    for i, (u, v, w) in enumerate(zip(a, b, c)):
        sc = max(1.0, u.float().abs().max().item())
        assert (u.float() - v.float()).abs().max().item() <= 2e-4 * sc, i
        assert torch.equal(v, w), i


@pytest.mark.parametrize("dtype", DTYPES)
def test_weight_gradients_over_many_row_segments_in_one_launch(dtype):
    """magic_gemm_dw_cat (csrc/gemm.hip gemm_dw_cat_kernel): dW_p += sum_s dY_{p,s}^T X_{p,s}, db_p += column sums, for several Linears at once with
    ragged segment rows (one of them 0: skipped) and shapes that are not multiples of the 64 x 64 tile -- against the fp32 reference, and bitwise
    reproducible from the same starting buffers (one workgroup per tile, sums in segment order: no atomics)."""
    import numpy as np
    shapes = [(768, 768, True), (128, 512, False), (24, 40, True), (768, 7, True), (1, 768, True)]           # (N out, K in, bias)
    rows = [624, 592, 0, 16, 37, 624]
    probs, tabs, ref = [], [], []
    for i, (N, K, hb) in enumerate(shapes):
        dW, db = rnd(N, K, seed=50 + i).contiguous(), (rnd(N, seed=60 + i) if hb else None)
        segs = []
        for s, M in enumerate(rows):
            Mr = max(M, 1)
            ld_k = (K + 7) // 8 * 8
            ld_n = (N + 7) // 8 * 8
            dy = rnd(Mr, ld_n, dtype=dtype, scale=0.3, seed=1000 + 10 * i + s)
            x = rnd(Mr, ld_k, dtype=dtype, seed=2000 + 10 * i + s)
            segs.append((dy, x, M, ld_n, ld_k))
        probs.append((dW, db, N, K, segs))
        rw, rb = dW.double().clone(), (db.double().clone() if hb else None)
        for dy, x, M, ld_n, ld_k in segs:
            if M:
                rw += dy[:M, :N].double().t() @ x[:M, :K].double()
                if hb:
                    rb += dy[:M, :N].double().sum(0)
        ref.append((rw, rb))
    n_seg = len(rows)
    dy_t = np.array([[sg[0].data_ptr() for sg in p[4]] for p in probs], np.int64)
    x_t = np.array([[sg[1].data_ptr() for sg in p[4]] for p in probs], np.int64)
    m_t = np.array([[sg[2] for sg in p[4]] for p in probs], np.int32)
    d_dy, d_x, d_m = torch.from_numpy(dy_t).to(DEV), torch.from_numpy(x_t).to(DEV), torch.from_numpy(m_t).to(DEV)
    plist = [(p[0].data_ptr(), p[1].data_ptr() if p[1] is not None else 0, p[2], p[3], p[4][0][3], p[4][0][4], p[0].stride(0)) for p in probs]
    start = [(p[0].clone(), None if p[1] is None else p[1].clone()) for p in probs]

    def run():
        for p, (w0, b0) in zip(probs, start):
            p[0].copy_(w0)
            if b0 is not None:
                p[1].copy_(b0)
        O.dw_cat(dtype, plist, n_seg, d_dy.data_ptr(), d_x.data_ptr(), d_m.data_ptr())
        torch.cuda.synchronize()
        return [(p[0].clone(), None if p[1] is None else p[1].clone()) for p in probs]
    first = run()
    tol = dict(rtol=2e-3, atol=2e-2) if dtype != torch.float32 else dict(rtol=1e-4, atol=1e-3)
    for (w, b), (rw, rb), (N, K, hb) in zip(first, ref, shapes):
        check(w, rw, f"dW cat {N}x{K}", **tol)
        if hb:
            check(b, rb, f"db cat {N}", **tol)
    for _ in range(3):
        again = run()
        for (w, b), (w1, b1) in zip(first, again):
            assert torch.equal(w, w1) and (b is None or torch.equal(b, b1))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("H,M", [(128, 257), (768, 70), (256, 33), (384, 5)])
def test_ln_bwd_tail_matches_torch_autograd_and_the_three_separate_launches(dtype, H, M):
    """magic_ln_bwd_tail (the MLM head's transform, backward): fp32 dy in, dx = LayerNorm'(dy) x gelu'(pre) out, gamma / beta gradients -- against
    fp32 torch autograd of LayerNorm(gelu(pre)) and against cast + magic_ln_bwd + magic_dact, the launches it replaces"""
    gen = torch.Generator(DEV).manual_seed(H + M)
    pre32 = torch.randn(M, H, device=DEV, generator=gen)
    gamma = (1.0 + 0.1 * torch.randn(H, device=DEV, generator=gen)).contiguous()
    beta = (0.1 * torch.randn(H, device=DEV, generator=gen)).contiguous()
    dy32 = torch.randn(M, H, device=DEV, generator=gen).contiguous()
    pre = pre32.to(dtype)
    # forward through the library (what the backward recovers xhat from): y = LN(gelu(pre)), rstd
    act = F.gelu(pre.float()).to(dtype)
    y, rstd = torch.empty(M, H, device=DEV, dtype=dtype), torch.empty(M, device=DEV, dtype=torch.float32)
    O.ln_fwd(M, H, y, in0=act, gamma=gamma, beta=beta, eps=1e-12, rstd=rstd)
    # reference: fp32 autograd on the same rounded operands
    p = pre.float().clone().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    F.layer_norm(F.gelu(p).to(dtype).float(), (H,), g_, b_, 1e-12).backward(dy32)
    # fused
    dx = torch.empty(M, H, device=DEV, dtype=dtype)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    O.ln_bwd_tail(M, H, dy32, y, gamma, beta, rstd, pre, 1, dx, dg, db)
    # the three launches it replaces
    dg3, db3 = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    d_ln = torch.empty(M, H, device=DEV, dtype=dtype)
    O.ln_bwd(M, H, dy32.to(dtype), y=y, gamma=gamma, beta=beta, rstd=rstd, dx=d_ln, dgamma=dg3, dbeta=db3)
    dx3 = O.dact(d_ln, pre, 1)
    torch.cuda.synchronize()
    tol = 3e-2 if dtype != torch.float32 else 2e-4
    scale = p.grad.abs().max().item()
    assert (dx.float() - p.grad).abs().max().item() <= tol * scale, ((dx.float() - p.grad).abs().max().item(), scale)
    assert (dx.float() - dx3.float()).abs().max().item() <= tol * scale
    # the fused form never rounds dy / the LayerNorm gradient to 16 bits: it is at least as close to the fp32 reference as the separate launches
    assert (dx.float() - p.grad).abs().max().item() <= (dx3.float() - p.grad).abs().max().item() * 1.5 + 1e-6
    for got, ref, sep in ((dg, g_.grad, dg3), (db, b_.grad, db3)):
        s_ = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= tol * s_ + 1e-5, ((got - ref).abs().max().item(), s_)
        assert (got - sep).abs().max().item() <= tol * s_ + 1e-5
    w = torch.zeros(2, device=DEV)
    with pytest.raises(L.MagicHipError):
        O.ln_bwd_tail(M, H, dy32, y, gamma, beta, rstd, pre, 3, dx, dg, db)             # unknown activation
    with pytest.raises(L.MagicHipError):
        L.call("magic_ln_bwd_tail", L.dt(dtype), M, 100, L.P(dy32), L.P(y), L.P(gamma), L.P(beta), L.P(rstd), L.P(pre), 1, L.P(dx), L.P(dg), L.P(db), L.stream())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K,S", [(339, 50265, 32), (70, 1000, 5), (33, 4096, 64)])
def test_split_k_into_slabs_summed_by_the_tail_is_bitwise_reproducible_and_matches_the_atomic_form(dtype, M, K, S):
    """magic_gemm with splitk = -S (round 6): every K-split STORES its partial into its own fp32 slab, magic_ln_bwd_tail adds the S slabs in slab order -- the
    MLM head's vocabulary input gradient d_hm = dlogits Wemb without fp32 atomics.  The slab sum equals the fp32 product (and the atomic split-K form) to
    summation order, splits without k-tiles store zeros (S = 64 over 64 k-tiles of 64: some are empty at K = 4096 only when S exceeds them -- (70, 1000, 5)
    has ragged last tiles), and four runs give bitwise the same dx; the unused slab count is refused."""
    H = 128
    gen = torch.Generator(DEV).manual_seed(M + S)
    ldv = (K + 7) // 8 * 8
    dlog = torch.zeros(M, ldv, device=DEV, dtype=dtype)
    dlog[:, :K] = (torch.randn(M, K, device=DEV, generator=gen) * 0.05).to(dtype)
    W = (torch.randn(K, H, device=DEV, generator=gen) * 0.1).to(dtype).contiguous()
    ref = dlog[:, :K].float() @ W.float()
    slabs = torch.full((S * M, H), float("nan"), device=DEV)               # every element must be stored by its split
    O.gemm(1, dlog, W, slabs, M, H, K, ldv, H, H, splitk=-S)
    acc = torch.zeros(M, H, device=DEV)
    O.gemm(1, dlog, W, acc, M, H, K, ldv, H, H, splitk=S, accumulate=True)
    torch.cuda.synchronize()
    assert torch.isfinite(slabs).all()
    tot = slabs.view(S, M, H).sum(0)
    scale = ref.abs().max().item()
    assert (tot - ref).abs().max().item() <= 2e-5 * scale + 1e-6 and (acc - ref).abs().max().item() <= 2e-5 * scale + 1e-6
    # the consumer: LayerNorm backward of the transform reading the slabs
    pre = torch.randn(M, H, device=DEV, generator=gen).to(dtype)
    gamma, beta = (1.0 + 0.1 * torch.randn(H, device=DEV, generator=gen)).contiguous(), (0.1 * torch.randn(H, device=DEV, generator=gen)).contiguous()
    y, rstd = torch.empty(M, H, device=DEV, dtype=dtype), torch.empty(M, device=DEV, dtype=torch.float32)
    O.ln_fwd(M, H, y, in0=F.gelu(pre.float()).to(dtype), gamma=gamma, beta=beta, eps=1e-12, rstd=rstd)
    outs = []
    for _ in range(4):
        slabs.fill_(float("nan"))
        O.gemm(1, dlog, W, slabs, M, H, K, ldv, H, H, splitk=-S)
        dx, dg, db = torch.empty(M, H, device=DEV, dtype=dtype), torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
        O.ln_bwd_tail(M, H, slabs, y, gamma, beta, rstd, pre, 1, dx, dg, db, nslab=S)
        outs.append(dx.clone())
    dx1 = torch.empty(M, H, device=DEV, dtype=dtype)
    O.ln_bwd_tail(M, H, tot.contiguous(), y, gamma, beta, rstd, pre, 1, dx1, torch.zeros(H, device=DEV), torch.zeros(H, device=DEV))
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    assert (outs[0].float() - dx1.float()).abs().max().item() <= 2e-2 * dx1.float().abs().max().item()
    with pytest.raises(L.MagicHipError):
        O.ln_bwd_tail(M, H, slabs, y, gamma, beta, rstd, pre, 1, dx1, dg, db, nslab=S + 1)        # more slabs than the buffer holds
    with pytest.raises(L.MagicHipError):
        O.gemm(1, dlog, W, slabs, M, H, K, ldv, H, H, splitk=-S, epilogue=1)                      # slabs: the plain product only


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("H,K,M,act", [(128, 128, 257, 1), (256, 256, 70, 2), (384, 128, 33, 1), (128, 768, 5, 2)])
def test_linear_act_ln_matches_torch(dtype, H, K, M, act):
    """magic_linear_act_ln: LayerNorm(act(x W^T + b)) with the pre-activation kept -- against fp32 torch on the same rounded operands"""
    gen = torch.Generator(DEV).manual_seed(H + K + M)
    x = torch.randn(M, K, device=DEV, generator=gen).to(dtype)
    W = (torch.randn(H, K, device=DEV, generator=gen) / math.sqrt(K)).to(dtype)
    b = 0.1 * torch.randn(H, device=DEV, generator=gen)
    gamma, beta = 1.0 + 0.1 * torch.randn(H, device=DEV, generator=gen), 0.1 * torch.randn(H, device=DEV, generator=gen)
    out, pre = torch.empty(M, H, device=DEV, dtype=dtype), torch.empty(M, H, device=DEV, dtype=dtype)
    rstd = torch.empty(M, device=DEV, dtype=torch.float32)
    O.linear_act_ln(x, W, b, M, act, pre, gamma, beta, 1e-12, out, rstd)
    torch.cuda.synchronize()
    z = x.float() @ W.float().t() + b
    a = F.gelu(z) if act == 1 else F.relu(z)
    ref = F.layer_norm(a, (H,), gamma, beta, 1e-12)
    tol = 3e-2 if dtype != torch.float32 else 2e-4
    assert (pre.float() - z).abs().max().item() <= tol * z.abs().max().item()
    assert (out.float() - ref).abs().max().item() <= tol * ref.abs().max().item()
    r_ref = torch.rsqrt(a.var(1, unbiased=False) + 1e-12)
    assert torch.allclose(rstd, r_ref, rtol=2e-2 if dtype != torch.float32 else 1e-3)
    with pytest.raises(L.MagicHipError):
        O.linear_act_ln(x, W, b, M, 0, pre, gamma, beta, 1e-12, out, rstd)
