"""Row-block pipeline kernel (csrc/rowblock.hip) vs plain fp32 torch, stage by stage (-m gpu)."""
from types import SimpleNamespace

import pytest
import torch
import torch.nn.functional as F

import magic_amd  # noqa: F401
from magic_amd.host import lib as L
from magic_amd.host import ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def lin(g, N, K, dtype, scale=0.08):
    W = (torch.randn(N, K, generator=g) * scale).to(DEV).to(dtype)
    b = (torch.randn(N, generator=g) * 0.1).to(DEV)
    return SimpleNamespace(W=W, b=b, N=N, K=K)


def export_mask(seed, p, site, shape):
    n = shape[0] * shape[1]
    ones, out = torch.ones(n, device=DEV), torch.empty(n, device=DEV)
    O.dropout(ones, out, 1, n, n, (seed, p, site))
    return out.view(*shape)


def run_chain(dtype, M, H, I, N4, p_drop, g):
    """the self-layer tail: o-proj+LN, FFN1+GELU, FFN2+LN, next projection (N4 columns; 0 = absent)"""
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    ctx, x = rnd(M, H).to(dtype), rnd(M, H).to(dtype)
    lo, l1, l2 = lin(g, H, H, dtype), lin(g, I, H, dtype), lin(g, H, I, dtype, 0.04)
    l4 = lin(g, N4, H, dtype) if N4 else None
    g1, b1, g2, b2 = 1 + 0.1 * rnd(H), 0.1 * rnd(H), 1 + 0.1 * rnd(H), 0.1 * rnd(H)
    new = lambda *s: torch.full(s, 7.0, dtype=dtype, device=DEV)
    a, z, gg, out = new(M, H), new(M, I), new(M, I), new(M, H)
    r1, r2 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    qn = new(M, N4) if N4 else None
    seed = torch.tensor([11, 22], dtype=torch.int32, device=DEV)
    d1, d2 = ((seed, p_drop, 101), (seed, p_drop, 202)) if p_drop > 0 else (None, None)
    stages = [O.rb_ln(lo, x, g1, b1, 1e-12, a, r1, drop=d1), O.rb_act(l1, gg, pre=z),
              O.rb_ln(l2, None, g2, b2, 1e-12, out, r2, drop=d2, res_stage=0)]
    if N4:
        stages.append(O.rb_lin(l4, qn))
    O.rowblock_fwd(ctx, M, stages)
    # reference, following the kernel's rounding points (stage outputs are stored in `dtype` and re-read)
    f = lambda t: t.float()
    m1 = export_mask(seed, p_drop, 101, (M, H)) if p_drop > 0 else 1.0
    m2 = export_mask(seed, p_drop, 202, (M, H)) if p_drop > 0 else 1.0
    pre1 = (f(ctx) @ f(lo.W).t() + lo.b) * m1 + f(x)
    a_ref = F.layer_norm(pre1, (H,), g1, b1, 1e-12)
    a_q = a_ref.to(dtype).float()
    z_ref = a_q @ f(l1.W).t() + l1.b
    g_ref = F.gelu(z_ref)
    g_q = g_ref.to(dtype).float()
    pre2 = (g_q @ f(l2.W).t() + l2.b) * m2 + a_q
    out_ref = F.layer_norm(pre2, (H,), g2, b2, 1e-12)
    out_q = out_ref.to(dtype).float()
    tol = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)

    def chk(got, want, name):
        err = (got.float() - want).abs().max().item()
        assert torch.allclose(got.float(), want, **tol), f"{name}: max|err| {err:.3e} (ref max {want.abs().max().item():.3e})"
    chk(a, a_ref, "a")
    chk(z, z_ref, "z (pre-GELU)")
    chk(gg, g_ref, "g")
    chk(out, out_ref, "out")
    rt = dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=1e-2)
    assert torch.allclose(r1, torch.rsqrt(pre1.var(-1, unbiased=False) + 1e-12), **rt)
    assert torch.allclose(r2, torch.rsqrt(pre2.var(-1, unbiased=False) + 1e-12), **rt)
    if N4:
        chk(qn, out_q @ f(l4.W).t() + l4.b, "next projection")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,H,I,N4,p", [(100, 128, 512, 384, 0.0), (33, 128, 512, 0, 0.0), (257, 128, 512, 384, 0.1),
                                        (96, 256, 1024, 768, 0.0), (70, 256, 1024, 256, 0.1)])
def test_self_layer_tail_chain(dtype, M, H, I, N4, p):
    if not O.rowblock_fits(dtype, H, I):
        assert dtype == torch.float32 and H == 256          # fp32 images of the 1024-wide GELU output exceed LDS
        pytest.skip("does not fit LDS: the engine uses the separate-launch path")
    run_chain(dtype, M, H, I, N4, p, torch.Generator().manual_seed(M + H))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_stage_chain_and_pairing(dtype):
    """cross-layer head of the chain: o-proj+LN then the cross-attention query projection; two such chains issued from
    two host threads pair into one launch (lib.lockstep) and give the same results as when launched alone."""
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    H = 128

    def make(M):
        ctx, x = rnd(M, H).to(dtype), rnd(M, H).to(dtype)
        lo, lq = lin(g, H, H, dtype), lin(g, H, H, dtype)
        gm, bt = 1 + 0.1 * rnd(H), 0.1 * rnd(H)
        return ctx, x, lo, lq, gm, bt

    def launch(args, M):
        ctx, x, lo, lq, gm, bt = args
        s, q, r = (torch.empty(M, H, dtype=dtype, device=DEV) for _ in range(2)) if False else (None, None, None)
        s = torch.empty(M, H, dtype=dtype, device=DEV)
        q = torch.empty(M, H, dtype=dtype, device=DEV)
        r = torch.empty(M, device=DEV)
        O.rowblock_fwd(ctx, M, [O.rb_ln(lo, x, gm, bt, 1e-12, s, r), O.rb_lin(lq, q)])
        return s, q

    A, B = make(75), make(40)
    sa, qa = launch(A, 75)
    sb, qb = launch(B, 40)
    (sa2, qa2), (sb2, qb2) = L.lockstep(lambda: launch(A, 75), lambda: launch(B, 40))
    torch.cuda.synchronize()
    for u, v in ((sa, sa2), (qa, qa2), (sb, sb2), (qb, qb2)):
        assert torch.equal(u, v)
    ctx, x, lo, lq, gm, bt = A
    s_ref = F.layer_norm(ctx.float() @ lo.W.float().t() + lo.b + x.float(), (H,), gm, bt, 1e-12)
    tol = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)
    assert torch.allclose(sa.float(), s_ref, **tol)
    assert torch.allclose(qa.float(), s_ref.to(dtype).float() @ lq.W.float().t() + lq.b, **tol)


def test_bad_chains_are_rejected_not_launched():
    g = torch.Generator().manual_seed(1)
    dtype, M, H = torch.bfloat16, 64, 128
    x = torch.randn(M, H, generator=g).to(DEV).to(dtype)
    l_bad = lin(g, 100, H, dtype)                       # N % 64 != 0
    out = torch.empty(M, 100, dtype=dtype, device=DEV)
    with pytest.raises(L.MagicHipError):
        O.rowblock_fwd(x, M, [O.rb_lin(l_bad, out)])
    l_ok, l_mis = lin(g, H, H, dtype), lin(g, H, 256, dtype)     # K of stage 1 != N of stage 0
    gm = torch.ones(H, device=DEV)
    o1, o2, r = torch.empty(M, H, dtype=dtype, device=DEV), torch.empty(M, H, dtype=dtype, device=DEV), torch.empty(M, device=DEV)
    with pytest.raises(L.MagicHipError):
        O.rowblock_fwd(x, M, [O.rb_ln(l_ok, x, gm, gm, 1e-12, o1, r), O.rb_lin(l_mis, o2)])
    assert not O.rowblock_fits(torch.bfloat16, 384, 1536)         # LayerNorm width > 256: separate launches


@pytest.mark.parametrize("task", ["sap", "mlm"])
def test_engine_with_rowblock_chains_matches_oracle(task, monkeypatch):
    """The chains are opt-in (measured slower than separate launches); with them switched on the whole
    model must still match the oracle: student H=128 runs its self/cross-layer tails through the chains in fp32, the
    teacher (H=256, fp32 images do not fit LDS) keeps the separate-launch path."""
    from tests import test_model_gpu as TM
    monkeypatch.setattr(O, "ROWBLOCK", True)
    calls = {"n": 0}
    real = O.rowblock_fwd

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    monkeypatch.setattr(O, "rowblock_fwd", counting)
    TM.test_fp32_forward_loss_and_gradients_match_oracle(task)
    assert calls["n"] >= 4          # text + panorama + cross-layer chains really ran
