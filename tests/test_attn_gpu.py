"""Fused attention kernels (csrc/attention.hip) vs plain fp32 torch attention + autograd (-m gpu)."""
import math

import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ref_attention(q, k, v, kmask, dist, sw, sb, scale):
    s = q @ k.transpose(-1, -2) * scale
    if kmask is not None:
        s = s + (1 - kmask.float())[:, None, None, :] * -10000.0
    if dist is not None:
        s = s + (sw * dist + sb)[:, None]
    p = torch.softmax(s, -1)
    return p, p @ v


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,nh,Nq,Nk,cross,use_dist", [(3, 2, 37, 37, False, False), (2, 2, 18, 18, False, True), (2, 4, 21, 80, True, False),
                                                        (2, 2, 80, 17, True, False), (1, 2, 64, 64, False, False), (2, 2, 100, 128, True, False)])
def test_fused_attention_fwd_bwd(dtype, B, nh, Nq, Nk, cross, use_dist):
    if dtype == torch.float32 and max(Nq, Nk) > 64 and not O.attn_supported(dtype, Nq, Nk, True):
        pytest.skip("fp32 backward does not fit LDS for this shape (engine falls back to the unfused path)")
    H = nh * 64
    g = torch.Generator().manual_seed(Nq * 131 + Nk)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    if cross:
        qb, kvb = rnd(B * Nq, H).to(dtype), rnd(B * Nk, 2 * H).to(dtype)
        q, k, v, ldq, ldkv = qb, kvb, kvb[:, H:], H, 2 * H
    else:
        qkv = rnd(B * Nq, 3 * H).to(dtype)
        q, k, v, ldq, ldkv = qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H
    kmask = torch.ones(B, Nk, dtype=torch.uint8, device=DEV)
    kmask[0, Nk - 3:] = 0
    if B > 1:
        kmask[1, 1] = 0
    dist = (rnd(B, Nq, Nk).abs() * 4).contiguous() if use_dist else None
    sw, sb = torch.tensor([0.2], device=DEV), torch.tensor([-0.1], device=DEV)
    scale = 1 / math.sqrt(64)
    ldp = (Nk + 7) // 8 * 8
    Pm = torch.full((B, nh, Nq, ldp), 7.0, dtype=dtype, device=DEV)
    ctx = torch.empty(B * Nq, H, dtype=dtype, device=DEV)
    O.attn_fwd(q, ldq, k, v, ldkv, Pm, ldp, ctx, B, nh, Nq, Nk, H, scale, kmask=kmask, dist=dist, sprel_w=sw if use_dist else None,
               sprel_b=sb if use_dist else None)
    heads = lambda t, N: t.float().reshape(B, N, nh, 64).transpose(1, 2)
    qh = heads(q[:, :H] if not cross else q, Nq).clone().requires_grad_(True)
    kh = heads(k[:, :H], Nk).clone().requires_grad_(True)
    vh = heads(v[:, :H], Nk).clone().requires_grad_(True)
    swr, sbr = sw.clone().requires_grad_(True), sb.clone().requires_grad_(True)
    p_ref, o_ref = ref_attention(qh, kh, vh, kmask, dist, swr, sbr, scale)
    tp = dict(rtol=1e-4, atol=2e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=4e-3)
    to = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)

    def chk(a, b, name, **kw):
        a, b = a.float().cpu(), b.detach().float().cpu()
        assert torch.allclose(a, b, **kw), f"{name}: max|err| {(a - b).abs().max().item():.3e} (ref {b.abs().max().item():.3e})"
    chk(Pm[..., :Nk], p_ref, "P", **tp)
    assert (Pm[..., Nk:] == 0).all()
    chk(ctx, o_ref.transpose(1, 2).reshape(B * Nq, H), "ctx", **to)
    # backward, with an extra gradient flowing into P (attention distillation)
    dO = rnd(B * Nq, H).to(dtype)
    dP_extra = torch.zeros(B, nh, Nq, ldp, device=DEV)
    dP_extra[..., :Nk] = rnd(B, nh, Nq, Nk) * 0.3
    loss = (o_ref * heads(dO, Nq)).sum() + (p_ref * dP_extra[..., :Nk]).sum()
    loss.backward()
    if cross:
        dq, dkv = torch.zeros(B * Nq, H, dtype=dtype, device=DEV), torch.zeros(B * Nk, 2 * H, dtype=dtype, device=DEV)
        dk, dv, lddq, lddkv = dkv, dkv[:, H:], H, 2 * H
    else:
        dqkv = torch.zeros(B * Nq, 3 * H, dtype=dtype, device=DEV)
        dq, dk, dv, lddq, lddkv = dqkv, dqkv[:, H:], dqkv[:, 2 * H:], 3 * H, 3 * H
    dsw, dsb = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    # the engine feeds the backward with the P the forward stored (bf16-rounded in bf16 mode)
    O.attn_bwd(q, ldq, k, v, ldkv, Pm, ldp, dO, B, nh, Nq, Nk, H, scale, dP_extra, dq, lddq, dk, dv, lddkv,
               dist=dist, dsprel_w=dsw if use_dist else None, dsprel_b=dsb if use_dist else None)
    O.flush_part_jobs()        # (inside a backward pass the graph-distance bias gradients go through partial rows: the flush adds them up -- a no-op outside one)
    unheads = lambda t, N: t.transpose(1, 2).reshape(B * N, H)
    tg = dict(rtol=2e-4, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=4e-2)
    chk(dq[:, :H], unheads(qh.grad, Nq), "dQ", **tg)
    chk(dk[:, :H], unheads(kh.grad, Nk), "dK", **tg)
    chk(dv[:, :H], unheads(vh.grad, Nk), "dV", **tg)
    if use_dist:
        ts = dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=5e-2)
        chk(dsw, swr.grad, "d sprel w", **ts)
        chk(dsb, sbr.grad, "d sprel b", **ts)


def test_unsupported_shapes_are_reported_not_launched():
    assert O.attn_supported(torch.bfloat16, 40, 512, False)              # K/V-tiled two-pass forward
    assert not O.attn_supported(torch.bfloat16, 40, 513, False)
    assert not O.attn_supported(torch.bfloat16, 40, 512, True)           # the backward for long keys stays GEMM + softmax
    assert O.attn_supported(torch.bfloat16, 80, 80, True)
    assert not O.attn_supported(torch.float32, 128, 128, True)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,nh,Nq,Nk,cross,use_dist,p_drop", [(2, 2, 40, 129, True, False, 0.0), (1, 2, 200, 200, False, False, 0.1),
                                                               (2, 4, 38, 512, True, False, 0.0), (1, 2, 70, 300, True, True, 0.1),
                                                               (1, 2, 512, 512, False, False, 0.0)])
def test_kv_tiled_forward_for_long_keys(dtype, B, nh, Nq, Nk, cross, use_dist, p_drop):
    """128 < Nk <= 512 (RxR-length instructions): two-pass softmax over 128-key tiles.  One key is spiked against one query in the
    LAST tile so the running maximum jumps late (the rescale of the running sum is exercised, not just carried); masks put -10000
    on keys of the first and the last tile."""
    H = nh * 64
    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    if cross:
        qb, kvb = rnd(B * Nq, H), rnd(B * Nk, 2 * H)
        kvb[Nk - 2, :H] = qb[3, :H] * 4.0                   # batch 0: key Nk-2 aligned with query 3 -> a late, large maximum
        qb, kvb = qb.to(dtype), kvb.to(dtype)
        q, k, v, ldq, ldkv = qb, kvb, kvb[:, H:], H, 2 * H
    else:
        qkv = rnd(B * Nq, 3 * H)
        qkv[Nk - 2, H:2 * H] = qkv[3, :H] * 4.0
        qkv = qkv.to(dtype)
        q, k, v, ldq, ldkv = qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H
    kmask = torch.ones(B, Nk, dtype=torch.uint8, device=DEV)
    kmask[0, Nk - 5:Nk - 3] = 0
    kmask[B - 1, 1] = 0
    dist = (rnd(B, Nq, Nk).abs() * 4).contiguous() if use_dist else None
    sw, sb = torch.tensor([0.2], device=DEV), torch.tensor([-0.1], device=DEV)
    scale = 1 / math.sqrt(64)
    ldp = (Nk + 7) // 8 * 8
    Pm = torch.full((B, nh, Nq, ldp), 7.0, dtype=dtype, device=DEV)
    Pd = torch.full((B, nh, Nq, ldp), 7.0, dtype=dtype, device=DEV) if p_drop > 0 else None
    ctx = torch.empty(B * Nq, H, dtype=dtype, device=DEV)
    seed = torch.tensor([11, 22], dtype=torch.int32, device=DEV)
    drop = (seed, p_drop, 4242) if p_drop > 0 else None
    O.attn_fwd(q, ldq, k, v, ldkv, Pm, ldp, ctx, B, nh, Nq, Nk, H, scale, kmask=kmask, dist=dist, sprel_w=sw if use_dist else None,
               sprel_b=sb if use_dist else None, drop=drop, Pd=Pd)
    torch.cuda.synchronize()
    heads = lambda t, N: t.float().reshape(B, N, nh, 64).transpose(1, 2)
    qh, kh, vh = heads(q[:, :H] if not cross else q, Nq), heads(k[:, :H], Nk), heads(v[:, :H], Nk)
    p_ref, _ = ref_attention(qh, kh, vh, kmask, dist, sw, sb, scale)
    tp = dict(rtol=2e-4, atol=2e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=4e-3)
    to = dict(rtol=2e-4, atol=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    assert torch.allclose(Pm[..., :Nk].float(), p_ref, **tp), (Pm[..., :Nk].float() - p_ref).abs().max().item()
    assert (Pm[..., Nk:] == 0).all() and torch.allclose(Pm[..., :Nk].float().sum(-1), torch.ones(B, nh, Nq, device=DEV), atol=2e-2)
    assert p_ref[0, :, 3, Nk - 2].min().item() > 0.5            # the spiked key really dominates its row
    if p_drop > 0:
        ones = torch.ones(B * nh * Nq * Nk, device=DEV)
        mask = torch.empty_like(ones)
        O.dropout(ones, mask, B * nh * Nq, Nk, Nk, drop)        # the kernel's own mask over the logical [B, nh, Nq, Nk] tensor
        want_pd = Pm[..., :Nk].float() * mask.view(B, nh, Nq, Nk)
        assert torch.allclose(Pd[..., :Nk].float(), want_pd, rtol=1e-2, atol=1e-3) and (Pd[..., Nk:] == 0).all()
        o_ref = Pd[..., :Nk].float() @ vh
    else:
        o_ref = Pm[..., :Nk].float() @ vh
    got = ctx.float()
    want = o_ref.transpose(1, 2).reshape(B * Nq, H)
    assert torch.allclose(got, want, **to), (got - want).abs().max().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,nh,Nq,Nk,cross,p_drop,acc", [(2, 4, 38, 512, True, 0.0, False), (1, 2, 200, 200, False, 0.1, False), (2, 2, 64, 129, True, 0.0, True),
                                                          (1, 2, 512, 512, False, 0.0, False), (2, 2, 39, 486, True, 0.1, True), (3, 2, 16, 300, True, 0.1, False)])
def test_key_split_backward_for_long_keys(dtype, B, nh, Nq, Nk, cross, p_drop, acc):
    """magic_attn_bwd_ks (128 < Nk <= 512, 16-bit storage): dQ / dK / dV of the fused key-split backward against fp32 autograd of the same
    attention, fed -- as the engine feeds it -- with the P and the output the key-split forward stored; with dropout the reference uses the
    kernel's own mask; acc: dK / dV are added to what the buffers hold (the per-episode K/V cache gradient collected over the steps);
    Nq > 64 loops over query tiles inside the workgroup."""
    assert O.attn_bwd_ks_ok(dtype, Nq, Nk)
    H = nh * 64
    g = torch.Generator().manual_seed(Nq * 13 + Nk)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    if cross:
        qb, kvb = rnd(B * Nq, H).to(dtype), rnd(B * Nk, 2 * H).to(dtype)
        q, k, v, ldq, ldkv = qb, kvb, kvb[:, H:], H, 2 * H
    else:
        qkv = rnd(B * Nq, 3 * H).to(dtype)
        q, k, v, ldq, ldkv = qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H
    kmask = torch.ones(B, Nk, dtype=torch.uint8, device=DEV)
    kmask[0, Nk - 5:] = 0
    kmask[B - 1, 1] = 0
    scale = 1 / math.sqrt(64)
    ldp = (Nk + 7) // 8 * 8
    Pm = torch.full((B, nh, Nq, ldp), 7.0, dtype=dtype, device=DEV)
    Pd = torch.full((B, nh, Nq, ldp), 7.0, dtype=dtype, device=DEV) if p_drop > 0 else None
    ctx = torch.empty(B * Nq, H, dtype=dtype, device=DEV)
    seed = torch.tensor([5, 77], dtype=torch.int32, device=DEV)
    drop = (seed, p_drop, 999) if p_drop > 0 else None
    O.attn_fwd(q, ldq, k, v, ldkv, Pm, ldp, ctx, B, nh, Nq, Nk, H, scale, kmask=kmask, drop=drop, Pd=Pd)
    heads = lambda t, N: t.float().reshape(B, N, nh, 64).transpose(1, 2)
    qh = heads(q[:, :H] if not cross else q, Nq).clone().requires_grad_(True)
    kh = heads(k[:, :H], Nk).clone().requires_grad_(True)
    vh = heads(v[:, :H], Nk).clone().requires_grad_(True)
    p_ref, _ = ref_attention(qh, kh, vh, kmask, None, None, None, scale)
    if p_drop > 0:
        ones = torch.ones(B * nh * Nq * Nk, device=DEV)
        mask = torch.empty_like(ones)
        O.dropout(ones, mask, B * nh * Nq, Nk, Nk, drop)
        p_used = p_ref * mask.view(B, nh, Nq, Nk)
    else:
        p_used = p_ref
    o_ref = p_used @ vh
    assert torch.allclose(ctx.float(), o_ref.detach().transpose(1, 2).reshape(B * Nq, H), rtol=2e-2, atol=2e-2)
    dO = rnd(B * Nq, H).to(dtype)
    (o_ref * heads(dO, Nq)).sum().backward()
    if cross:
        dq, dkv = torch.zeros(B * Nq, H, dtype=dtype, device=DEV), torch.zeros(B * Nk, 2 * H, dtype=dtype, device=DEV)
        dk, dv, lddq, lddkv = dkv, dkv[:, H:], H, 2 * H
    else:
        dqkv = torch.zeros(B * Nq, 3 * H, dtype=dtype, device=DEV)
        dq, dk, dv, lddq, lddkv = dqkv, dqkv[:, H:], dqkv[:, 2 * H:], 3 * H, 3 * H
    base_k = base_v = 0.0
    if acc:
        base = (rnd(B * Nk, 2 * H) * 0.5).to(dtype)
        dk[:, :H] = base[:, :H]
        dv[:, :H] = base[:, H:]
        base_k, base_v = base[:, :H].float(), base[:, H:].float()
    O.attn_bwd_ks(q, ldq, k, v, ldkv, Pm, ldp, ctx, dO, B, nh, Nq, Nk, H, scale, dq, lddq, dk, dv, lddkv, accumulate_kv=acc, drop=drop)
    torch.cuda.synchronize()
    unheads = lambda t, N: t.transpose(1, 2).reshape(B * N, H)

    def chk(a, b, name):
        a, b = a.float().cpu(), b.detach().float().cpu()
        err = (a - b).abs().max().item()
        assert torch.allclose(a, b, rtol=3e-2, atol=4e-2 + (3e-2 if acc else 0)), f"{name}: max|err| {err:.3e} (ref {b.abs().max().item():.3e})"
        cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
        assert cos > 0.999, f"{name}: cosine {cos}"
    chk(dq[:, :H], unheads(qh.grad, Nq), "dQ")
    chk(dk[:, :H], unheads(kh.grad, Nk) + base_k, "dK")
    chk(dv[:, :H], unheads(vh.grad, Nk) + base_v, "dV")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,H,K", [(100, 128, 128), (77, 128, 512), (200, 256, 256), (64, 256, 1024), (33, 384, 384)])
def test_fused_linear_residual_layernorm(dtype, M, H, K):
    g = torch.Generator().manual_seed(M + H + K)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    x, W, r = rnd(M, K).to(dtype), (rnd(H, K) * 0.1).to(dtype), rnd(M, H).to(dtype)
    b, gamma, beta = rnd(H) * 0.1, 1 + 0.1 * rnd(H), 0.1 * rnd(H)
    ref = torch.nn.functional.layer_norm(x.float() @ W.float().t() + b + r.float(), (H,), gamma, beta, 1e-12)
    out, rstd = torch.empty(M, H, dtype=dtype, device=DEV), torch.empty(M, device=DEV)
    O.linear_ln(x, W, b, M, r, gamma, beta, 1e-12, out, rstd)
    pre = x.float() @ W.float().t() + b + r.float()
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)
    assert torch.allclose(out.float(), ref, **tol), (out.float() - ref).abs().max().item()
    assert torch.allclose(rstd, 1 / torch.sqrt(pre.var(-1, unbiased=False) + 1e-12), rtol=1e-3 if dtype == torch.float32 else 2e-2, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,H,K,res,p", [(100, 128, 512, True, 0.0), (77, 128, 384, True, 0.1), (200, 256, 256, False, 0.0),
                                         (33, 128, 128, True, 0.0), (64, 256, 512, True, 0.1),
                                         (3840, 128, 512, True, 0.1), (130, 128, 264, False, 0.0), (8200, 128, 384, True, 0.0)])
def test_fused_input_gradient_gemm_plus_layernorm_backward(dtype, M, H, K, res, p):
    """magic_linear_lnbwd: (x @ W + residual) pushed through the backward of the LayerNorm whose output is y, vs autograd."""
    g = torch.Generator().manual_seed(M + H + K)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    x, W = (rnd(M, K) * 0.5).to(dtype), (rnd(K, H) * 0.1).to(dtype)
    R = rnd(M, H).to(dtype) if res else None
    gamma, beta = 1 + 0.1 * rnd(H), 0.1 * rnd(H)
    pre = rnd(M, H).requires_grad_(True)                      # the LayerNorm's input in the forward
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(pre, (H,), gm, bt, 1e-12)
    y = y_ref.detach().to(dtype)
    rstd = torch.rsqrt(pre.detach().var(-1, unbiased=False) + 1e-12)
    v = x.float() @ W.float() + (R.float() if res else 0)
    y_ref.backward(v)
    dx, dxm = torch.empty(M, H, dtype=dtype, device=DEV), torch.empty(M, H, dtype=dtype, device=DEV)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    seed = torch.tensor([3, 4], dtype=torch.int32, device=DEV)
    drop = (seed, p, 777) if p > 0 else None
    O.linear_lnbwd(x, W, M, R, y, gamma, beta, rstd, dx, dg, db, drop=drop, dxm=dxm if p > 0 else None)
    tol = dict(rtol=2e-3, atol=2e-4) if dtype == torch.float32 else dict(rtol=4e-2, atol=6e-2)
    assert torch.allclose(dx.float(), pre.grad, **tol), (dx.float() - pre.grad).abs().max().item()
    sc = max(1.0, gm.grad.abs().max().item())
    tp = dict(rtol=2e-3, atol=2e-3 * sc) if dtype == torch.float32 else dict(rtol=5e-2, atol=5e-2 * sc)
    assert torch.allclose(dg, gm.grad, **tp), (dg - gm.grad).abs().max().item()
    assert torch.allclose(db, bt.grad, **tp), (db - bt.grad).abs().max().item()
    if p > 0:
        ones, mask = torch.ones(M * H, device=DEV), torch.empty(M * H, device=DEV)
        O.dropout(ones, mask, 1, M * H, M * H, drop)
        assert torch.allclose(dxm.float(), (dx.float() * mask.view(M, H)).to(dtype).float(), rtol=1e-2, atol=1e-3)
