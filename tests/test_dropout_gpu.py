"""Counter-based dropout (csrc/common.hpp DropDesc) -- -m gpu.

The fused kernels never store a mask: forward and backward regenerate it from (seed, site, logical element index).
These tests export the very masks through `magic_dropout` (input = ones) and replay them in plain torch / in the
fp64 oracle (oracle/model_ref.DROPOUT hook), so dropout-on training steps are checked as tightly as dropout-off ones:
kernel level (attention fwd/bwd, dense+dropout+add+LayerNorm, embedding LayerNorm+dropout) and whole model
(every output, loss term and parameter gradient of a MAKD step with the reference's dropout 0.1)."""
import math

import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import ops as O
from magic_amd.host import synth
from magic_amd.host.engine import MagicNet
from oracle import model_ref as R
from tests.test_model_gpu import KDL, RW, close, to64, view_outputs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def seed_of(a, b):
    return torch.tensor([a, b], dtype=torch.int32, device=DEV)


def export_mask(seed, p, site, shape):
    """mask * 1/(1-p) exactly as the kernels see it (row-major logical index)"""
    n = 1
    for s in shape:
        n *= s
    ones, out = torch.ones(n, device=DEV), torch.empty(n, device=DEV)
    O.dropout(ones, out, 1, n, n, (seed, p, site))
    return out.view(*shape)


def test_mask_statistics_and_determinism():
    p, n = 0.1, 1 << 20
    s1, s2 = seed_of(123, 456), seed_of(124, 456)
    m = export_mask(s1, p, 7, (n,))
    vals = torch.unique(m)
    assert vals.numel() == 2 and vals[0] == 0 and abs(vals[1].item() - 1 / (1 - p)) < 1e-6
    keep = (m > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 4 * math.sqrt(p * (1 - p) / n), keep          # 4 sigma
    assert abs(m.mean().item() - 1.0) < 2e-3                                   # expectation preserved
    assert torch.equal(m, export_mask(s1, p, 7, (n,)))                          # pure function of (seed, site, index)
    for other in (export_mask(s2, p, 7, (n,)), export_mask(s1, p, 8, (n,))):    # new seed / other site: independent masks
        agree = ((m > 0) == (other > 0)).float().mean().item()
        assert abs(agree - (p * p + (1 - p) * (1 - p))) < 3e-3, agree
    # no structure along rows of a [*, 128] activation: per-column keep rates are all near 1-p
    col = (m.view(-1, 128) > 0).float().mean(0)
    assert (col - (1 - p)).abs().max().item() < 0.02
    # p = 0 -> identity, no seed needed
    x = torch.randn(1000, device=DEV)
    assert torch.equal(O.dropout(x, torch.empty_like(x), 1, 1000, 1000, None), x)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,nh,Nq,Nk,cross", [(3, 2, 37, 37, False), (2, 4, 21, 80, True), (2, 2, 80, 17, True), (1, 2, 64, 64, False)])
def test_fused_attention_with_dropout(dtype, B, nh, Nq, Nk, cross):
    H, p, site = nh * 64, 0.1, 0xABCDEF01
    seed = seed_of(Nq * 7 + 1, Nk * 3 + 5)
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
    scale = 1 / math.sqrt(64)
    ldp = (Nk + 7) // 8 * 8
    Pm = torch.full((B, nh, Nq, ldp), 7.0, dtype=dtype, device=DEV)
    Pd = torch.full((B, nh, Nq, ldp), 7.0, dtype=dtype, device=DEV)
    ctx = torch.empty(B * Nq, H, dtype=dtype, device=DEV)
    drop = (seed, p, site)
    O.attn_fwd(q, ldq, k, v, ldkv, Pm, ldp, ctx, B, nh, Nq, Nk, H, scale, kmask=kmask, drop=drop, Pd=Pd)
    mask = export_mask(seed, p, site, (B, nh, Nq, Nk))
    heads = lambda t, N: t.float().reshape(B, N, nh, 64).transpose(1, 2)
    qh = heads(q[:, :H] if not cross else q, Nq).clone().requires_grad_(True)
    kh = heads(k[:, :H], Nk).clone().requires_grad_(True)
    vh = heads(v[:, :H], Nk).clone().requires_grad_(True)
    s = qh @ kh.transpose(-1, -2) * scale + (1 - kmask.float())[:, None, None, :] * -10000.0
    p_ref = torch.softmax(s, -1)
    pd_ref = p_ref * mask
    o_ref = pd_ref @ vh
    tp = dict(rtol=1e-4, atol=2e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=4e-3)
    to = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)

    def chk(a, b, name, **kw):
        a, b = a.float().cpu(), b.detach().float().cpu()
        assert torch.allclose(a, b, **kw), f"{name}: max|err| {(a - b).abs().max().item():.3e} (ref {b.abs().max().item():.3e})"
    chk(Pm[..., :Nk], p_ref, "P (clean)", **tp)
    chk(Pd[..., :Nk], pd_ref, "P (dropped)", **tp)
    assert (Pd[..., Nk:] == 0).all() and (Pm[..., Nk:] == 0).all()
    chk(ctx, o_ref.transpose(1, 2).reshape(B * Nq, H), "ctx", **to)
    dO = rnd(B * Nq, H).to(dtype)
    dP_extra = torch.zeros(B, nh, Nq, ldp, device=DEV)
    dP_extra[..., :Nk] = rnd(B, nh, Nq, Nk) * 0.3                    # gradient into the EXPOSED (dropped) probabilities
    ((o_ref * heads(dO, Nq)).sum() + (pd_ref * dP_extra[..., :Nk]).sum()).backward()
    if cross:
        dq, dkv = torch.zeros(B * Nq, H, dtype=dtype, device=DEV), torch.zeros(B * Nk, 2 * H, dtype=dtype, device=DEV)
        dk, dv, lddq, lddkv = dkv, dkv[:, H:], H, 2 * H
    else:
        dqkv = torch.zeros(B * Nq, 3 * H, dtype=dtype, device=DEV)
        dq, dk, dv, lddq, lddkv = dqkv, dqkv[:, H:], dqkv[:, 2 * H:], 3 * H, 3 * H
    O.attn_bwd(q, ldq, k, v, ldkv, Pm, ldp, dO, B, nh, Nq, Nk, H, scale, dP_extra, dq, lddq, dk, dv, lddkv, drop=drop)
    unheads = lambda t, N: t.transpose(1, 2).reshape(B * N, H)
    tg = dict(rtol=2e-4, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=4e-2)
    chk(dq[:, :H], unheads(qh.grad, Nq), "dQ", **tg)
    chk(dk[:, :H], unheads(kh.grad, Nk), "dK", **tg)
    chk(dv[:, :H], unheads(vh.grad, Nk), "dV", **tg)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,H,K,fused", [(100, 128, 128, True), (77, 128, 512, True), (200, 256, 256, True), (64, 256, 1024, False)])
def test_dense_dropout_add_layernorm_fwd_bwd(dtype, M, H, K, fused):
    """BertSelfOutput/BertOutput: LN(dropout(x W^T + b) + r), fused (linear_ln) or dense + ln_fwd(drop_in0), and its backward
    (ln_bwd writes the residual-branch gradient and the masked dense-branch gradient)."""
    p, site = 0.1, 4242
    seed = seed_of(M, K)
    g = torch.Generator().manual_seed(M + H + K)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    x, W, r = rnd(M, K).to(dtype), (rnd(H, K) * 0.1).to(dtype), rnd(M, H).to(dtype)
    b, gamma, beta = rnd(H) * 0.1, 1 + 0.1 * rnd(H), 0.1 * rnd(H)
    drop = (seed, p, site)
    out, rstd = torch.empty(M, H, dtype=dtype, device=DEV), torch.empty(M, device=DEV)
    if fused:
        O.linear_ln(x, W, b, M, r, gamma, beta, 1e-12, out, rstd, drop=drop)
    else:
        d = O.linear_fwd(x, W, b, M)
        O.ln_fwd(M, H, out, in0=d, in1=r, gamma=gamma, beta=beta, eps=1e-12, rstd=rstd, drop_in0=drop)
    mask = export_mask(seed, p, site, (M, H))
    dense = (x.float() @ W.float().t() + b).requires_grad_(True)
    res = r.float().clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(dense * mask + res, (H,), gamma, beta, 1e-12)
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=4e-2)
    assert torch.allclose(out.float(), ref, **tol), (out.float() - ref).abs().max().item()
    dy = rnd(M, H).to(dtype)
    ref.backward(dy.float())
    dx, dxm = torch.empty(M, H, dtype=dtype, device=DEV), torch.empty(M, H, dtype=dtype, device=DEV)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    O.ln_bwd(M, H, dy, y=out, gamma=gamma, beta=beta, rstd=rstd, dx=dx, dgamma=dg, dbeta=db, drop_dx=drop, dxm=dxm)
    tg = dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=4e-2)
    assert torch.allclose(dx.float(), res.grad, **tg), (dx.float() - res.grad).abs().max().item()
    assert torch.allclose(dxm.float(), dense.grad, **tg), (dxm.float() - dense.grad).abs().max().item()
    assert ((dxm.float() == 0) | (mask > 0)).all()          # dropped positions carry no gradient


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_embedding_layernorm_output_dropout(dtype):
    """BertEmbeddings / ImageEmbeddings: dropout(LN(sum)).  `out` keeps the clean y (the backward rebuilds xhat from it)."""
    M, H, p, site = 150, 128, 0.1, 99
    seed = seed_of(5, 6)
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    a, c = rnd(M, H).to(dtype), rnd(M, H).to(dtype)
    gamma, beta = 1 + 0.1 * rnd(H), 0.1 * rnd(H)
    drop = (seed, p, site)
    y, yd, rstd = torch.empty(M, H, dtype=dtype, device=DEV), torch.empty(M, H, dtype=dtype, device=DEV), torch.empty(M, device=DEV)
    O.ln_fwd(M, H, y, in0=a, in1=c, gamma=gamma, beta=beta, eps=1e-12, rstd=rstd, drop_out=drop, out_drop=yd)
    mask = export_mask(seed, p, site, (M, H))
    s = (a.float() + c.float()).requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(s, (H,), gamma, beta, 1e-12)
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)
    assert torch.allclose(y.float(), y_ref, **tol)
    assert torch.allclose(yd.float(), y_ref * mask, **tol)
    dy = rnd(M, H).to(dtype)
    (y_ref * mask).backward(dy.float())
    dx = torch.empty(M, H, dtype=dtype, device=DEV)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    O.ln_bwd(M, H, dy, y=y, gamma=gamma, beta=beta, rstd=rstd, dx=dx, dgamma=dg, dbeta=db, drop_dy=drop)
    tg = dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=4e-2)
    assert torch.allclose(dx.float(), s.grad, **tg), (dx.float() - s.grad).abs().max().item()


def _models(dtype, p_drop):
    from magic_amd.host.config import make_config
    from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining
    kw = dict(vocab_size=600, num_l_layers=2, num_x_layers=1, num_pano_layers=1, hidden_dropout_prob=p_drop, attention_probs_dropout_prob=p_drop)
    tcfg = make_config(256, role="teacher", **kw)
    scfg = make_config(128, role="student", teacher_hidden_size=256, kdl=KDL, **kw)
    torch.manual_seed(0)
    o_t, o_s = R.RefPretrainModel(tcfg).eval(), R.RefPretrainModel(scfg).eval()
    with torch.no_grad():
        for m in (o_t, o_s):
            for n, q in m.named_parameters():
                if n.endswith("bias"):
                    q.normal_(0, 0.02)
    g_t = GlocalTextPathCMTPreTraining.from_pretrained(None, config=tcfg, state_dict=o_t.state_dict(), device=DEV, compute_dtype=dtype)
    g_s = GlocalTextPathCMTPreTraining.from_pretrained(None, config=scfg, state_dict=o_s.state_dict(), device=DEV, compute_dtype=dtype)
    g_s.keep_mlm_logits = True
    return o_t, o_s, g_t, g_s


@pytest.mark.parametrize("task,lens", [("sap", (8, 19)), ("mlm", (8, 19)), ("sap", (130, 150))])
def test_training_step_with_dropout_matches_oracle_replaying_the_masks(task, lens):
    """The reference recipe (dropout 0.1, model.train()): engine forward/backward in fp32 vs the fp64 oracle fed with the
    engine's own masks at every dropout module.  The 130-150-token case takes the unfused attention path (keys > 128:
    GEMM + softmax + standalone dropout kernels) in the text encoder and in every node->text cross-attention."""
    p = 0.1
    o_t, o_s, g_t, g_s = _models(torch.float32, p)
    g_s.train()
    g_t.eval()
    batch = synth.make_batch(task, batch_size=5 if lens[1] < 100 else 3, seed=33, vocab=600, min_len=lens[0], max_len=lens[1], min_steps=2, max_steps=4)
    rw = torch.tensor(RW, dtype=torch.float64)
    o_t, o_s = o_t.double(), o_s.double()
    b64 = to64(batch)
    with torch.no_grad():
        gt = g_t(batch, task, compute_loss=False, return_outputs=True)
        assert g_t.net.drop is None                               # frozen teacher: never dropped
        ot = o_t(b64, task, compute_loss=True)["outputs"]
    plan = gt["plan"]
    g_s.store.zero_grad()
    got = g_s(batch, task, compute_loss=True, teacher_outputs=gt, rw=RW, plan=plan)
    seed, ph, pa = g_s.net.drop
    assert ph == pytest.approx(p) and pa == pytest.approx(p)
    used = []

    def hook(site, x):
        used.append(site)
        m = export_mask(seed, p, MagicNet.site_id(site), tuple(x.shape)).cpu().double()
        return x * m
    R.DROPOUT = hook
    try:
        want = o_s(b64, task, compute_loss=True, teacher_outputs=ot, rw=rw)
    finally:
        R.DROPOUT = None
    assert len(used) == len(set(used)) and len(used) >= 10, used      # every module dropped once, with its own site id
    assert len({MagicNet.site_id(s) for s in used}) == len(used)
    want["loss"].backward()
    for k, v in view_outputs(got["outputs"], plan, 128).items():
        close(v, want["outputs"][k], f"student {k}", 2e-4, 2e-5)
    if task == "sap":
        for k in ("global_logits", "local_logits", "fused_logits"):
            a, b = got["outputs"][k].cpu(), want["outputs"][k]
            close(torch.nan_to_num(a, neginf=0), torch.nan_to_num(b, neginf=0), k, 1e-4, 1e-5)
            assert torch.equal(a.argmax(1), b.argmax(1)), f"{k} argmax"
    close(got["supervised_loss"], want["supervised_loss"], "supervised loss", 1e-4, 1e-6)
    for k, v in want["kdl_terms"].items():
        close(got["kdl_terms"][k], v, f"kd term {k}", 2e-4, 1e-7)
    close(got["loss"], want["loss"], "total loss", 1e-4, 1e-6)
    got["loss"].backward()
    torch.cuda.synchronize()
    params = dict(g_s.named_parameters())
    gmax = max(q.grad.abs().max().item() for q in o_s.parameters() if q.grad is not None)
    n_checked = 0
    for name, q in o_s.named_parameters():
        g = params[name].grad
        if q.grad is None:
            assert g.abs().max().item() == 0.0, name
            continue
        scale = q.grad.abs().max().item()
        close(g, q.grad, f"grad {name}", 2e-3, 2e-4 * scale + 2e-6 * gmax)
        n_checked += 1
    assert n_checked > 60
    # a second forward draws a new seed -> different masks -> different loss; eval() turns dropout off
    got2 = g_s(batch, task, compute_loss=True, teacher_outputs=gt, rw=RW, plan=plan)
    assert not torch.equal(g_s.net.drop[0], seed)
    assert abs(float(got2["loss"]) - float(got["loss"])) > 1e-6
    g_s.eval()
    with torch.no_grad():
        g_s(batch, task, compute_loss=False, return_outputs=True, plan=plan)
    assert g_s.net.drop is None


def test_bf16_dropout_step_tracks_fp32_and_graph_replay_draws_fresh_masks():
    from magic_amd.host.plan import build_plan
    from magic_amd.host.trainer import PretrainStep
    _, _, g_t, g_s = _models(torch.bfloat16, 0.1)
    g_s.train()
    batch = synth.make_batch("sap", batch_size=4, seed=5, vocab=600, min_len=8, max_len=19, min_steps=2, max_steps=4)
    plan = build_plan(batch, "sap", torch.device(DEV))
    batch = synth.batch_to(batch, torch.device(DEV))        # capture() needs a device-resident batch
    tr = PretrainStep(g_s, g_t, warmup_steps=10, num_train_steps=100)
    tr.step(batch, "sap", plan=plan)                        # eager warm-up
    cs = tr.capture(batch, "sap", plan)
    losses, seeds = [], []
    for _ in range(4):
        out = tr.replay(cs)
        torch.cuda.synchronize()
        losses.append(float(out["loss"]))
        seeds.append(g_s.net.drop[0].clone())
    assert all(math.isfinite(v) for v in losses)
    assert len({tuple(s.tolist()) for s in seeds}) == 4     # the in-graph generator advances on every replay
    assert torch.isfinite(g_s.store.flat).all()
