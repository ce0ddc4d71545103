"""Whole-encoder launch (csrc/encoder.hip) against the per-op kernels it replaces (-m gpu): same inputs, same weights, same dropout
seed -> every tensor the backward reads (qkv, probabilities clean and dropped, context, both LayerNorm outputs + rstd, FFN
pre-activation, GELU output) must agree to bf16 rounding (the two paths share rounding points and dropout masks; only fp32
summation order differs), and a whole training step must give the same loss and gradients."""
import pytest
import torch
import torch.nn.functional as F

import magic_amd  # noqa: F401
from magic_amd.host import ops as O
from magic_amd.host import synth
from magic_amd.host.config import make_config
from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining
from magic_amd.host.plan import build_plan
from tests.test_model_gpu import KDL

pytestmark = pytest.mark.gpu
DEV = "cuda"


def student(p_drop=0.0, **kw):
    cfg = make_config(128, role="student", teacher_hidden_size=256, kdl=KDL, hidden_dropout_prob=p_drop, attention_probs_dropout_prob=p_drop, **kw)
    m = GlocalTextPathCMTPreTraining(cfg, device=DEV, compute_dtype=torch.bfloat16, seed=3)
    with torch.no_grad():          # non-trivial biases / LayerNorm parameters
        g = torch.Generator().manual_seed(1)
        for n, p in m.named_parameters():
            if n.endswith("bias") and p.dim() == 1:
                p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(DEV))
            if "LayerNorm.weight" in n:
                p.add_((torch.randn(p.shape, generator=g) * 0.1).to(DEV))
    m.store.shadow_clean = False
    return m


def ulp_close(a, b, name, ulps=2.0, floor=2e-3, frac_ok=1e-2):
    """the two paths round at the same points, so almost every element agrees to 1-2 bf16 ulps; a one-ulp flip early on moves a few
    elements of the deeper layers a little further (both results are equally far from the exact value)"""
    a, b = a.float(), b.float()
    tol = ulps * 2.0 ** -8 * b.abs().clamp_min(floor)
    bad = ((a - b).abs() > tol)
    frac = bad.float().mean().item()
    worst = ((a - b).abs() / b.abs().clamp_min(max(0.05, 0.5 * b.pow(2).mean().sqrt().item()))).max().item()      # floor: half the tensor's rms
    rel_l2 = ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    # the first layers agree to 1-2 ulps almost everywhere; six layers down the one-ulp flips have spread, so the bound there is on the
    # error NORM (a bf16 rounding alone is ~2e-3 relative) and on the worst element
    assert (frac < frac_ok or rel_l2 < 6e-3) and worst < 0.1, \
        f"{name}: {frac:.4%} of elements off by more than {ulps} bf16 ulps, rel L2 {rel_l2:.2e} (max |d| {(a - b).abs().max().item():.3e}, worst rel {worst:.3f})"


@pytest.mark.parametrize("row_split", [True, False])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("max_len", [80, 41, 19])
def test_fused_encoders_save_what_the_unfused_kernels_save(p_drop, max_len, row_split, monkeypatch):
    """row_split: the round-3 form (one workgroup per (sample, 16-row tile), layer outputs handed between the tiles of a sample inside the
    launch) and the per-sample form, each against the per-op kernels"""
    monkeypatch.setattr(O, "ENC_ROW_SPLIT", row_split)       # (panoramas: 3 tiles -> per-sample form inside the same launch; MAGIC_ENC_RS_ALL=1 splits them too)
    m = student(p_drop)
    m.train()
    batch = synth.make_batch("sap", batch_size=7, seed=5, step=0, max_len=max_len, min_len=min(12, max_len), dup_view_prob=0.3)
    plan = build_plan(batch, "sap", torch.device(DEV))
    inp = m._inputs(batch, plan)
    m.store.sync_shadow()
    seed = torch.tensor([12345, 678], dtype=torch.int32, device=DEV)
    res = {}
    for fused in (False, True):
        O.FUSED_ENC = fused
        try:
            m.net.set_dropout(seed if p_drop > 0 else None, p_drop, p_drop)
            assert m.net.enc_ok(plan["L"], 6) == fused
            ct = m.net.text_fwd(plan)
            cp = m.net.pano_fwd(plan, inp.feats, inp.loc)
            torch.cuda.synchronize()
            res[fused] = (ct, cp)
            if fused and row_split:
                assert int(O.ENC_SYNC_LAST[0][0]) == 0, "a bounded hand-off wait gave up"
        finally:
            O.FUSED_ENC = True
    for name, (a, b) in (("text", (res[True][0], res[False][0])), ("pano", (res[True][1], res[False][1]))):
        assert len(a.layers) == len(b.layers)
        for i, (la, lb) in enumerate(zip(a.layers, b.layers)):
            for k in ("qkv", "Ppre", "P", "ctx", "a"):
                ulp_close(getattr(la.sa, k), getattr(lb.sa, k), f"{name} layer {i} sa.{k}")
            for k in ("z", "g", "out"):
                ulp_close(getattr(la.ffn, k), getattr(lb.ffn, k), f"{name} layer {i} ffn.{k}")
            for k, (x, y) in (("rstd_a", (la.sa.rstd_a, lb.sa.rstd_a)), ("rstd", (la.ffn.rstd, lb.ffn.rstd))):
                assert torch.allclose(x, y, rtol=2e-2, atol=1e-3), f"{name} layer {i} {k}"
            if p_drop > 0:          # identical dropout masks: the dropped probabilities are zero at exactly the same places
                za, zb = (la.sa.P.float() == 0), (lb.sa.P.float() == 0)
                assert (za != zb).float().mean().item() < 1e-4, f"{name} layer {i}: attention dropout masks differ"
        ulp_close(a.out, b.out, f"{name} encoder output")
    ulp_close(res[True][1].fused, res[False][1].fused, "pano fused embedding", ulps=3.0)
    ulp_close(res[True][1].img_attn, res[False][1].img_attn, "img_attns", ulps=3.0)


@pytest.mark.parametrize("row_split", [True, False])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("task", ["sap", "mlm"])
def test_fused_cross_encoders_save_what_the_unfused_kernels_save(task, p_drop, row_split, monkeypatch):
    """global || local co-attention encoders (sap) and the text-attends-to-map path (mlm) as one launch vs the per-op kernels: every
    tensor cross_layer_bwd reads, layer by layer; row_split: one workgroup per (sample, 16-row query tile) vs one per sample"""
    monkeypatch.setattr(O, "ENC_ROW_SPLIT", row_split)
    m = student(p_drop)
    m.train()
    batch = synth.make_batch(task, batch_size=6, seed=9, step=0, dup_view_prob=0.3)
    plan = build_plan(batch, task, torch.device(DEV))
    inp = m._inputs(batch, plan)
    m.store.sync_shadow()
    seed = torch.tensor([777, 31], dtype=torch.int32, device=DEV)
    n = m.net
    B, L, K, Vp = plan["B"], plan["L"], plan["K"], plan["Vp"]
    tl, gl_, vl = plan["lens"]["txt"], plan["lens"]["gmap"], [Vp] * B
    n.set_dropout(seed if p_drop > 0 else None, p_drop, p_drop)
    ct = n.text_fwd(plan)
    cp = n.pano_fwd(plan, inp.feats, inp.loc)
    gin = n.gmap_in_fwd(plan, cp, inp.gpos)
    vin = n.vp_in_fwd(plan, cp, inp.vpos)
    if task == "sap":
        specs = [("global", plan, gin.out, K, plan["gmap_mask"], gl_, plan["gmap_nodes"], ct.out, L, plan["txt_mask"], tl, plan["txt_tokens"], inp.dist),
                 ("local", plan, vin.out, Vp, plan["vp_mask"], vl, B * Vp, ct.out, L, plan["txt_mask"], tl, plan["txt_tokens"])]
    else:
        specs = [("global", plan, ct.out, L, plan["txt_mask"], tl, plan["txt_tokens"], gin.out, K, plan["gmap_mask"], gl_, plan["gmap_nodes"])]
    assert all(n.xenc_ok(sp[3], sp[8]) for sp in specs)
    fused = n.cross_fwd_fused(specs)
    torch.cuda.synchronize()
    if row_split:
        assert int(O.ENC_SYNC_LAST[0][0]) == 0, "a bounded hand-off wait gave up"
    plain = [n.cross_fwd(*sp[:12], dist=(sp[12] if len(sp) > 12 else None)) for sp in specs]
    torch.cuda.synchronize()
    for sp, a, b in zip(specs, fused, plain):
        for i, (la, lb) in enumerate(zip(a.layers, b.layers)):
            nm = f"{task} {sp[0]} layer {i}"
            for k in ("qkv", "Ppre", "P", "ctx", "a"):
                ulp_close(getattr(la.sa, k), getattr(lb.sa, k), f"{nm} sa.{k}")
            for k in ("q", "kv", "Ppre", "P", "cctx", "c"):
                ulp_close(getattr(la, k), getattr(lb, k), f"{nm} cross {k}")
            for k in ("z", "g", "out"):
                ulp_close(getattr(la.ffn, k), getattr(lb.ffn, k), f"{nm} ffn.{k}")
            for k, (x, y) in (("rstd_a", (la.sa.rstd_a, lb.sa.rstd_a)), ("rstd_c", (la.rstd_c, lb.rstd_c)), ("rstd", (la.ffn.rstd, lb.ffn.rstd))):
                assert torch.allclose(x, y, rtol=2e-2, atol=1e-3), f"{nm} {k}"
            if p_drop > 0:
                for pa, pb, what in ((la.sa.P, lb.sa.P, "self"), (la.P, lb.P, "cross")):
                    assert ((pa.float() == 0) != (pb.float() == 0)).float().mean().item() < 1e-4, f"{nm}: {what}-attention dropout masks differ"
        ulp_close(a.out, b.out, f"{task} {sp[0]} encoder output")


@pytest.mark.parametrize("task", ["sap", "mlm", "cfp"])
def test_training_step_with_fused_encoders_matches_unfused(task):
    from tests.test_fullsize_gpu import models, step
    batch = synth.make_batch(task, batch_size=16, seed=31, step=0)
    plan = build_plan(batch, task, torch.device(DEV))
    res = {}
    for fused in (False, True):
        O.FUSED_ENC = fused
        try:
            t, s = models(torch.bfloat16)
            out = step(t, s, batch, task, plan)
            res[fused] = (float(out["loss"].detach()), float(out["kdl_loss"].detach()), s.store.grad.clone())
        finally:
            O.FUSED_ENC = True
    (l1, k1, g1), (l0, k0, g0) = res[True], res[False]
    assert abs(l1 - l0) <= 2e-3 * abs(l0) and abs(k1 - k0) <= 3e-3 * abs(k0), (l1, l0, k1, k0)
    assert F.cosine_similarity(g1, g0, dim=0).item() > 0.9995


def test_unsupported_shapes_fall_back():
    m = student()
    assert not m.net.enc_ok(81, 6) and not m.net.enc_ok(80, 7) and m.net.enc_ok(80, 6) and m.net.enc_ok(37, 2)
    big = GlocalTextPathCMTPreTraining(make_config(256, role="teacher"), device=DEV, compute_dtype=torch.bfloat16)
    assert not big.net.enc_ok(36, 2)
    f32 = GlocalTextPathCMTPreTraining(make_config(128, role="student"), device=DEV, compute_dtype=torch.float32)
    assert not f32.net.enc_ok(36, 2)


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_rowblock_backward_matches_the_per_op_backward(p_drop):
    """text + panorama stacks' backward on magic_rowbwd (2 launches per block) vs the per-op chain (5 launches per block): same saved
    forward tensors, same upstream gradients, same dropout seed -> every parameter gradient of both encoders and the gradients wrt
    the embeddings agree to bf16 accumulation noise"""
    m = student(p_drop)
    m.train()
    batch = synth.make_batch("sap", batch_size=7, seed=5, step=0, dup_view_prob=0.3)
    plan = build_plan(batch, "sap", torch.device(DEV))
    inp = m._inputs(batch, plan)
    m.store.sync_shadow()
    n = m.net
    assert n.rbw_ok()
    seed = torch.tensor([4321, 99], dtype=torch.int32, device=DEV)
    g = torch.Generator().manual_seed(3)
    B, L, H, Np, V = plan["B"], plan["L"], n.H, plan["Np"], plan["V"]
    d_txt0 = (torch.randn(B * L, H, generator=g) * 0.1).to(DEV).bfloat16()
    d_pano0 = (torch.randn(Np * V, H, generator=g) * 0.1).to(DEV).bfloat16()
    d_fused0 = (torch.randn(Np, H, generator=g) * 0.1).to(DEV).bfloat16()
    res = {}
    for fused in (False, True):
        O.FUSED_RBW = fused
        try:
            n.set_dropout(seed if p_drop > 0 else None, p_drop, p_drop)
            ct = n.text_fwd(plan)
            cp = n.pano_fwd(plan, inp.feats, inp.loc)
            dPt = torch.randn(B, n.nh, L, ct.ldp, generator=g).to(DEV) * 0.01 if False else None
            m.store.zero_grad()
            O.defer_dw(True)
            if fused:
                assert n.rbw_ok()
                n.encoders_bwd(ct, cp, plan, d_txt0.clone(), None, d_pano0.clone(), d_fused0.clone(), None)
            else:
                n.text_bwd(ct, plan, d_txt0.clone(), None)
                n.pano_bwd(cp, plan, d_pano0.clone(), d_fused0.clone(), None)
            O.flush_dw()
            torch.cuda.synchronize()
            res[fused] = m.store.grad.clone()
        finally:
            O.FUSED_RBW = True
    a, b = res[True], res[False]
    assert torch.isfinite(a).all() and a.abs().max() > 0
    cos = F.cosine_similarity(a, b, dim=0).item()
    rel = ((a - b).norm() / b.norm()).item()
    assert cos > 0.9995 and rel < 3e-2, (cos, rel)
    # per tensor, for the tensors the two encoders own
    names = [nm for nm, _ in m.named_parameters() if ("lang_encoder" in nm or "pano_encoder" in nm or "embeddings" in nm)]
    worst = 0.0
    top = max(b[m.store.offsets[nm][0]:m.store.offsets[nm][0] + m.store.offsets[nm][1]].norm().item() for nm in names)
    for nm in names:
        off, cnt, _ = m.store.offsets[nm]
        ga, gb = a[off:off + cnt], b[off:off + cnt]
        if gb.norm() == 0:
            assert ga.norm() == 0, nm
            continue
        if gb.norm().item() < 1e-3 * top:          # analytically zero gradients (e.g. the key bias in front of a softmax): noise on both sides
            assert ga.norm().item() < 1e-2 * top, nm
            continue
        worst = max(worst, ((ga - gb).norm() / gb.norm()).item())
        assert ((ga - gb).norm() / gb.norm()).item() < 6e-2, (nm, ((ga - gb).norm() / gb.norm()).item())
    print(f"row-block backward vs per-op: cosine {cos:.6f}, rel L2 {rel:.2e}, worst tensor {worst:.2e}")


@pytest.mark.parametrize("with_seed", [False, True])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("max_len", [80, 41, 19])
def test_attention_backward_inside_the_rowblock_launches_matches_the_alternating_launches(max_len, p_drop, with_seed):
    """round 6 (csrc/encbwd.hip attn_tile_stage): a stack's backward as n + 1 launches -- the attention backward of block j+1 done per 16-row tile of
    one sample in front of block j's per-token chain, rowsum(P dP) taken as dO . O (+ the distillation seed's term) -- against the round 2-5 structure
    (magic_rowbwd and magic_attn_bwd alternating) and against the per-op chain: same saved tensors, same upstream gradients, same dropout seed, ragged
    instruction lengths (5 / 3 / 2 tiles per sample, partial last tiles), 36- and 37-view panoramas, with and without a gradient seeded into the top
    blocks' attention maps (attention distillation)."""
    m = student(p_drop)
    m.train()
    batch = synth.make_batch("sap", batch_size=7, seed=5, step=0, dup_view_prob=0.3, max_len=max_len, min_len=min(20, max_len - 4))
    plan = build_plan(batch, "sap", torch.device(DEV))
    inp = m._inputs(batch, plan)
    m.store.sync_shadow()
    n = m.net
    assert n.rbw_ok() and O.rowbwd_attn_ok(n.dtype, n.H, n.I, n.nh, plan["L"]) and O.rowbwd_attn_ok(n.dtype, n.H, n.I, n.nh, plan["V"])      # (module default: mode 1)
    seed = torch.tensor([4321, 99], dtype=torch.int32, device=DEV)
    g = torch.Generator().manual_seed(3)
    B, L, H, Np, V = plan["B"], plan["L"], n.H, plan["Np"], plan["V"]
    d_txt0 = (torch.randn(B * L, H, generator=g) * 0.1).to(DEV).bfloat16()
    d_pano0 = (torch.randn(Np * V, H, generator=g) * 0.1).to(DEV).bfloat16()
    d_fused0 = (torch.randn(Np, H, generator=g) * 0.1).to(DEV).bfloat16()
    res = {}
    for mode in ("per_op", "alternating", "inside", "hybrid"):      # hybrid = the default: inside once only the text stack is left (MAGIC_RBW_ATTN=1)
        O.FUSED_RBW, O.RBW_ATTN_MODE = mode != "per_op", {"inside": 2, "hybrid": 1}.get(mode, 0)
        try:
            n.set_dropout(seed if p_drop > 0 else None, p_drop, p_drop)
            ct = n.text_fwd(plan)
            cp = n.pano_fwd(plan, inp.feats, inp.loc)
            gs = torch.Generator().manual_seed(11)
            dPt = dPp = None
            if with_seed:      # fp32 [B, heads, N, ldp] with zero pad columns, as the distillation loss kernels leave it
                dPt = torch.zeros(B, n.nh, L, ct.ldp)
                dPt[..., :L] = torch.randn(B, n.nh, L, L, generator=gs) * 0.02
                dPp = torch.zeros(Np, n.nh, V, cp.ldp)
                dPp[..., :V] = torch.randn(Np, n.nh, V, V, generator=gs) * 0.02
                dPt, dPp = dPt.to(DEV), dPp.to(DEV)
            m.store.zero_grad()
            O.defer_dw(True)
            if mode == "per_op":
                n.text_bwd(ct, plan, d_txt0.clone(), dPt)
                n.pano_bwd(cp, plan, d_pano0.clone(), d_fused0.clone(), dPp)
            else:
                n.encoders_bwd(ct, cp, plan, d_txt0.clone(), dPt, d_pano0.clone(), d_fused0.clone(), dPp)
            O.flush_dw()
            torch.cuda.synchronize()
            res[mode] = m.store.grad.clone()
        finally:
            O.FUSED_RBW, O.RBW_ATTN_MODE = True, 1
    names = [nm for nm, _ in m.named_parameters() if ("lang_encoder" in nm or "pano_encoder" in nm or "embeddings" in nm)]
    for got, ref, tol_all, tol_one in (("inside", "alternating", 1.5e-2, 4e-2), ("inside", "per_op", 3e-2, 6e-2), ("hybrid", "alternating", 1.5e-2, 4e-2)):
        a, b = res[got], res[ref]
        assert torch.isfinite(a).all() and a.abs().max() > 0
        cos = F.cosine_similarity(a, b, dim=0).item()
        rel = ((a - b).norm() / b.norm()).item()
        assert cos > 0.9995 and rel < tol_all, (got, ref, cos, rel)
        top = max(b[m.store.offsets[nm][0]:m.store.offsets[nm][0] + m.store.offsets[nm][1]].norm().item() for nm in names)
        worst = (0.0, "")
        for nm in names:
            off, cnt, _ = m.store.offsets[nm]
            ga, gb = a[off:off + cnt], b[off:off + cnt]
            if gb.norm() == 0:
                assert ga.norm() == 0, nm
                continue
            if gb.norm().item() < 1e-3 * top:
                assert ga.norm().item() < 1e-2 * top, nm
                continue
            r = ((ga - gb).norm() / gb.norm()).item()
            worst = max(worst, (r, nm))
            assert r < tol_one, (got, ref, nm, r)
        print(f"attention backward inside the row-block launches ({got}) vs {ref}: cosine {cos:.6f}, rel L2 {rel:.2e}, worst tensor {worst[0]:.2e} ({worst[1]})")


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("task", ["sap", "mlm", "mrc"])
def test_cross_encoder_rowblock_backward_matches_the_per_op_backward(task, p_drop):
    """whole backward with the cross-modal encoders (and the text / panorama stacks) on magic_rowbwd -- full chain + short chain per
    block -- vs the per-op backward: same forward, same dropout seed -> the same gradient for every parameter"""
    m = student(p_drop, **({"pretrain_tasks": ["mlm", "sap", "mrc"]} if task == "mrc" else {}))
    m.train()
    batch = synth.make_batch(task, batch_size=6, seed=11, step=0, dup_view_prob=0.3)
    res = {}
    for fused in (False, True):
        O.FUSED_RBW = fused
        try:
            m.store.zero_grad()
            torch.manual_seed(0)                   # the dropout seed of the step is drawn from the device generator
            out = m(batch, task, compute_loss=True)
            assert m.net.rbw_ok() == fused
            m.backward()
            torch.cuda.synchronize()
            res[fused] = (float(out["loss"].detach()), m.store.grad.clone())
        finally:
            O.FUSED_RBW = True
    (l1, a), (l0, b) = res[True], res[False]
    assert abs(l1 - l0) <= 1e-6 * max(abs(l0), 1.0), (l1, l0)      # the forward is the same code on both sides
    assert torch.isfinite(a).all() and a.abs().max() > 0
    cos = F.cosine_similarity(a, b, dim=0).item()
    rel = ((a - b).norm() / b.norm()).item()
    names = [nm for nm, _ in m.named_parameters() if ("global_encoder" in nm or "local_encoder" in nm)]
    top = max(b[m.store.offsets[nm][0]:m.store.offsets[nm][0] + m.store.offsets[nm][1]].norm().item() for nm in names)
    worst = 0.0
    for nm in names:
        off, cnt, _ = m.store.offsets[nm]
        ga, gb = a[off:off + cnt], b[off:off + cnt]
        if gb.norm().item() < 1e-3 * top:
            assert ga.norm().item() < 1e-2 * top, nm
            continue
        worst = max(worst, ((ga - gb).norm() / gb.norm()).item())
        assert ((ga - gb).norm() / gb.norm()).item() < 6e-2, (nm, ((ga - gb).norm() / gb.norm()).item())
    print(f"{task} p={p_drop}: cross row-block backward vs per-op: cosine {cos:.6f}, rel L2 {rel:.2e}, worst cross tensor {worst:.2e}")
    assert cos > 0.9995 and rel < 3e-2, (cos, rel)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_node_inputs_in_one_launch_are_bit_identical_to_the_per_op_chain(dtype):
    """magic_node_in_fwd (gathers + position embedding + step embedding of the map / viewpoint tokens, both encoders in one launch) keeps the
    per-op chain's rounding points: csr_gather (+ accumulate) -> smallk_ln_fwd -> ln_fwd(do_ln = False)"""
    cfg = make_config(128, role="student", teacher_hidden_size=256, kdl=KDL)
    m = GlocalTextPathCMTPreTraining(cfg, device=DEV, compute_dtype=dtype, seed=3)
    batch = synth.make_batch("sap", batch_size=7, seed=5, step=0, dup_view_prob=0.3)
    plan = build_plan(batch, "sap", torch.device(DEV))
    inp = m._inputs(batch, plan)
    m.store.sync_shadow()
    n = m.net
    n.set_dropout(None, 0.0, 0.0)
    ct = n.text_fwd(plan)
    cp = n.pano_fwd(plan, inp.feats, inp.loc)
    g0, v0 = n.gmap_in_fwd(plan, cp, inp.gpos), n.vp_in_fwd(plan, cp, inp.vpos)
    g1, v1 = n.nodes_in_fwd(plan, cp, inp.gpos, inp.vpos)
    go, _ = n.nodes_in_fwd(plan, cp, inp.gpos, None)                   # one encoder only (mlm / mrc)
    _, vo = n.nodes_in_fwd(plan, cp, None, inp.vpos)
    gimg = torch.randn(plan["B"] * plan["K"], n.H, device=DEV).to(dtype)
    ga, gb = n.gmap_in_fwd(plan, None, inp.gpos, gimg=gimg), n.nodes_in_fwd(plan, None, inp.gpos, None, gimg=gimg)[0]   # navigator form
    torch.cuda.synchronize()
    for a, b, nm in ((g0, g1, "gmap"), (v0, v1, "vp"), (g0, go, "gmap alone"), (v0, vo, "vp alone"), (ga, gb, "gmap from given embeddings")):
        assert torch.equal(a.out, b.out), nm
        assert torch.equal(a.A, b.A) and torch.equal(a.rstd, b.rstd), nm
    assert g1.out.abs().max() > 0 and v1.out.abs().max() > 0


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("dtype,H", [(torch.bfloat16, 128), (torch.float16, 128), (torch.float32, 128), (torch.bfloat16, 256), (torch.bfloat16, 768)])
def test_input_embeddings_in_one_launch_are_bit_identical_to_the_per_op_chain(dtype, H, p_drop):
    """magic_embed_in_fwd (round 4): the panorama stage's image LayerNorm, location linear + LayerNorm, sum LayerNorm + dropout AND the text embedding's
    gathers + LayerNorm + dropout in ONE launch behind the image projection -- every tensor the backward reads is bit-identical to
    magic_ln_fwd -> magic_smallk_ln_fwd -> magic_ln_fwd (+ the text magic_ln_fwd), with the same dropout masks"""
    cfg = make_config(H, role="student" if H == 128 else "teacher", hidden_dropout_prob=p_drop, attention_probs_dropout_prob=p_drop,
                      **(dict(teacher_hidden_size=256, kdl=KDL) if H == 128 else {}))
    m = GlocalTextPathCMTPreTraining(cfg, device=DEV, compute_dtype=dtype, seed=3)
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        for nme, p in m.named_parameters():
            if ("LayerNorm" in nme or "layer_norm" in nme) and p.dim() == 1:
                p.add_((torch.randn(p.shape, generator=g) * 0.1).to(DEV))
    m.store.shadow_clean = False
    batch = synth.make_batch("sap", batch_size=7, seed=5, step=0, dup_view_prob=0.3)
    plan = build_plan(batch, "sap", torch.device(DEV))
    inp = m._inputs(batch, plan)
    m.store.sync_shadow()
    n = m.net
    seed = torch.tensor([12345, 678], dtype=torch.int32, device=DEV)
    n.set_dropout(seed if p_drop > 0 else None, p_drop, p_drop)
    assert n.embed_in_ok()
    ct0, cp0 = n.text_fwd(plan, defer=True), n.pano_fwd(plan, inp.feats, inp.loc, defer=True)      # per-op: four launches behind the projection
    if not n.enc_ok(plan["L"], cfg.num_l_layers):          # widths without the whole-encoder launch ran their layers too: only the embeddings matter here
        pass
    ct1, cp1 = n.embeds_fwd(plan, inp.feats, inp.loc)
    torch.cuda.synchronize()
    for nm in ("E", "rstd_e") + (("Ed",) if p_drop > 0 else ()):
        assert torch.equal(getattr(ct0, nm), getattr(ct1, nm)), f"text {nm}"
    for nm in ("A1", "rstd_a1", "A2", "rstd_a2", "X0", "rstd_x0") + (("X0d",) if p_drop > 0 else ()):
        assert torch.equal(getattr(cp0, nm), getattr(cp1, nm)), f"panorama {nm}"
    assert ct1.E.float().abs().max() > 0 and cp1.X0.float().abs().max() > 0 and (p_drop == 0 or (cp1.X0d == 0).float().mean() > 0.05)


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("task", ["sap", "mlm"])
def test_input_embeddings_backward_in_one_launch_matches_the_per_op_sequence(task, p_drop, monkeypatch):
    """magic_embed_in_bwd (round 4): the panorama stage's three LayerNorm backwards (+ nav-type / token-type / loc_linear gradients) and the text
    embedding's backward in ONE launch, inside a whole training step: same loss (the forward is bit-identical), and every parameter gradient
    agrees with the per-op sequence to fp32 summation order -- the stage's own parameters, the image projection (its dY operand dP0 is
    bit-identical) and the embedding tables are named explicitly"""
    from tests.test_model_gpu import KDL as _K  # noqa: F401
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(O, "EMBED_IN", fused)
        m = student(p_drop)
        m.train()
        batch = synth.make_batch(task, batch_size=7, seed=5, step=0, dup_view_prob=0.3)
        plan = build_plan(batch, task, torch.device(DEV))
        m.dropout_seed = torch.tensor([4242, 99], dtype=torch.int32, device=DEV)
        m.store.ensure_grads()
        m.store.zero_grad()
        out = m(synth.batch_to(batch, DEV), task, compute_loss=True, plan=plan)
        m.backward()
        torch.cuda.synchronize()
        res[fused] = (float(out["loss"]), m.store.grad.clone(), m)
    assert res[True][0] == res[False][0]
    a, b, m = res[True][1], res[False][1], res[True][2]
    top = b.abs().max().item()
    names = ["bert.img_embeddings.img_layer_norm.weight", "bert.img_embeddings.img_layer_norm.bias", "bert.img_embeddings.loc_layer_norm.weight",
             "bert.img_embeddings.loc_layer_norm.bias", "bert.img_embeddings.layer_norm.weight", "bert.img_embeddings.layer_norm.bias",
             "bert.img_embeddings.loc_linear.weight", "bert.img_embeddings.loc_linear.bias", "bert.img_embeddings.nav_type_embedding.weight",
             "bert.img_embeddings.img_linear.weight", "bert.img_embeddings.img_linear.bias", "bert.embeddings.token_type_embeddings.weight",
             "bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight", "bert.embeddings.LayerNorm.weight"]
    for nm in names:
        off, cnt, _ = m.store.offsets[nm]
        ga, gb = a[off:off + cnt], b[off:off + cnt]
        assert gb.abs().max().item() > 0, nm
        # (fp32 atomics in both forms: the sums agree to summation order -- 3e-4 of an element was seen once in round 5 on the location projection)
        assert torch.allclose(ga, gb, rtol=1e-3, atol=2e-6 * top + 1e-4 * gb.abs().max().item()), (nm, (ga - gb).abs().max().item(), gb.abs().max().item())
    assert torch.allclose(a, b, rtol=1e-3, atol=1e-5 * top), (a - b).abs().max().item()


def test_node_inputs_backward_in_shared_launches_matches_the_per_op_sequence():
    """nodes_in_bwd (step-table gradient, both position-embedding backwards as a pair, the three transposed gathers as one launch) against
    vp_in_bwd + gmap_in_bwd: the gathered gradients are bit-identical (same rounding order), the parameter gradients agree to fp32
    atomic-order noise"""
    m = student(0.0)
    batch = synth.make_batch("sap", batch_size=7, seed=5, step=0, dup_view_prob=0.3)
    plan = build_plan(batch, "sap", torch.device(DEV))
    inp = m._inputs(batch, plan)
    m.store.sync_shadow()
    m.store.ensure_grads()
    n = m.net
    n.set_dropout(None, 0.0, 0.0)
    n.text_fwd(plan)
    cp = n.pano_fwd(plan, inp.feats, inp.loc)
    gin, vin = n.nodes_in_fwd(plan, cp, inp.gpos, inp.vpos)
    g = torch.Generator().manual_seed(7)
    H, B, K, Vp, Np, V = n.H, plan["B"], plan["K"], plan["Vp"], plan["Np"], plan["V"]
    d_gin = (torch.randn(B * K, H, generator=g) * 0.1).to(DEV).bfloat16()
    d_vin = (torch.randn(B * Vp, H, generator=g) * 0.1).to(DEV).bfloat16()
    base_p = (torch.randn(Np * V, H, generator=g) * 0.1).to(DEV).bfloat16()
    base_f = (torch.randn(Np, H, generator=g) * 0.1).to(DEV).bfloat16()
    res = {}
    for shared in (False, True):
        m.store.zero_grad()
        dp, df = base_p.clone(), base_f.clone()
        if shared:
            n.nodes_in_bwd(plan, gin, d_gin, vin, d_vin, dp, df)
        else:
            n.vp_in_bwd(vin, plan, d_vin, dp)
            n.gmap_in_bwd(gin, plan, d_gin, dp, df)
        torch.cuda.synchronize()
        res[shared] = (dp, df, m.store.grad.clone())
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert not torch.equal(res[True][0], base_p)
    ga, gb = res[True][2], res[False][2]
    assert gb.abs().max() > 0 and torch.allclose(ga, gb, rtol=1e-4, atol=1e-5), (ga - gb).abs().max().item()


def test_encoder_health_counts_no_given_up_hand_off_after_the_launches_of_this_file():
    """the sticky process-wide counter behind trainer.check_health(): every row-split launch so far handed all its rows over"""
    gave_up, launches = O.encoder_health(DEV)
    assert gave_up == 0 and launches > 0
    O.check_encoder_health(DEV)


def _streams_overlap(side):
    """do the main stream and `side` run kernels side by side IN THIS PROCESS?  (HIP maps streams onto a few hardware queues; two streams on
    one queue serialise.)  Two 3 ms gate waits without an encoder launch: ~3 ms together when they overlap, ~6 ms when they do not."""
    sa, sb = O.gate_stats_new(DEV), O.gate_stats_new(DEV)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream()
    e0.record(main)
    side.wait_event(e0)
    with torch.cuda.stream(side):
        O.encoder_start_gate(sa, 3000, 0)
    O.encoder_start_gate(sb, 3000, 0)
    main.wait_stream(side)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) < 4.5


def test_encoder_start_gate_counts_what_it_does_and_switches_itself_off():
    """magic_encoder_start_gate: a stream parked on it resumes when another stream's whole-encoder launch has its last workgroup on a CU (the
    teacher's forward is held back that way).  Nothing in HIP promises that two streams overlap, so the properties asserted are the ones
    the product relies on: every outcome is COUNTED, a launch that is already resident opens the gate at once, and after three consecutive
    timeouts the gate is an empty launch -- a process whose streams serialise pays 3 timeouts, not one per step."""
    m = student()
    m.eval()
    b = synth.make_batch("sap", batch_size=8, seed=3, step=0)
    bd, plan = synth.batch_to(b, DEV), build_plan(b, "sap", DEV)
    assert m.will_fuse_encoders(plan)
    with torch.no_grad():
        m(bd, "sap", compute_loss=False, plan=plan)              # warm-up (code objects, allocator)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    st = O.gate_stats_new(DEV)

    # (1) no encoder launch follows: the gate gives up after its timeout and says so
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record()
        O.encoder_start_gate(st, 2000, 0)
        e1.record()
    torch.cuda.synchronize()
    r = O.gate_report(st)
    assert (r["calls"], r["timeouts"], r["consecutive_timeouts"], r["opened"], r["disabled"]) == (1, 1, 1, 0, 0), r
    assert e0.elapsed_time(e1) > 1.5, e0.elapsed_time(e1)

    # (2) a launch that became resident a moment ago IS the launch the gate was meant to follow: it opens at once (and the run of timeouts ends)
    with torch.no_grad():
        m(bd, "sap", compute_loss=False, plan=plan)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        O.encoder_start_gate(st, 50000, 100000)
    torch.cuda.synchronize()
    r = O.gate_report(st)
    assert (r["calls"], r["already_resident"], r["timeouts"], r["consecutive_timeouts"]) == (2, 1, 1, 0), r

    # (3) an encoder launch on the main stream opens a parked gate -- where this process's streams overlap at all; where they do not, the
    # gate must time out and COUNT it (that is what switches it off in the product)
    overlap = _streams_overlap(side)
    seen = []
    for _ in range(3):                                           # (a host-side stall between the two enqueues may cost one attempt, not the property)
        st = O.gate_stats_new(DEV)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            O.encoder_start_gate(st, 30000, 0)
        with torch.no_grad():
            m(bd, "sap", compute_loss=False, plan=plan)
        torch.cuda.synchronize()
        r = O.gate_report(st)
        assert r["calls"] == 1 and r["opened"] + r["timeouts"] == 1, r
        seen.append(r["opened"])
        if r["opened"] or not overlap:
            break
    if overlap:
        assert seen[-1] == 1, ("the streams overlap, yet the gate never saw the encoder launch", seen)
    print("streams overlap:", overlap, "opened:", seen)

    # (4) three consecutive timeouts switch the gate off: later calls return at once and are counted as skipped
    st = O.gate_stats_new(DEV)
    with torch.cuda.stream(side):
        for _ in range(3):
            O.encoder_start_gate(st, 500, 0)
        e0.record()
        for _ in range(5):
            O.encoder_start_gate(st, 100000, 0)              # 5 x 100 ms if they waited
        e1.record()
    torch.cuda.synchronize()
    r = O.gate_report(st)
    assert (r["calls"], r["timeouts"], r["disabled"], r["skipped"]) == (8, 3, 1, 5), r
    assert e0.elapsed_time(e1) < 50.0, e0.elapsed_time(e1)
    # (4b) ... and the gate re-arms itself after a backoff of 32 skipped calls (a transient loss of overlap must not leave it off for the rest of
    # the run); switched off again, it waits twice as long
    with torch.cuda.stream(side):
        for _ in range(27):
            O.encoder_start_gate(st, 500, 0)                 # skipped calls 6 .. 32: the last one re-arms
    torch.cuda.synchronize()
    r = O.gate_report(st)
    assert (r["skipped"], r["disabled"], r["timeouts"]) == (32, 0, 3), r
    with torch.cuda.stream(side):
        for _ in range(3):
            O.encoder_start_gate(st, 500, 0)                 # no encoder launch: three more timeouts switch it off again
        for _ in range(63):
            O.encoder_start_gate(st, 500, 0)                 # backoff 64 now: 63 skipped calls do not re-arm it ...
    torch.cuda.synchronize()
    r = O.gate_report(st)
    assert (r["timeouts"], r["disabled"], r["skipped"]) == (6, 1, 32 + 63), r
    with torch.cuda.stream(side):
        O.encoder_start_gate(st, 500, 0)                     # ... the 64th does
    torch.cuda.synchronize()
    assert O.gate_report(st)["disabled"] == 0
    st.zero_()                                                   # the owner re-arms it
    with torch.cuda.stream(side):
        O.encoder_start_gate(st, 500, 0)
    torch.cuda.synchronize()
    assert O.gate_report(st)["timeouts"] == 1
