"""csrc/chain.hip (-m gpu): the forward-only row chain of a post-LN block at the teacher's width (H = 256, FFN 1024) against a plain PyTorch fp32
reference of the same ops with the same 16-bit rounding points (y1, GELU output, y2, projection), and against the per-op kernels it replaces
(magic_linear_ln + magic_gemm + magic_ln_fwd)."""
import pytest
import torch
import torch.nn.functional as F

import magic_amd  # noqa: F401
from magic_amd.host import lib as L
from magic_amd.host import ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
H, I = 256, 1024


def _weights(g, dtype, n_proj):
    r = lambda *s, sc=0.05: (torch.randn(*s, generator=g) * sc)
    w = dict(Wa=r(H, H), ba=r(H), g1=1 + r(H), b1=r(H), W1=r(I, H), bi=r(I), W2=r(H, I, sc=0.03), bo2=r(H), g2=1 + r(H), b2=r(H),
             Wp=r(n_proj, H), bp=r(n_proj))
    return {k: (v.to(DEV).to(dtype) if k[0] == "W" else v.to(DEV)) for k, v in w.items()}


def _reference(x, res, w, dtype, ffn, proj):
    rnd = lambda t: t.to(dtype).float()
    f = lambda k: w[k].float()
    y1 = rnd(F.layer_norm(x.float() @ f("Wa").T + f("ba") + res.float(), (H,), f("g1"), f("b1"), 1e-12))
    out = dict(y1=y1)
    last = y1
    if ffn:
        gl = rnd(F.gelu(y1 @ f("W1").T + f("bi")))
        last = out["y2"] = rnd(F.layer_norm(gl @ f("W2").T + f("bo2") + y1, (H,), f("g2"), f("b2"), 1e-12))
    if proj:
        out["proj"] = rnd(last @ f("Wp").T + f("bp"))
    return out


def _run(x, res, M, w, dtype, ffn, proj, store_y1):
    y1 = torch.full((M, H), 7.0, dtype=dtype, device=DEV) if store_y1 else None
    y2 = torch.full((M, H), 7.0, dtype=dtype, device=DEV) if ffn else None
    po = torch.full((M, w["Wp"].shape[0]), 7.0, dtype=dtype, device=DEV) if proj else None
    def pk(k):                         # the chain reads its weights in MFMA-fragment order; the packed copies live as long as `w`
        if "pk_" + k not in w:             # (a grouped launch happens at the end of the `with L.group()` block)
            w["pk_" + k] = O.pack_frag(w[k])
        return w["pk_" + k]
    O.chain_fwd(x, res, M, pk("Wa"), w["ba"], w["g1"], w["b1"], 1e-12, y1=y1,
                ffn=(pk("W1"), w["bi"], pk("W2"), w["bo2"], w["g2"], w["b2"], I) if ffn else None, y2=y2,
                proj=(pk("Wp"), w["bp"], w["Wp"].shape[0]) if proj else None, proj_out=po)
    return dict(y1=y1, y2=y2, proj=po)


TOL = {torch.bfloat16: dict(rtol=2e-2, atol=3e-2), torch.float16: dict(rtol=3e-3, atol=4e-3)}


@pytest.fixture(params=[64, 32], ids=["rows64", "rows32"])
def tile_rows(request):
    """both forms of the kernel, forced: 64-row tiles on the 32 x 32 x 16 product and 32-row tiles on 16 x 16 x 32 (the default setting, 1,
    picks by launch size)"""
    before = O.chain_tile_rows()
    assert O.chain_tile_rows(request.param) == request.param
    yield request.param
    O.chain_tile_rows(before)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 16, 37, 100, 3840])
@pytest.mark.parametrize("ffn,n_proj,store_y1", [(True, 768, False), (True, 0, True), (False, 256, True), (False, 512, False), (True, 256, False)])
def test_chain_matches_fp32_reference(dtype, M, ffn, n_proj, store_y1, tile_rows):
    assert O.chain_ok(dtype, H, I)
    g = torch.Generator().manual_seed(M + n_proj)
    w = _weights(g, dtype, max(n_proj, 256))
    x = torch.randn(M, H, generator=g).to(DEV).to(dtype)
    res = torch.randn(M, H, generator=g).to(DEV).to(dtype)
    got = _run(x, res, M, w, dtype, ffn, n_proj > 0, store_y1)
    want = _reference(x, res, w, dtype, ffn, n_proj > 0)
    torch.cuda.synchronize()
    for k, v in got.items():
        if v is None:
            continue
        assert torch.isfinite(v.float()).all() and not (v.float() == 7.0).all(), k
        err = (v.float() - want[k]).abs().max().item()
        assert torch.allclose(v.float(), want[k], **TOL[dtype]), f"{k}: max|err| {err:.3e} (ref max {want[k].abs().max().item():.2f})"


def test_chain_equals_the_per_op_kernels_it_replaces(tile_rows):
    dtype, M = torch.bfloat16, 3840
    g = torch.Generator().manual_seed(5)
    w = _weights(g, dtype, 768)
    x = torch.randn(M, H, generator=g).to(DEV).to(dtype)
    res = torch.randn(M, H, generator=g).to(DEV).to(dtype)
    got = _run(x, res, M, w, dtype, True, True, True)
    a, rstd = torch.empty(M, H, dtype=dtype, device=DEV), torch.empty(M, dtype=torch.float32, device=DEV)
    O.linear_ln(x, w["Wa"], w["ba"], M, res, w["g1"], w["b1"], 1e-12, a, rstd)
    gl = O.linear_fwd(a, w["W1"], w["bi"], M, epilogue=1)
    d = O.linear_fwd(gl, w["W2"], w["bo2"], M, residual=a)
    out = torch.empty(M, H, dtype=dtype, device=DEV)
    O.ln_fwd(M, H, out, in0=d, gamma=w["g2"], beta=w["b2"], eps=1e-12, rstd=rstd)
    qkv = O.linear_fwd(out, w["Wp"], w["bp"], M)
    torch.cuda.synchronize()
    # y1: the same arithmetic up to summation order -> a few bf16 ulps; later stages inherit y1's rounding differences
    for name, u, v in (("y1", got["y1"], a), ("y2", got["y2"], out), ("proj", got["proj"], qkv)):
        diff = (u.float() - v.float()).abs()
        assert diff.max().item() < 0.07 and diff.mean().item() < 2e-3, (name, diff.max().item(), diff.mean().item())


def test_two_chains_share_one_launch_and_strided_input(tile_rows):
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(9)
    wa, wb = _weights(g, dtype, 768), _weights(g, dtype, 256)
    Ma, Mb = 333, 50
    xa_full = torch.randn(Ma, 3 * H, generator=g).to(DEV).to(dtype)         # a [M, 3H] buffer: the chain reads a column slice of it (pitch 3H)
    xa = xa_full[:, H:2 * H]
    ra = torch.randn(Ma, H, generator=g).to(DEV).to(dtype)
    xb = torch.randn(Mb, H, generator=g).to(DEV).to(dtype)
    rb = torch.randn(Mb, H, generator=g).to(DEV).to(dtype)
    with L.group():
        ga = _run(xa, ra, Ma, wa, dtype, True, True, False)
        gb = _run(xb, rb, Mb, wb, dtype, False, True, True)
    torch.cuda.synchronize()
    wa_ref, wb_ref = _reference(xa, ra, wa, dtype, True, True), _reference(xb, rb, wb, dtype, False, True)
    for got, want in ((ga, wa_ref), (gb, wb_ref)):
        for k, v in got.items():
            if v is not None:
                assert torch.allclose(v.float(), want[k], **TOL[dtype]), k


def test_the_two_tile_forms_agree_to_16_bit_rounding():
    dtype, M = torch.bfloat16, 1000
    g = torch.Generator().manual_seed(21)
    w = _weights(g, dtype, 768)
    x = torch.randn(M, H, generator=g).to(DEV).to(dtype)
    res = torch.randn(M, H, generator=g).to(DEV).to(dtype)
    before = O.chain_tile_rows()
    try:
        outs = {}
        for rows in (32, 64):
            O.chain_tile_rows(rows)
            outs[rows] = _run(x, res, M, w, dtype, True, True, True)
            torch.cuda.synchronize()
    finally:
        O.chain_tile_rows(before)
    for k in ("y1", "y2", "proj"):
        d = (outs[32][k].float() - outs[64][k].float()).abs()
        assert d.max().item() < 0.07 and d.mean().item() < 1e-3, (k, d.max().item(), d.mean().item())
    with pytest.raises(L.MagicHipError):
        O.chain_tile_rows(48)
    # the default setting picks by launch size: more 32-row tiles than CUs -> the 64-row form (same outputs as forcing it), else the 32-row form
    assert before == 1 or "MAGIC_CHAIN_ROWS" in __import__("os").environ
    O.chain_tile_rows(1)
    try:
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        Mbig = 32 * (cus + 3)
        xb = torch.randn(Mbig, H, generator=g).to(DEV).to(dtype)
        rb = torch.randn(Mbig, H, generator=g).to(DEV).to(dtype)
        auto_big, auto_small = _run(xb, rb, Mbig, w, dtype, True, True, True), _run(x, res, M, w, dtype, True, True, True)
        O.chain_tile_rows(64)
        f64 = _run(xb, rb, Mbig, w, dtype, True, True, True)
        torch.cuda.synchronize()
        for k in ("y1", "y2", "proj"):
            assert torch.equal(auto_big[k], f64[k]), ("a launch over one round takes the 64-row form", k)
            assert torch.equal(auto_small[k], outs[32][k]), ("a launch within one round takes the 32-row form", k)
    finally:
        O.chain_tile_rows(before)


def test_pack_frag_layout():
    """chunk ((nt (K/32) + ks) 64 + l) of 8 elements = W[16 nt + (l & 15), 32 ks + 8 (l >> 4) .. +7] (include/magic_hip.h)"""
    for dtype in (torch.bfloat16, torch.float16):
        W = torch.randn(64, 96).to(DEV).to(dtype)
        want = W.view(4, 16, 3, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)
        assert torch.equal(O.pack_frag(W), want)


def test_layout_spans_writes_w_and_wt_in_fragment_order_in_one_launch():
    """magic_layout_spans (the launch that follows AdamW): flag 1 -> W, flag 2 -> W^T, both in fragment order, several spans of one flat buffer"""
    import ctypes as C
    frag = lambda W: W.reshape(W.shape[0] // 16, 16, W.shape[1] // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)
    for dtype in (torch.bfloat16, torch.float16):
        shapes, flags = [(128, 128), (384, 128), (128, 512), (64, 64)], [3, 1, 2, 3]
        offs, total = [], 0
        for r, c in shapes:
            offs.append(total)
            total += r * c + 64            # (gaps between the spans stay untouched)
        src = torch.randn(total).to(DEV).to(dtype)
        df, dtf = torch.full((total,), 9.0, dtype=dtype, device=DEV), torch.full((total,), 9.0, dtype=dtype, device=DEV)
        n = len(shapes)
        a_off, a_r, a_c, a_f = (C.c_longlong * n)(*offs), (C.c_int * n)(*[s[0] for s in shapes]), (C.c_int * n)(*[s[1] for s in shapes]), (C.c_int * n)(*flags)
        L.call("magic_layout_spans", L.P(src), L.P(df), L.P(dtf), n, C.addressof(a_off), C.addressof(a_r), C.addressof(a_c), C.addressof(a_f), L.stream())
        torch.cuda.synchronize()
        for (r, c), o, fl in zip(shapes, offs, flags):
            W = src[o:o + r * c].view(r, c)
            if fl & 1:
                assert torch.equal(df[o:o + r * c], frag(W))
            else:
                assert (df[o:o + r * c] == 9.0).all()
            if fl & 2:
                assert torch.equal(dtf[o:o + r * c], frag(W.t().contiguous()))
            else:
                assert (dtf[o:o + r * c] == 9.0).all()
            assert (df[o + r * c:o + r * c + 64] == 9.0).all() and (dtf[o + r * c:o + r * c + 64] == 9.0).all()


def test_chain_rejects_bad_arguments():
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(1)
    w = _weights(g, dtype, 384)
    x = torch.randn(16, H, generator=g).to(DEV).to(dtype)
    with pytest.raises(L.MagicHipError):          # a projection width that is not H, 2H or 3H
        O.chain_fwd(x, x, 16, O.pack_frag(w["Wa"]), w["ba"], w["g1"], w["b1"], 1e-12, proj=(O.pack_frag(w["Wp"]), w["bp"], 384),
                    proj_out=torch.empty(16, 384, dtype=dtype, device=DEV))
    with pytest.raises(L.MagicHipError):          # nothing to write
        O.chain_fwd(x, x, 16, O.pack_frag(w["Wa"]), w["ba"], w["g1"], w["b1"], 1e-12)
    assert not O.chain_ok(torch.float32, H, I) and not O.chain_ok(dtype, 128, 512)


def test_fragment_order_copies_follow_the_weights_through_training_steps_and_foreign_optimizers():
    """ParamStore.f_span / tf_span (round 3): the MFMA-fragment-order copies of W and W^T the whole-encoder / row-block kernels read must equal
    the row-major 16-bit shadow after (a) the fused AdamW step (which refreshes them in its own launch sequence), (b) a torch optimizer step on
    the store's parameters (global post-step hook marks the shadows stale; the next forward re-casts and re-lays them out), (c) load_state_dict."""
    from magic_amd.host import synth
    from magic_amd.host.plan import build_plan
    from magic_amd.host.trainer import PretrainStep
    from tests.test_model_gpu import RW, build
    frag = lambda W: W.reshape(W.shape[0] // 16, 16, W.shape[1] // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)
    _, _, g_t, g_s = build(torch.bfloat16)
    st = g_s.store
    assert st.f_spans, "a trainable H = 128 student registers its encoder matrices at construction"

    def consistent():
        torch.cuda.synchronize()
        n = 0
        for off, (rows, cols, flags) in st.f_spans.items():
            W = st.shadow[off:off + rows * cols].view(rows, cols)
            if flags & 1:
                assert torch.equal(st.shadow_f[off:off + rows * cols], frag(W)), (off, rows, cols)
            if flags & 2:
                assert torch.equal(st.shadow_tf[off:off + rows * cols], frag(W.t().contiguous())), (off, rows, cols)
            n += 1
        return n
    b = synth.make_batch("sap", batch_size=4, seed=5, step=0, vocab=600, min_len=8, max_len=15, min_steps=2, max_steps=3)
    bd, plan = synth.batch_to(b, DEV), build_plan(b, "sap", DEV)
    rw = torch.tensor(RW, device=DEV)
    tr = PretrainStep(g_s, g_t, lr=1e-3, warmup_steps=1, num_train_steps=10)
    before = st.shadow.clone()
    tr.step(bd, "sap", rw=rw, plan=plan)                       # (a)
    assert consistent() >= 10 and not torch.equal(before, st.shadow)
    opt = torch.optim.SGD(g_s.parameters(), lr=0.5)             # (b)
    out = g_s(bd, "sap", compute_loss=True, teacher_outputs=None, plan=plan)
    out["loss"].backward()
    mid = st.shadow.clone()
    opt.step()
    with torch.no_grad():
        g_s(bd, "sap", compute_loss=False, plan=plan)           # the forward that follows syncs the shadows
    assert consistent() >= 10 and not torch.equal(mid, st.shadow)
    sd = {k: v.clone() * 0.5 for k, v in g_s.state_dict().items()}      # (c)
    g_s.load_state_dict(sd)
    with torch.no_grad():
        g_s(bd, "sap", compute_loss=False, plan=plan)
    consistent()
    assert torch.allclose(st.shadow.float(), (st.flat * 1.0).to(torch.bfloat16).float())
