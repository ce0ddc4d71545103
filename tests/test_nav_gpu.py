"""Navigation-time API parity (-m gpu): VLNBert('language' | 'panorama' | 'navigation') + the drop-in MAKD
primitives, driven exactly like GMapNavAgent.rollout drives them (agent.py:785-1024) for one step, with
torch autograd composing our per-mode Functions; checked against the CPU oracle (oracle/nav_ref.py +
oracle/makd_ref.nav_makd, the latter pinned to the reference's compute_kd_losses by golden vectors)."""
from collections import defaultdict

import pytest
import torch
import torch.nn.functional as F

import magic_amd  # noqa: F401
from magic_amd.host import kd_loss as K
from magic_amd.host.config import make_config
from magic_amd.host.model_nav import VLNBert
from oracle import makd_ref as M
from oracle.nav_ref import RefVLNBert

pytestmark = pytest.mark.gpu
DEV = "cuda"
HEADS = ("txt_emb_w", "kdl_img_w", "kdl_avg_img_w", "global_cross_w", "local_cross_w")


def nav_inputs(B=4, L=14, V=36, seed=0):
    g = torch.Generator().manual_seed(seed)
    txt_lens = torch.randint(6, L + 1, (B,), generator=g)
    txt_lens[0] = L
    txt_ids = torch.randint(3, 290, (B, L), generator=g)
    txt_masks = torch.arange(L)[None] < txt_lens[:, None]
    txt_ids[~txt_masks] = 0
    n_cand = torch.randint(2, 6, (B,), generator=g)
    view_fts = torch.randn(B, V, 768, generator=g)
    ang = torch.rand(B, V, 2, generator=g) * 6.28
    loc = torch.cat([ang.sin()[..., :1], ang.cos()[..., :1], ang.sin()[..., 1:], ang.cos()[..., 1:], torch.ones(B, V, 3)], -1)
    nav_types = (torch.arange(V)[None] < n_cand[:, None]).long()
    view_lens = torch.full((B,), V)
    # map: [stop, mem, visited x nv, unvisited x nu]
    nv = torch.randint(1, 4, (B,), generator=g)
    nu = torch.randint(2, 5, (B,), generator=g)
    Kn = int((2 + nv + nu).max())
    gmap_vpids, vis, gmask, step_ids = [], torch.zeros(B, Kn, dtype=torch.bool), torch.zeros(B, Kn, dtype=torch.bool), torch.zeros(B, Kn, dtype=torch.long)
    vp_cand = []
    for b in range(B):
        ids = [None, None] + [f"v{b}_{i}" for i in range(int(nv[b]))] + [f"u{b}_{i}" for i in range(int(nu[b]))]
        gmap_vpids.append(ids + [None] * (Kn - len(ids)))
        vis[b, 1:2 + int(nv[b])] = True                     # [0,1,1..,0..] agent.py:200
        gmask[b, :len(ids)] = True
        gmask[b, 1] = False                                  # mem column masked out (agent.py:232-233)
        step_ids[b, 2:2 + int(nv[b])] = torch.arange(1, int(nv[b]) + 1)
        cands = []
        for j in range(int(n_cand[b])):                      # candidates: some unvisited map nodes, one visited (backtrack)
            cands.append(f"u{b}_{j}" if j < int(nu[b]) - 1 else f"v{b}_0")
        vp_cand.append([None, None] + cands + [None] * (V - len(cands)))
    pos = torch.rand(B, Kn, 7, generator=g)
    d = torch.rand(B, Kn, Kn, generator=g) * 20
    d = (d + d.transpose(1, 2)) / 2
    d[:, :2] = 0
    d[:, :, :2] = 0
    vp_pos = torch.rand(B, V + 2, 14, generator=g)
    vp_masks = torch.ones(B, V + 2, dtype=torch.bool)
    vp_nav = torch.cat([torch.ones(B, 1, dtype=torch.bool), torch.zeros(B, 1, dtype=torch.bool), nav_types == 1], 1)
    targets = torch.tensor([2 + int(nv[b]) for b in range(B)])          # first unvisited node
    targets[-1] = -100
    return dict(txt_ids=txt_ids, txt_masks=txt_masks, view_img_fts=view_fts, loc_fts=loc, nav_types=nav_types, view_lens=view_lens,
                gmap_vpids=gmap_vpids, gmap_visited_masks=vis, gmap_masks=gmask, gmap_step_ids=step_ids, gmap_pos_fts=pos,
                gmap_pair_dists=d, vp_pos_fts=vp_pos, vp_masks=vp_masks, vp_nav_masks=vp_nav, vp_cand_vpids=vp_cand,
                nv=nv, nu=nu, n_cand=n_cand, targets=targets, Kn=Kn)


def to_dev(d, dev, f64=False):
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            v = v.to(dev)
            if f64 and v.is_floating_point():
                v = v.double()
        out[k] = v
    return out


def one_step(model, inp, heads=None, teacher=None, t_heads=None, mse_fn=None, kd_fn=None, rw=None, teacher_out=None):
    """language -> panorama -> (agent-style map embedding assembly) -> navigation -> CE (+ MAKD vs teacher_out)."""
    B = inp["txt_ids"].shape[0]
    txt, txt_attn = model("language", dict(txt_ids=inp["txt_ids"], txt_masks=inp["txt_masks"]))
    pe, pm, pf, ia = model("panorama", dict(view_img_fts=inp["view_img_fts"], loc_fts=inp["loc_fts"], nav_types=inp["nav_types"],
                                            view_lens=inp["view_lens"], already_dropout=True))
    H = txt.shape[-1]
    Kn = inp["Kn"]
    rows = []
    for b in range(B):          # GraphMap-style assembly: visited nodes <- fused pano, unvisited <- candidate view embeds (agent.py:905-924)
        r = [pe.new_zeros(H), pe.new_zeros(H)]
        r += [pf[b] * (0.5 + 0.1 * i) for i in range(int(inp["nv"][b]))]
        r += [pe[b, i % int(inp["n_cand"][b])] for i in range(int(inp["nu"][b]))]
        r += [pe.new_zeros(H)] * (Kn - len(r))
        rows.append(torch.stack(r))
    gmap_img = torch.stack(rows)
    vp_img = torch.cat([pe.new_zeros(B, 2, H), pe], 1)
    nav = model("navigation", dict(gmap_img_embeds=gmap_img, vp_img_embeds=vp_img, txt_embeds=txt, **{k: inp[k] for k in (
        "txt_masks", "gmap_masks", "vp_masks", "gmap_step_ids", "gmap_pos_fts", "gmap_pair_dists", "gmap_visited_masks", "gmap_vpids",
        "vp_pos_fts", "vp_nav_masks", "vp_cand_vpids")}))
    out = defaultdict(lambda: None)
    out.update(txt_embeds=txt, txt_attns=txt_attn, pano_embeds=pe, pano_fused_embeds=pf, img_attns=ia, nav_outs=nav,
               nav_logits=nav["fused_logits"])
    ce = F.cross_entropy(nav["fused_logits"].float(), inp["targets"], reduction="none", ignore_index=-100)
    res = dict(out=out, ce=ce)
    if teacher_out is not None:
        teacher_out["sample_weights"] = M.exponential_decay(
            F.cross_entropy(teacher_out["nav_logits"].float().detach(), inp["targets"], reduction="none", ignore_index=-100), 0.7).detach()
        acc = defaultdict(float)
        M.nav_makd(0, out, teacher_out, heads, acc, role="t2s", loss_type="sum", temperature=2.0, weights=rw, weight_mode="RW",
                   mse_fn=mse_fn, kd_fn=kd_fn)
        res["kd"] = acc
        res["loss"] = M.episode_loss(sum(acc.values()), ce.sum(), B, 1.0, 0.5)
    else:
        res["loss"] = ce.sum() / B
    return res


def test_nav_step_forward_losses_and_gradients_match_oracle_fp32():
    kw = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, num_l_layers=2, num_x_layers=1, num_pano_layers=1)
    tcfg, scfg = make_config(256, role="teacher", **kw), make_config(128, role="student", teacher_hidden_size=256, **kw)
    torch.manual_seed(0)
    o_t, o_s = RefVLNBert(tcfg).double().eval(), RefVLNBert(scfg).double().eval()
    with torch.no_grad():
        for m in (o_t, o_s):
            for n, p in m.named_parameters():
                if n.endswith("bias"):
                    p.normal_(0, 0.02)
    g_t = VLNBert(None, role="teacher", config=tcfg, device=DEV, compute_dtype=torch.float32)
    g_s = VLNBert(None, role="student", config=scfg, device=DEV, compute_dtype=torch.float32)
    g_t.load_state_dict(o_t.state_dict())
    g_s.load_state_dict(o_s.state_dict())
    inp = nav_inputs()
    rw = [1.2, 0.8, 1.1, 0.9, 1.0]
    # oracle (fp64)
    i64 = to_dev(inp, "cpu", f64=True)
    with torch.no_grad():
        ot = one_step(o_t, i64)["out"]
    heads = {n: getattr(o_s.vln_bert, n) for n in HEADS}
    want = one_step(o_s, i64, heads=heads, teacher_out=ot, rw=rw)
    want["loss"].backward()
    # engine
    idev = to_dev(inp, DEV)
    with torch.no_grad():
        gt = one_step(g_t, idev)["out"]
    g_s.store.zero_grad()
    gheads = {n: getattr(g_s.vln_bert, n) for n in HEADS}
    got = one_step(g_s, idev, heads=gheads, teacher_out=gt, rw=rw,
                   mse_fn=lambda a, b, w, lt: K.mse_loss(a, b, w, lt), kd_fn=lambda s, t, T, w, lt: K.kd_loss(s, t, T, t_sample_weights=w, loss_type=lt))

    def close(a, b, name, rtol=2e-4, atol=2e-5):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{name}: max|err| {(a - b).abs().max().item():.3e} (ref {b.abs().max().item():.3e})"
    for k in ("txt_embeds", "txt_attns", "pano_embeds", "pano_fused_embeds", "img_attns"):
        close(got["out"][k], want["out"][k], k)
    for k in ("gmap_embeds", "vp_embeds", "gmap_attns", "vp_attns", "cls_embeds"):
        close(got["out"]["nav_outs"][k], want["out"]["nav_outs"][k], k)
    for k in ("global_logits", "local_logits", "fused_logits"):
        a, b = got["out"]["nav_outs"][k].cpu(), want["out"]["nav_outs"][k]
        assert torch.equal(torch.isinf(a), torch.isinf(b)), k
        close(torch.nan_to_num(a, neginf=0), torch.nan_to_num(b, neginf=0), k, 1e-4, 1e-5)
        assert torch.equal(a.argmax(1), b.argmax(1)), f"{k}: action argmax must be bit-exact"
    for k, v in want["kd"].items():
        close(torch.as_tensor(got["kd"][k]), torch.as_tensor(v), f"kd {k}", 3e-4, 1e-6)
    close(got["loss"], want["loss"], "episode loss", 2e-4, 1e-6)
    got["loss"].backward()
    torch.cuda.synchronize()
    params = dict(g_s.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in o_s.parameters() if p.grad is not None)
    n = 0
    for name, p in o_s.named_parameters():
        g = params[name].grad
        if p.grad is None:
            assert g.abs().max().item() == 0.0, name
            continue
        close(g, p.grad, f"grad {name}", 2e-3, 1e-3 * p.grad.abs().max().item() + 2e-6 * gmax)
        n += 1
    assert n > 40


def test_drop_in_kd_primitives_match_reference_golden_vectors(golden_dir):
    """The GPU kd_loss / mse_loss replacements reproduce the numbers minted from the reference's own functions."""
    import os
    fx = torch.load(os.path.join(golden_dir, "makd_primitives.pt"), weights_only=False)
    s, t, w, c = fx["s"].to(DEV), fx["t"].to(DEV), fx["w"].to(DEV), fx["cases"]
    chk = lambda a, b, n: torch.testing.assert_close(a.cpu(), b, rtol=2e-5, atol=2e-6, msg=n)
    for T in (1, 2):
        chk(K.kd_loss_pretrain(s, t, temperature=T), c[f"pre_kd_T{T}"], "pre kd")
        chk(K.kd_loss_pretrain(s, t, temperature=T, t_sample_weights=w), c[f"pre_kd_T{T}_w"], "pre kd w")
        for lt in ("sum", "mean"):
            chk(K.kd_loss(s, t, temperature=T, loss_type=lt), c[f"nav_kd_T{T}_{lt}"], "nav kd")
            chk(K.kd_loss(s, t, temperature=T, t_sample_weights=w, loss_type=lt), c[f"nav_kd_T{T}_{lt}_w"], "nav kd w")
    fs, ft, fa, fb = fx["fs"].to(DEV), fx["ft"].to(DEV), fx["fa"].to(DEV), fx["fb"].to(DEV)
    chk(K.mse_loss_pretrain(fs, ft), c["pre_mse"], "pre mse")
    chk(K.mse_loss_pretrain(fs, ft, w), c["pre_mse_w"], "pre mse w")
    chk(K.mse_loss_pretrain(fs, ft, fx["wbad"].to(DEV)), c["pre_mse_wbad"], "pre mse wbad")
    chk(K.mse_loss_pretrain(fa, fb, w), c["pre_mse4_w"], "pre mse4")
    for lt in ("sum", "mean"):
        chk(K.mse_loss(fs, ft, loss_type=lt), c[f"nav_mse_{lt}"], "nav mse")
        chk(K.mse_loss(fs, ft, w, lt), c[f"nav_mse_{lt}_w"], "nav mse w")
        chk(K.mse_loss(fa, fb, w, lt), c[f"nav_mse4_{lt}_w"], "nav mse4 w")
    with pytest.raises(ValueError):
        K.mse_loss(fs, ft, fx["wbad"].to(DEV))
    chk(K.exponential_decay(fx["losses"].to(DEV), 0.7), c["exp_decay_0.7"], "exp decay")
    chk(K.invert_normalized_losses(fx["losses"].to(DEV)), c["invert_norm"], "invert")
    with pytest.raises(magic_amd.host.lib.MagicHipError):
        K.mse_loss(fs.cpu(), ft.cpu())


def test_critic_head_api_and_numerics():
    """agent.py:39 `Critic(args).cuda()`, agent_base.py:116-139 (.parameters() into an optimizer, .train(), state_dict)."""
    from types import SimpleNamespace
    from magic_amd.host.model_nav import Critic
    args = SimpleNamespace(hidden_size=128, dropout=0.0)
    c = Critic(args, compute_dtype=torch.float32).cuda()
    sd = c.state_dict()
    assert set(sd) == {"state2value.0.weight", "state2value.0.bias", "state2value.3.weight", "state2value.3.bias"}
    x = torch.randn(6, 128, device=DEV, requires_grad=True)
    w0, b0, w3, b3 = (sd[k].clone() for k in ("state2value.0.weight", "state2value.0.bias", "state2value.3.weight", "state2value.3.bias"))
    c.train()
    v = c(x)
    xr = x.detach().clone().requires_grad_(True)
    want = (torch.relu(xr @ w0.t() + b0) @ w3.t() + b3).squeeze()
    assert v.shape == (6,) and torch.allclose(v, want, rtol=1e-4, atol=1e-5)
    opt = torch.optim.AdamW(c.parameters(), lr=1e-3)
    opt.zero_grad()
    v.sum().backward()
    want.sum().backward()
    assert torch.allclose(x.grad, xr.grad, rtol=1e-4, atol=1e-5)
    g0 = dict(c.named_parameters())["state2value.0.weight"].grad
    assert g0 is not None and g0.abs().max() > 0
    opt.step()
    assert not torch.equal(c.state_dict()["state2value.0.weight"], w0)


def test_unmodified_loop_with_a_torch_optimizer_zero_grad_none_and_bf16_shadow():
    """The reference loops call `optimizer.zero_grad()` (torch >= 2: grads become None) and step a stock torch optimizer
    (agent_base.py:128-139, optim/misc.py:12-37).  Gradients must reappear as views of the flat buffer, must not
    accumulate across steps, and the bf16 shadow weights must follow the optimizer's in-place update."""
    from magic_amd.host import synth
    from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining
    kw = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, num_l_layers=1, num_x_layers=1, num_pano_layers=1)
    model = GlocalTextPathCMTPreTraining(make_config(128, **kw), device=DEV, compute_dtype=torch.bfloat16, seed=3)
    opt = torch.optim.AdamW(model.parameters(), lr=3e-4)
    batch = synth.make_batch("sap", batch_size=4, seed=2, vocab=300, min_len=6, max_len=10, min_steps=2, max_steps=3)
    name = "bert.lang_encoder.layer.0.attention.self.query.weight"
    p = dict(model.named_parameters())[name]
    grads, losses = [], []
    for it in range(3):
        opt.zero_grad()                                   # set_to_none=True
        assert p.grad is None
        out = model(batch, "sap", compute_loss=True)
        out["loss"].backward()
        assert p.grad is not None and p.grad.data_ptr() == model.store.g(name).data_ptr()
        grads.append(p.grad.clone())
        losses.append(float(out["loss"].detach()))
        before = p.detach().clone()
        opt.step()
        assert not torch.equal(before, p.detach())
    # same batch, tiny steps: gradient magnitudes stay comparable (no accumulation: 3rd would be ~3x the 1st)
    r = grads[2].norm().item() / grads[0].norm().item()
    assert 0.3 < r < 1.8, r
    assert losses[2] < losses[0]                          # the bf16 weights the kernels read follow the torch optimizer
    model(batch, "sap", compute_loss=False)
    assert torch.equal(model.store.w(name).float(), p.detach().to(torch.bfloat16).float())


def test_nav_modes_with_dropout_train_vs_eval():
    """vln_bert.train() (agent rollout under feedback='sample') arms the in-kernel dropout of every mode; eval() is
    deterministic; gradients stay finite and flow into the flat buffer."""
    from types import SimpleNamespace
    from magic_amd.host.model_nav import VLNBert
    cfg = make_config(128, vocab_size=300, num_l_layers=1, num_x_layers=1, num_pano_layers=1)      # dropouts 0.1 (config default)
    m = VLNBert(None, role="student", device=DEV, compute_dtype=torch.float32, config=cfg)
    B, L = 3, 9
    g = torch.Generator().manual_seed(0)
    txt = dict(txt_ids=torch.randint(3, 290, (B, L), generator=g).to(DEV), txt_masks=torch.ones(B, L, dtype=torch.bool, device=DEV))
    m.eval()
    e1, _ = m("language", txt)
    e2, _ = m("language", txt)
    assert torch.equal(e1, e2) and m.net.drop is None
    m.train()
    t1, _ = m("language", txt)
    s1 = m.net.drop[0].clone()
    t2, _ = m("language", txt)
    assert not torch.equal(m.net.drop[0], s1) and not torch.equal(t1, t2)          # fresh seed per call
    assert (t1 - e1).abs().max() > 1e-3
    m.store.zero_grad()
    t1.float().square().sum().backward()
    assert torch.isfinite(m.store.grad).all() and m.store.grad.abs().max() > 0


def test_f4_modes_dictionaries_off_match_the_oracle_and_refuse_dictionary_inputs_when_switched_off():
    """SURVEY §8 f-4 in its dictionaries-off form: 'instr_zdict_update' (agent.py:1231-1233) and 'extract_cfp_features'
    (agent.py:1535-1541) against the CPU oracle in fp32; a back-door / front-door input is refused, never ignored."""
    from magic_amd.host import synth
    kw = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, num_l_layers=2, num_x_layers=1, num_pano_layers=1)
    cfg = make_config(128, role="student", **kw)
    torch.manual_seed(3)
    o = RefVLNBert(cfg).double().eval()
    g = VLNBert(None, role="student", config=cfg, device=DEV, compute_dtype=torch.float32)
    g.load_state_dict(o.state_dict())
    g.eval()

    def close(a, b, name, rtol=2e-4, atol=2e-5):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        assert a.shape == b.shape, (name, a.shape, b.shape)
        assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{name}: max|err| {(a - b).abs().max().item():.3e}"
    inp = nav_inputs(B=5, L=17, seed=4)
    zin = dict(z_txt=inp["txt_ids"], z_txt_mask=inp["txt_masks"], instr_z_direction_features=None, instr_z_direction_pzs=None,
               instr_z_landmark_features=None, instr_z_landmark_pzs=None, front_txt_feats=None)
    with torch.no_grad():
        want = o("instr_zdict_update", zin)[0]
        got = g("instr_zdict_update", to_dev(zin, DEV))[0]
    close(got, want, "instr_zdict_update")
    # whole-trajectory batch in GMapNavAgent.cfp_collate's layout (tensors already on the GPU, agent.py:1508-1512)
    batch = synth.make_batch("cfp", batch_size=6, seed=11, step=0, vocab=300)
    b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    with torch.no_grad():
        want = o("extract_cfp_features", b64)
        got = g("extract_cfp_features", to_dev(batch, DEV))
    assert set(got) == {"txt_outputs", "vp_outputs", "gmap_outputs"}
    for k in got:
        close(got[k], want[k], k)
    # a dictionary input while its do_back_* / do_front_* switch is off (this config: all off) is refused, never ignored;
    # with the switches on: tests/test_causal_gpu.py
    bad = dict(zin, instr_z_direction_features=torch.zeros(5, 3, 128))
    with pytest.raises(ValueError, match="instr_z_direction_features"):
        g("instr_zdict_update", bad)
    with pytest.raises(ValueError, match="front_txt_feats"):
        g("language", dict(txt_ids=inp["txt_ids"].to(DEV), txt_masks=inp["txt_masks"].to(DEV), front_txt_feats=torch.zeros(5, 4, 128)))
