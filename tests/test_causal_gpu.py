"""Causal-intervention blocks on the GPU (-m gpu): one navigator step -- language (back-door over the
instruction z-dictionary + front-door text), panorama (back-door over the room-type image dictionary), navigation (front-door
viewpoint / map dictionaries) -- driven as GMapNavAgent.rollout drives the model (agent.py:76-89,:162-172,:942-944), fp32 engine
against the fp64 oracle (oracle/causal_ref.py, oracle/nav_ref.py): outputs, loss, every parameter gradient including the blocks'.
The blocks' arithmetic is this build's restatement (parity unpinned); their inputs are the reference's."""
import pytest
import torch
import torch.nn.functional as F

import magic_amd  # noqa: F401
from magic_amd.host.config import make_config
from magic_amd.host.model_nav import VLNBert
from oracle.nav_ref import RefVLNBert
from tests.test_nav_gpu import nav_inputs, to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dictionaries(B, H, seed=0):
    g = torch.Generator().manual_seed(seed)
    rep = lambda t: t.unsqueeze(0).repeat(B, *([1] * t.dim()))            # the agent repeats ONE dictionary over the batch (agent.py:78-81)
    pz = lambda n: rep(torch.softmax(torch.randn(n, generator=g), 0).reshape(n, 1))
    return dict(instr_z_direction_features=rep(torch.randn(5, H, generator=g)), instr_z_direction_pzs=pz(5),
                instr_z_landmark_features=rep(torch.randn(6, H, generator=g)), instr_z_landmark_pzs=pz(6),
                front_txt_feats=rep(torch.randn(7, H, generator=g)), z_img_features=rep(torch.randn(9, 768, generator=g)), z_img_pzs=pz(9),
                front_vp_feats=rep(torch.randn(6, H, generator=g)), front_gmap_feats=rep(torch.randn(6, H, generator=g)))


def step(model, inp, dz):
    B = inp["txt_ids"].shape[0]
    txt, _ = model("language", dict(txt_ids=inp["txt_ids"], txt_masks=inp["txt_masks"], **{k: dz[k] for k in (
        "instr_z_direction_features", "instr_z_direction_pzs", "instr_z_landmark_features", "instr_z_landmark_pzs", "front_txt_feats")}))
    pe, pm, pf, ia = model("panorama", dict(view_img_fts=inp["view_img_fts"], loc_fts=inp["loc_fts"], nav_types=inp["nav_types"],
                                            view_lens=inp["view_lens"], already_dropout=True, z_img_features=dz["z_img_features"], z_img_pzs=dz["z_img_pzs"]))
    H, Kn = txt.shape[-1], inp["Kn"]
    rows = []
    for b in range(B):
        r = [pe.new_zeros(H), pe.new_zeros(H)]
        r += [pf[b] * (0.5 + 0.1 * i) for i in range(int(inp["nv"][b]))]
        r += [pe[b, i % int(inp["n_cand"][b])] for i in range(int(inp["nu"][b]))]
        r += [pe.new_zeros(H)] * (Kn - len(r))
        rows.append(torch.stack(r))
    nav = model("navigation", dict(gmap_img_embeds=torch.stack(rows), vp_img_embeds=torch.cat([pe.new_zeros(B, 2, H), pe], 1), txt_embeds=txt,
                                   front_txt_feats=dz["front_txt_feats"], front_vp_feats=dz["front_vp_feats"], front_gmap_feats=dz["front_gmap_feats"],
                                   **{k: inp[k] for k in ("txt_masks", "gmap_masks", "vp_masks", "gmap_step_ids", "gmap_pos_fts", "gmap_pair_dists",
                                                          "gmap_visited_masks", "gmap_vpids", "vp_pos_fts", "vp_nav_masks", "vp_cand_vpids")}))
    ce = F.cross_entropy(nav["fused_logits"].float() if nav["fused_logits"].dtype != torch.float64 else nav["fused_logits"],
                         inp["targets"].to(nav["fused_logits"].device), reduction="none", ignore_index=-100)
    return dict(txt=txt, pano=pe, nav=nav, loss=ce.sum() / B + 0.1 * txt.float().square().mean() + 0.1 * pe.float().square().mean())


def close(a, b, name, rtol=3e-4, atol=3e-5):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{name}: max|err| {(a - b).abs().max().item():.3e} (ref max {b.abs().max().item():.3e})"


@pytest.mark.parametrize("add_method,txt_type,img_type", [("add", "type_2", "type_1"), ("door", "type_2", "type_2"), ("add", "type_1", "type_1")])
def test_all_five_blocks_forward_loss_and_gradients_match_oracle(add_method, txt_type, img_type):
    kw = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, num_l_layers=2, num_x_layers=1, num_pano_layers=1,
              do_back_txt=True, do_back_img=True, do_front_txt=True, do_front_img=True, do_front_his=True,
              do_back_txt_type=txt_type, do_back_imgobj_type=img_type, do_add_method=add_method)
    cfg = make_config(128, role="student", **kw)
    torch.manual_seed(5)
    o = RefVLNBert(cfg).double().eval()
    with torch.no_grad():
        for n, p in o.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.02)
    g = VLNBert(None, role="student", config=cfg, device=DEV, compute_dtype=torch.float32)
    missing = g.load_state_dict(o.state_dict(), strict=True)
    assert any(k.startswith("vln_bert.causal.back_txt.") for k in g.state_dict()) and len(g.causal_blocks) == 5
    g.eval()
    inp = nav_inputs(B=4, L=13, seed=2)
    dz = dictionaries(4, 128, seed=3)
    want = step(o, to_dev(inp, "cpu", f64=True), to_dev(dz, "cpu", f64=True))
    want["loss"].backward()
    g.store.zero_grad()
    got = step(g, to_dev(inp, DEV), to_dev(dz, DEV))
    close(got["txt"], want["txt"], "adjusted txt_embeds")
    close(got["pano"], want["pano"], "adjusted pano_embeds")
    for k in ("gmap_embeds", "vp_embeds", "cls_embeds"):
        close(got["nav"][k], want["nav"][k], k)
    a, b = got["nav"]["fused_logits"].cpu(), want["nav"]["fused_logits"]
    assert torch.equal(torch.isinf(a), torch.isinf(b)) and torch.equal(a.argmax(1), b.argmax(1))
    close(torch.nan_to_num(a, neginf=0), torch.nan_to_num(b, neginf=0), "fused_logits", 1e-4, 1e-5)
    close(got["loss"], want["loss"], "loss", 1e-4, 1e-6)
    got["loss"].backward()
    torch.cuda.synchronize()
    params = dict(g.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in o.parameters() if p.grad is not None)
    n_causal = 0
    for name, p in o.named_parameters():
        if p.grad is None:
            continue
        close(params[name].grad, p.grad, f"grad {name}", 3e-3, 2e-3 * p.grad.abs().max().item() + 3e-6 * gmax)
        n_causal += ".causal." in name
    # type_1 blocks have no use for their query / key projections: those parameters get no gradient on either side
    assert n_causal >= (5 * 8 + (20 if add_method == "door" else 0)) - 2 * 4 * ((txt_type == "type_1") + (img_type == "type_1"))


def test_bf16_blocks_run_and_dictionary_without_switch_is_refused():
    kw = dict(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, vocab_size=300, num_l_layers=1, num_x_layers=1, num_pano_layers=1)
    g = VLNBert(None, role="student", config=make_config(128, role="student", do_back_txt=True, do_front_img=True, **kw), device=DEV)
    g.train()
    inp, dz = to_dev(nav_inputs(B=3, L=11, seed=1), DEV), to_dev(dictionaries(3, 128), DEV)
    txt, _ = g("language", dict(txt_ids=inp["txt_ids"], txt_masks=inp["txt_masks"], **{k: dz[k] for k in (
        "instr_z_direction_features", "instr_z_direction_pzs", "instr_z_landmark_features", "instr_z_landmark_pzs")}))
    g.store.zero_grad()
    txt.float().square().mean().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(txt.float()).all() and torch.isfinite(g.store.grad).all()
    assert dict(g.named_parameters())["vln_bert.causal.back_txt.value.weight"].grad.abs().max() > 0
    with pytest.raises(ValueError, match="front_txt_feats"):            # do_front_txt is off in this config
        g("language", dict(txt_ids=inp["txt_ids"], txt_masks=inp["txt_masks"], front_txt_feats=dz["front_txt_feats"]))


# ---- the PRETRAINING model with its back-door inputs (pretrain_src/data/tasks.py:156-164, :441-449) -----------------------------------------
def _pretrain_dicts(batch, g, n_dir=5, n_lm=7, n_img=9, H=128, D=768):
    """the collate's output shape: ONE dictionary repeated over the batch (instruction: direction + landmark entries of the model's width, priors
    [B, Nz, 1]); the image dictionary repeated over the trajectory's panoramas (`traj_img_len x Nz x 768`, priors `x 1`)"""
    B, Np = len(batch["traj_step_lens"]), batch["traj_view_img_fts"].shape[0]
    rep = lambda t, n: t.unsqueeze(0).repeat(n, *([1] * t.dim()))
    pz = lambda n: torch.softmax(torch.randn(n, generator=g), 0).unsqueeze(-1)
    batch = dict(batch)
    batch["instr_z_direction_features"], batch["instr_z_direction_pzs"] = rep(torch.randn(n_dir, H, generator=g), B), rep(pz(n_dir), B)
    batch["instr_z_landmark_features"], batch["instr_z_landmark_pzs"] = rep(torch.randn(n_lm, H, generator=g), B), rep(pz(n_lm), B)
    batch["img_z_features"], batch["img_z_pzs"] = rep(torch.randn(n_img, D, generator=g), Np), rep(pz(n_img), Np)
    return batch


@pytest.mark.parametrize("task,method,itype", [("sap", "add", "type_2"), ("mlm", "door", "type_1"), ("cfp", "add", "type_1")])
def test_pretraining_model_serves_the_back_door_inputs_of_its_collates(task, method, itype):
    """GlocalTextPathCMTPreTraining with do_back_txt / do_back_img on (off in the shipped config, r2r_magic_model_config.json:60-66): the
    `instr_z_*` / `img_z_*` keys the collates add are served by the navigation model's blocks at the same two places (text encoder output,
    panorama view embeddings) -- fp32 engine vs fp64 oracle: every output, every loss term, every parameter gradient incl. the blocks'; and
    such a key with its switch off raises."""
    from magic_amd.host import synth
    from tests.test_model_gpu import RW, build, close, to64, view_outputs
    # (one batch feeds teacher and student: the instruction dictionary has ONE width, here the student's -- `back_txt_dict_size`)
    extra = dict(do_back_txt=True, do_back_img=True, do_back_txt_type="type_2", do_back_imgobj_type=itype, do_add_method=method, back_txt_dict_size=128)
    o_t, o_s, g_t, g_s = build(torch.float32, **extra)
    assert set(g_s.causal_blocks) == {"back_txt", "back_img"} and any(k.startswith("bert.causal.back_img.") for k in g_s.state_dict())
    g = torch.Generator().manual_seed(11)
    batch = _pretrain_dicts(synth.make_batch(task, batch_size=5, seed=31, vocab=600, min_len=8, max_len=17, min_steps=2, max_steps=4), g)
    rw = torch.tensor(RW, dtype=torch.float64)
    o_t, o_s = o_t.double(), o_s.double()
    b64 = to64(batch)
    with torch.no_grad():
        ot = o_t(b64, task, compute_loss=True)["outputs"]
    want = o_s(b64, task, compute_loss=True, teacher_outputs=ot, rw=rw)
    want["loss"].backward()
    with torch.no_grad():
        gt = g_t(batch, task, compute_loss=False, return_outputs=True)
    plan = gt["plan"]
    for k, v in view_outputs(gt, plan, 256).items():
        close(v, ot[k], f"teacher {k}", 2e-4, 2e-5)
    g_s.store.zero_grad()
    got = g_s(batch, task, compute_loss=True, teacher_outputs=gt, rw=RW, plan=plan)
    for k, v in view_outputs(got["outputs"], plan, 128).items():
        close(v, want["outputs"][k], f"student {k}", 2e-4, 2e-5)
    close(got["supervised_loss"], want["supervised_loss"], "supervised loss", 1e-4, 1e-6)
    for k, v in want["kdl_terms"].items():
        close(got["kdl_terms"][k], v, f"kd term {k}", 2e-4, 1e-7)
    close(got["loss"], want["loss"], "total loss", 1e-4, 1e-6)
    got["loss"].backward()
    torch.cuda.synchronize()
    params = dict(g_s.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in o_s.parameters() if p.grad is not None)
    n_blocks = 0
    for name, p in o_s.named_parameters():
        gr = params[name].grad
        if p.grad is None:
            assert gr.abs().max().item() == 0.0, name
            continue
        close(gr, p.grad, f"grad {name}", 2e-3, 1e-3 * p.grad.abs().max().item() + 2e-6 * gmax)
        n_blocks += ".causal." in name and p.grad.abs().max().item() > 0
    assert n_blocks >= 10, n_blocks                 # both blocks' projections, norms (+ the gate) carried gradient
    # the same keys into a model whose switches are off: an error, never ignored
    _, _, _, plain = build(torch.float32)
    with pytest.raises(ValueError, match="do_back"):
        plain(batch, task, compute_loss=False)
