"""ORACLE (test infrastructure): CPU restatement of the MAKD distillation arithmetic.

Follows /root/reference/pretrain_src/optim/kd_loss.py:5-54 ("pretrain" flavour: mean reductions) and
/root/reference/map_nav_src/utils/kd_loss.py:6-67 ("nav" flavour: loss_type sum|mean), and the
aggregation of /root/reference/map_nav_src/r2r/agent.py:546-719 (compute_kd_losses).
PINNED: tests/test_oracle_golden.py checks every function here against tests/golden/makd_*.pt,
minted by tests/golden/mint_golden.py from the reference's own code in the authoring container.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import torch
import torch.nn.functional as F


def _bw(w, x):
    return w.view(-1, *([1] * (x.dim() - 1)))


def mse_loss(s, t, w=None, loss_type="sum", flavour="nav"):
    """kd_loss.py mse_loss. pretrain flavour: always mean; silently unweighted when w.shape[0] != B
    (pretrain kd_loss.py:11-16).  nav flavour: ValueError on mismatch (nav kd_loss.py:17)."""
    e = (s - t) ** 2
    if flavour == "pretrain":
        if w is not None and e.shape[0] == w.shape[0]:
            e = e * _bw(w, e)
        return e.mean()
    if w is not None:
        if e.shape[0] != w.shape[0]:
            raise ValueError("Shape mismatch between sample weights and inputs")
        e = e * _bw(w, e)
    if loss_type == "sum":
        return e.sum()
    if loss_type == "mean":
        return e.mean()
    raise ValueError("Unsupported loss_type. Choose 'sum' or 'mean'.")


def kd_loss(s_logits, t_logits, temperature=1, w=None, loss_type="sum", flavour="nav"):
    """Temperature KL (kd_loss.py kd_loss): -inf -> -1e6, p_t = softmax(t/T), log p_s = log_softmax(s/T);
    unweighted: elementwise KL reduced by sum or mean over ALL elements (not batchmean);
    weighted: per-row KL summed over dim 1, times w, then sum / mean over rows; all times T^2."""
    if flavour == "pretrain":
        loss_type = "mean"
    neg = float("-inf")
    s = torch.where(s_logits == neg, torch.full_like(s_logits, -1e6), s_logits)
    t = torch.where(t_logits == neg, torch.full_like(t_logits, -1e6), t_logits)
    pt = torch.softmax(t / temperature, dim=1)
    ls = torch.log_softmax(s / temperature, dim=1)
    kl = torch.where(pt > 0, pt * (pt.log() - ls), torch.zeros_like(pt))     # == torch.kl_div pointwise
    if w is None:
        r = kl.sum() if loss_type == "sum" else kl.mean()
    else:
        row = kl.sum(1) * w.view(-1)
        r = row.sum() if loss_type == "sum" else row.mean()
    return r * temperature ** 2


def exponential_decay(losses, decay_rate=0.1):
    return torch.exp(-decay_rate * losses)


def invert_normalized_losses(losses):
    lo, hi = losses.min(), losses.max()
    return 1 - (losses - lo) / (hi - lo)


ABILITY_INDEX = dict(txt=0, img=1, glob=2, local=3, action=4)       # agent.py:585-716
LOSS_KEYS = ("txt_emb_loss", "txt_attn_loss", "img_emb_loss", "avg_img_emb_loss", "img_attn_loss",
             "global_emb_loss", "global_attn_loss", "local_emb_loss", "local_attn_loss", "predict_loss")


def nav_makd(t_step, s_out, t_out, proj, acc, *, role="t2s", loss_type="sum", temperature=2.0,
             abilities=("txt", "img", "global", "local", "action"), weights=None, weight_mode="RW",
             no_feat=False, no_attn=False, no_logit=False, have_targets=True, mse_fn=None, kd_fn=None, learned=None):
    """compute_kd_losses (agent.py:546-719) for the mse/mse/kd loss selection of agent_base.py:155-175.

    proj: dict of the 5 projection heads (txt_emb_w, kdl_img_w, kdl_avg_img_w, global_cross_w,
    local_cross_w) of the model being *projected* (student for t2s; for s2t the real student's heads
    are applied to the target side, agent.py:571,606-607,647,664).  weights: 5 MKRW scalars (RW) or
    None with weight_mode=None (no adaptive weights: the two img emb terms are halved, :624-625).
    weight_mode='learned_weight' (:583-586, :616-619, :632-634, :680-686, :712-713): `learned` = the five raw scalars
    kdl_{txt,img,global,local,predict}_weight of `s_model` (the student for t2s; for s2t the TEACHER model, whose role is the
    learner, :553-557), each term scaled by softplus(raw); the two image-embedding terms ALSO halved, the image-attention term not.
    acc: dict accumulating the 10 entries (txt entries are assigned, the others added)."""
    if role == "s2t":
        loss_type = "mean"
    w = t_out["sample_weights"]
    hmin = min(s_out["txt_attns"].shape[1], t_out["txt_attns"].shape[1])
    if weight_mode == "learned_weight":
        import torch.nn.functional as F
        lw = [F.softplus(learned[n]) for n in ("kdl_txt_weight", "kdl_img_weight", "kdl_global_weight", "kdl_local_weight", "kdl_predict_weight")]
        k = lambda i: lw[i]
    else:
        k = lambda i: (weights[i] if weight_mode in ("RW", "grad") else 1.0)

    def sides(name, s_val, t_val):
        if role == "t2s":
            return proj[name](s_val), t_val.detach()
        return s_val, proj[name](t_val).detach()

    def feat(a, b):
        return 0 if no_feat else (mse_fn or mse_loss)(a, b, w, loss_type)

    def attn(a, b):
        return 0 if no_attn else (mse_fn or mse_loss)(a, b.detach(), w, loss_type)

    if t_step == 0 and "txt" in abilities:
        a, b = sides("txt_emb_w", s_out["txt_embeds"], t_out["txt_embeds"])
        acc["txt_emb_loss"] = feat(a, b) * k(0)
        acc["txt_attn_loss"] = attn(s_out["txt_attns"][:, :hmin], t_out["txt_attns"][:, :hmin]) * k(0)
    if "img" in abilities:
        a, b = sides("kdl_img_w", s_out["pano_embeds"], t_out["pano_embeds"])
        a2, b2 = sides("kdl_avg_img_w", s_out["pano_fused_embeds"], t_out["pano_fused_embeds"])
        half = 1.0 if weight_mode in ("RW", "grad") else 0.5
        acc["img_emb_loss"] = acc["img_emb_loss"] + feat(a, b) * k(1) * half
        acc["avg_img_emb_loss"] = acc["avg_img_emb_loss"] + feat(a2, b2) * k(1) * half
        acc["img_attn_loss"] = acc["img_attn_loss"] + attn(s_out["img_attns"], t_out["img_attns"]) * k(1)
    sn, tn = s_out["nav_outs"], t_out["nav_outs"]
    if "global" in abilities:
        a, b = sides("global_cross_w", sn["gmap_embeds"], tn["gmap_embeds"])
        acc["global_emb_loss"] = acc["global_emb_loss"] + feat(a, b) * k(2)
        acc["global_attn_loss"] = acc["global_attn_loss"] + attn(sn["gmap_attns"][:, :hmin], tn["gmap_attns"][:, :hmin]) * k(2)
    if "local" in abilities:
        a, b = sides("local_cross_w", sn["vp_embeds"], tn["vp_embeds"])
        acc["local_emb_loss"] = acc["local_emb_loss"] + feat(a, b) * k(3)
        acc["local_attn_loss"] = acc["local_attn_loss"] + attn(sn["vp_attns"][:, :hmin], tn["vp_attns"][:, :hmin]) * k(3)
    if "action" in abilities:
        p = 0
        if not no_logit and have_targets:
            p = (kd_fn or kd_loss)(s_out["nav_logits"], t_out["nav_logits"].detach(), temperature, w, loss_type)
        acc["predict_loss"] = acc["predict_loss"] + p * k(4)
    return acc


def episode_loss(kdl_sum, ml_sum, batch_size, train_ml, kdl_alpha=0.5):
    """Loss assembly agent.py:1110-1123: total = a * (sum kdl / B) + (1-a) * (ml * train_ml / B)."""
    return kdl_alpha * (kdl_sum / batch_size) + (1 - kdl_alpha) * (ml_sum * train_ml / batch_size)


def mktd_weights(teacher_logits, targets, rate=0.7, ignore_index=-100):
    """MKTD sample weights agent.py:1013-1020: exp(-rate * CE_b(teacher)); ignored rows -> CE 0 -> 1."""
    ce = F.cross_entropy(teacher_logits, targets, reduction="none", ignore_index=ignore_index)
    return exponential_decay(ce, rate)


def pretrain_makd(s_bert, s, t, batch, task, cfg, rw=None):
    """In-model MAKD of the (withheld) pretraining model, defined here from the nav aggregation above
    with the pretrain-flavour primitives (mean reductions; pano terms fall back to unweighted because
    their leading dim is sum(T) != B, pretrain kd_loss.py:11-16).  DESIGN.md open choice O8."""
    kdl = cfg.kdl
    T = float(kdl["kd_temperature"])
    rw = torch.ones(5) if rw is None else rw
    w = None
    if kdl.get("teacher_sample_hard_mining", False) and task == "sap":
        w = mktd_weights(t["fused_logits"], batch["global_act_labels"], float(kdl["t_sample_preprocess_exp_decay"]))
    tasks, types = kdl["kdl_tasks"], kdl["kdl_task_types"]
    emb, att = "emb" in types, "attn" in types
    hmin = min(s["txt_attns"].shape[1], t["txt_attns"].shape[1])
    m = lambda a, b, ww: mse_loss(a, b.detach(), ww, flavour="pretrain")
    out = {}
    if "txt" in tasks:
        if emb:
            out["txt_emb_loss"] = rw[0] * m(s_bert.txt_emb_w(s["txt_embeds"]), t["txt_embeds"], w)
        if att:
            out["txt_attn_loss"] = rw[0] * m(s["txt_attns"][:, :hmin], t["txt_attns"][:, :hmin], w)
    if "img" in tasks:
        if emb:
            out["img_emb_loss"] = rw[1] * m(s_bert.kdl_img_w(s["pano_embeds"]), t["pano_embeds"], w)
            out["avg_img_emb_loss"] = rw[1] * m(s_bert.kdl_avg_img_w(s["pano_fused_embeds"]), t["pano_fused_embeds"], w)
        if att:
            out["img_attn_loss"] = rw[1] * m(s["img_attns"], t["img_attns"], w)
    if "global" in tasks and "gmap_embeds" in s:
        if emb:
            out["global_emb_loss"] = rw[2] * m(s_bert.global_cross_w(s["gmap_embeds"]), t["gmap_embeds"], w)
        if att:
            out["global_attn_loss"] = rw[2] * m(s["gmap_attns"][:, :hmin], t["gmap_attns"][:, :hmin], w)
    if "local" in tasks and "vp_embeds" in s:
        if emb:
            out["local_emb_loss"] = rw[3] * m(s_bert.local_cross_w(s["vp_embeds"]), t["vp_embeds"], w)
        if att:
            out["local_attn_loss"] = rw[3] * m(s["vp_attns"][:, :hmin], t["vp_attns"][:, :hmin], w)
    if "predict" in tasks and task == "sap":
        out["predict_loss"] = rw[4] * kd_loss(s["fused_logits"], t["fused_logits"].detach(), T, w, flavour="pretrain")
    return out
