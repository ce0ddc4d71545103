"""MAKD aggregation of one navigator step -- `GMapNavAgent.compute_kd_losses` (map_nav_src/r2r/agent.py:546-719) for the loss
selection the shipped scripts use (feature/attention terms = mse_loss, logits = kd_loss; agent_base.py:155-175) -- over the
fused HIP primitives of host/kd_loss.py.  Five meta-abilities x {embedding, attention} (+ action logits):

  txt (weight 0, step 0 only :562)   img (1; panorama tokens + fused panorama :596-640)   global (2 :641-658)
  local (3 :659-676)                 action (4 :677-716)

role 't2s': the student's tensors go through its projection heads (`txt_emb_w`, `kdl_img_w`, `kdl_avg_img_w`,
`global_cross_w`, `local_cross_w`) to the teacher's width and the teacher side is detached, reduction 'sum';
role 's2t' (ICoD, :555-558): the roles swap, the heads are applied to the target side, reduction 'mean'.
Attention maps are compared on the first min(h_s, h_t) heads (:560).  Ability weighting (args.kdl_adaptive_ability_weight[_type]):
'RW' / 'grad' = the step's five MKRW scalars (`weights`); 'learned_weight' = softplus of the learner model's five raw scalars
`kdl_{txt,img,global,local,predict}_weight` (`learned`; :583-586, :616-619, :632-634, :680-686, :712-713 -- the two image-embedding terms
are halved on top, the image-attention term is not); neither = unweighted with the two image-embedding terms halved (:624-625).
The unit tests check this function against the reference's own method (tests/golden/makd_agent.pt, minted by running it).
"""
import torch.nn.functional as F

from . import kd_loss as K

LOSS_KEYS = ("txt_emb_loss", "txt_attn_loss", "img_emb_loss", "avg_img_emb_loss", "img_attn_loss",
             "global_emb_loss", "global_attn_loss", "local_emb_loss", "local_attn_loss", "predict_loss")


LEARNED_NAMES = ("kdl_txt_weight", "kdl_img_weight", "kdl_global_weight", "kdl_local_weight", "kdl_predict_weight")   # ability order 0..4


def compute_kd_losses(t, s_out, t_out, heads, acc, *, role="t2s", temperature=2.0, weights=None, have_targets=True,
                      abilities=("txt", "img", "global", "local", "action"), feat=True, attn=True, logit=True, learned=None,
                      loss_type="sum"):
    """acc: dict of running sums (updated and returned).  weights: the 5 MKRW scalars of this step (device tensor or
    floats) or None (no ability weighting: the two image-embedding terms are halved, :624-625).  learned: the model whose
    `kdl_*_weight` parameters weight the abilities ('learned_weight': `s_model` of agent.py:553-557 -- the student's inner model for
    't2s', the TEACHER's for 's2t'), or a mapping name -> tensor; the softplus runs in torch so the raw scalars receive their gradient."""
    loss_type = loss_type if role == "t2s" else "mean"          # args.kd_loss_type for 't2s' (:553-554); 's2t' is always 'mean' (:558)
    w = t_out.get("sample_weights")
    hmin = min(s_out["txt_attns"].shape[1], t_out["txt_attns"].shape[1])
    if learned is not None:
        if weights is not None:
            raise ValueError("compute_kd_losses: pass either the MKRW `weights` or the `learned` ability weights, not both")
        raw = (lambda n: learned[n]) if isinstance(learned, dict) else (lambda n: getattr(learned, n))
        lw = [F.softplus(raw(n)) for n in LEARNED_NAMES]
        k = lambda i: lw[i]
    else:
        k = (lambda i: weights[i]) if weights is not None else (lambda i: 1.0)

    def pair(name, a, b):
        if role == "t2s":
            return heads[name](a), b.detach()
        return a, heads[name](b).detach()

    def f_term(name, a, b):
        if not feat:
            return 0.0
        x, y = pair(name, a, b)
        return K.mse_loss(x, y, w, loss_type)

    def a_term(a, b):
        return K.mse_loss(a, b.detach(), w, loss_type) if attn else 0.0

    def add(key, v):
        acc[key] = acc.get(key, 0.0) + v

    if t == 0 and "txt" in abilities:
        acc["txt_emb_loss"] = f_term("txt_emb_w", s_out["txt_embeds"], t_out["txt_embeds"]) * k(0)
        acc["txt_attn_loss"] = a_term(s_out["txt_attns"][:, :hmin], t_out["txt_attns"][:, :hmin]) * k(0)
    if "img" in abilities:
        half = 1.0 if weights is not None else 0.5
        add("img_emb_loss", f_term("kdl_img_w", s_out["pano_embeds"], t_out["pano_embeds"]) * k(1) * half)
        add("avg_img_emb_loss", f_term("kdl_avg_img_w", s_out["pano_fused_embeds"], t_out["pano_fused_embeds"]) * k(1) * half)
        add("img_attn_loss", a_term(s_out["img_attns"], t_out["img_attns"]) * k(1))
    sn, tn = s_out["nav_outs"], t_out["nav_outs"]
    if "global" in abilities:
        add("global_emb_loss", f_term("global_cross_w", sn["gmap_embeds"], tn["gmap_embeds"]) * k(2))
        add("global_attn_loss", a_term(sn["gmap_attns"][:, :hmin], tn["gmap_attns"][:, :hmin]) * k(2))
    if "local" in abilities:
        add("local_emb_loss", f_term("local_cross_w", sn["vp_embeds"], tn["vp_embeds"]) * k(3))
        add("local_attn_loss", a_term(sn["vp_attns"][:, :hmin], tn["vp_attns"][:, :hmin]) * k(3))
    if "action" in abilities:
        p = 0.0
        if logit and have_targets:
            p = K.kd_loss(s_out["nav_logits"], t_out["nav_logits"].detach(), temperature, t_sample_weights=w, loss_type=loss_type)
        add("predict_loss", p * k(4))
    return acc


VEC_KEYS = LOSS_KEYS            # order of the running vector of compute_kd_losses_fused


def compute_kd_losses_fused(t, s_out, t_out, heads, acc, *, role="t2s", temperature=2.0, weights=None):
    """compute_kd_losses for the configuration the rollout loop runs (all five abilities, feature + attention + logit terms, MKRW `weights`
    or none) with the nine mse terms of the step in ONE launch and ONE autograd node (kd_loss._MseMulti) and the running sums as one vector:
    acc["_vec"] [10] in VEC_KEYS order (`kd_terms(acc)` turns it into the dict compute_kd_losses keeps).  Same values and gradients
    (tests/test_rollout_gpu.py runs both against the oracle)."""
    import torch
    loss_type = "sum" if role == "t2s" else "mean"
    w = t_out.get("sample_weights")
    hmin = min(s_out["txt_attns"].shape[1], t_out["txt_attns"].shape[1])
    half = 1.0 if weights is not None else 0.5
    sn, tn = s_out["nav_outs"], t_out["nav_outs"]
    feats, attns = [], []                        # (slot, head name, learner tensor, target tensor, ability, host coefficient) / (slot, a, b, ability)

    def feat(slot, name, a, b, ab, c=1.0):
        feats.append((slot, name, a, b, ab, c))

    def attn(slot, a, b, ab):
        attns.append((slot, a, b.detach(), ab, 1.0))

    if t == 0:
        feat(0, "txt_emb_w", s_out["txt_embeds"], t_out["txt_embeds"], 0)
        attn(1, s_out["txt_attns"][:, :hmin], t_out["txt_attns"][:, :hmin], 0)
    feat(2, "kdl_img_w", s_out["pano_embeds"], t_out["pano_embeds"], 1, half)
    feat(3, "kdl_avg_img_w", s_out["pano_fused_embeds"], t_out["pano_fused_embeds"], 1, half)
    attn(4, s_out["img_attns"], t_out["img_attns"], 1)
    feat(5, "global_cross_w", sn["gmap_embeds"], tn["gmap_embeds"], 2)
    attn(6, sn["gmap_attns"][:, :hmin], tn["gmap_attns"][:, :hmin], 2)
    feat(7, "local_cross_w", sn["vp_embeds"], tn["vp_embeds"], 3)
    attn(8, sn["vp_attns"][:, :hmin], tn["vp_attns"][:, :hmin], 3)
    # the projection heads of the step in one autograd node, their GEMMs as grouped launches (model_nav.hip_linear_multi); 't2s' projects the
    # learner side, 's2t' the (detached) target side
    from .model_nav import hip_linear_multi
    mods = [heads[f[1]] for f in feats]
    proj = hip_linear_multi(mods, [f[2] if role == "t2s" else f[3].detach() for f in feats])
    terms = []                                   # (slot, x, y, ability, host coefficient)
    for f, y in zip(feats, proj):
        terms.append((f[0], y, f[3].detach(), f[4], f[5]) if role == "t2s" else (f[0], f[2], y.detach(), f[4], f[5]))
    terms += attns
    terms.sort(key=lambda tm: tm[0])
    on_dev = weights is not None and torch.is_tensor(weights) and weights.is_cuda
    meta, xy = [], []
    for slot, x, y, ab, c in terms:
        norm = 1.0 if loss_type == "sum" else 1.0 / x.numel()
        if weights is not None and not on_dev:
            c = c * float(weights[ab])           # (host-side MKRW scalars)
        meta.append(dict(norm=norm, coef=c, coef_dev=weights[ab:ab + 1] if on_dev else None, w=w))
        xy += [x, y]
    vals = K._MseMulti.apply(meta, *xy)
    vec = acc.get("_vec")
    if vec is None:
        vec = torch.zeros(len(VEC_KEYS), dtype=torch.float32, device=vals.device)
    idx = _slot_index(tuple(tm[0] for tm in terms), vals.device)
    p = K.kd_loss(s_out["nav_logits"], t_out["nav_logits"].detach(), temperature, t_sample_weights=w, loss_type=loss_type)
    if weights is not None:
        p = p * weights[4]
    acc["_vec"] = vec.index_add(0, idx, torch.cat([vals, p.reshape(1)]))
    return acc


_SLOT_IDX = {}


def _slot_index(slots, dev):
    key = (slots, str(dev))
    v = _SLOT_IDX.get(key)
    if v is None:
        import torch
        v = _SLOT_IDX[key] = torch.tensor(list(slots) + [9], dtype=torch.int64, device=dev)
    return v


def kd_terms(acc):
    """the per-key running sums of either accumulator form as a dict (compute_kd_losses' own form passes through)"""
    vec = acc.get("_vec") if isinstance(acc, dict) else None
    if vec is None:
        return acc
    return {k: vec[i] for i, k in enumerate(VEC_KEYS)}
