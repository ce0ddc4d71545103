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
