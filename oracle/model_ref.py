"""ORACLE (test infrastructure, never shipped, never measured except as bench.py's cpu_baseline).

CPU / fp32 / plain-PyTorch restatement of the MAGIC cross-modal transformer whose source the
reference withholds (/root/reference/readme.md:75; imports at pretrain_src/train_r2r_magic.py:40 and
map_nav_src/r2r/agent.py:29-31).  **Parity unpinned for the model forward as a whole**: no reference
implementation exists to diff against.  What IS pinned and how:
  * call/return contract    -> pretrain_src/train_r2r_magic.py:440-587 (validate_*), SURVEY App. A
  * parameter names         -> the METER remap train_r2r_magic.py:183-209
  * block arithmetic        -> HF BertLayer (tests/test_oracle_blocks.py cross-checks RefSelfLayer
                               against transformers' BertLayer with identical parameter names)
  * distillation arithmetic -> oracle/makd_ref.py, pinned by fixtures minted from the reference's own
                               kd_loss.py / agent.compute_kd_losses (tests/golden/*)
Open choices fixed here and mirrored by the HIP engine are listed in DESIGN.md §3.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

NEG = -10000.0
HEAD_DIM = 64

# Dropout hook (tests only): callable(site_name, tensor) -> tensor, or None (= eval mode / dropout 0, the default).
# Site names are the qualified module names of HF's BertSelfAttention.dropout / BertSelfOutput.dropout /
# BertOutput.dropout / BertEmbeddings.dropout / ImageEmbeddings.dropout; the HIP engine derives its counter-based
# masks from the same names, so a test can export the engine's masks and replay them here exactly.
DROPOUT = None


def _drop(owner, suffix, x):
    if DROPOUT is None:
        return x
    return DROPOUT(getattr(owner, "_qual", "") + suffix, x)


def name_modules(root):
    """record every sub-module's qualified name (dropout site names)"""
    for n, m in root.named_modules():
        m._qual = n


def _ln(h, eps):
    return nn.LayerNorm(h, eps=eps)


class RefAttention(nn.Module):
    """HF BertAttention naming: self.{query,key,value}, output.{dense,LayerNorm}."""

    def __init__(self, cfg):
        super().__init__()
        H = cfg.hidden_size
        self.nh = cfg.num_attention_heads
        self.self = nn.Module()
        self.self.query, self.self.key, self.self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)
        self.output = nn.Module()
        self.output.dense = nn.Linear(H, H)
        self.output.LayerNorm = _ln(H, cfg.layer_norm_eps)

    def forward(self, x, ctx, bias):
        """x [B,Nq,H], ctx [B,Nk,H], bias broadcastable to [B,h,Nq,Nk] (additive). Returns (out, probs)."""
        B, Nq, H = x.shape
        Nk = ctx.shape[1]
        d = H // self.nh
        q = self.self.query(x).view(B, Nq, self.nh, d).transpose(1, 2)
        k = self.self.key(ctx).view(B, Nk, self.nh, d).transpose(1, 2)
        v = self.self.value(ctx).view(B, Nk, self.nh, d).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / math.sqrt(d) + bias
        p = _drop(self, ".self.dropout", torch.softmax(s, dim=-1))    # HF returns the probabilities AFTER dropout
        c = (p @ v).transpose(1, 2).reshape(B, Nq, H)
        out = self.output.LayerNorm(x + _drop(self, ".output.dropout", self.output.dense(c)))
        return out, p


class RefFFN(nn.Module):
    def __init__(self, cfg, owner):
        H, I = cfg.hidden_size, cfg.intermediate_size
        owner.intermediate = nn.Module()
        owner.intermediate.dense = nn.Linear(H, I)
        owner.output = nn.Module()
        owner.output.dense = nn.Linear(I, H)
        owner.output.LayerNorm = _ln(H, cfg.layer_norm_eps)


def _ffn(layer, a):
    f = _drop(layer, ".output.dropout", layer.output.dense(F.gelu(layer.intermediate.dense(a))))
    return layer.output.LayerNorm(a + f)


class RefSelfLayer(nn.Module):
    """Post-LN BERT block."""

    def __init__(self, cfg):
        super().__init__()
        self.attention = RefAttention(cfg)
        RefFFN(cfg, self)

    def forward(self, x, bias):
        a, p = self.attention(x, x, bias)
        return _ffn(self, a), p


class RefCrossLayer(nn.Module):
    """METER BertCrossLayer: self-attn -> cross-attn to the other modality -> FFN."""

    def __init__(self, cfg):
        super().__init__()
        self.attention = RefAttention(cfg)
        self.crossattention = RefAttention(cfg)
        RefFFN(cfg, self)

    def forward(self, x, self_bias, ctx, ctx_bias):
        s, _ = self.attention(x, x, self_bias)
        c, p = self.crossattention(s, ctx, ctx_bias)
        return _ffn(self, c), p


def key_bias(mask):
    """mask [B,N] bool (True = valid) -> additive [B,1,1,N]."""
    return ((~mask).float() * NEG)[:, None, None, :]


def seq_mask(lens, n):
    return torch.arange(n, device=lens.device)[None, :] < lens[:, None]


class ClsPrediction(nn.Module):
    def __init__(self, H, inp=None, eps=1e-12):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(inp or H, H), nn.ReLU(), _ln(H, eps), nn.Linear(H, 1))

    def forward(self, x):
        return self.net(x)


class RefMagicBert(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H, eps = cfg.hidden_size, cfg.layer_norm_eps
        self.cfg = cfg
        e = self.embeddings = nn.Module()
        e.word_embeddings = nn.Embedding(cfg.vocab_size, H)
        e.position_embeddings = nn.Embedding(cfg.max_position_embeddings, H)
        e.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, H)
        e.LayerNorm = _ln(H, eps)
        self.lang_encoder = nn.Module()
        self.lang_encoder.layer = nn.ModuleList([RefSelfLayer(cfg) for _ in range(cfg.num_l_layers)])
        ie = self.img_embeddings = nn.Module()
        ie.img_linear = nn.Linear(cfg.image_feat_size, H)
        ie.img_layer_norm = _ln(H, eps)
        ie.loc_linear = nn.Linear(cfg.angle_feat_size + 3, H)
        ie.loc_layer_norm = _ln(H, eps)
        ie.nav_type_embedding = nn.Embedding(3, H)
        ie.layer_norm = _ln(H, eps)
        ie.pano_encoder = nn.Module()
        ie.pano_encoder.layer = nn.ModuleList([RefSelfLayer(cfg) for _ in range(cfg.num_pano_layers)])
        ie.pano_fuse_linear = nn.Linear(H, 1)
        g = self.global_encoder = nn.Module()
        g.gmap_pos_embeddings = nn.Sequential(nn.Linear(cfg.angle_feat_size + 3, H), _ln(H, eps))
        g.gmap_step_embeddings = nn.Embedding(cfg.max_action_steps, H)
        g.sprel_linear = nn.Linear(1, 1)
        g.encoder = nn.Module()
        g.encoder.crossattention = nn.ModuleList([RefCrossLayer(cfg) for _ in range(cfg.num_x_layers)])
        l = self.local_encoder = nn.Module()
        l.vp_pos_embeddings = nn.Sequential(nn.Linear(cfg.angle_feat_size * 2 + 6, H), _ln(H, eps))
        l.encoder = nn.Module()
        l.encoder.crossattention = nn.ModuleList([RefCrossLayer(cfg) for _ in range(cfg.num_x_layers)])
        if getattr(cfg, "teacher_hidden_size", None):
            Ht = cfg.teacher_hidden_size
            for n in ("txt_emb_w", "kdl_img_w", "kdl_avg_img_w", "global_cross_w", "local_cross_w"):
                setattr(self, n, nn.Linear(H, Ht))     # names: map_nav_src/r2r/agent_base.py:330

    # ---- encoders -------------------------------------------------------------------------
    def text(self, txt_ids, txt_masks):
        e = self.embeddings
        L = txt_ids.shape[1]
        pos = torch.arange(L, device=txt_ids.device) + 2        # RoBERTa offset (padding_idx 1)
        x = e.word_embeddings(txt_ids) + e.position_embeddings(pos)[None] + e.token_type_embeddings.weight[0]
        x = _drop(e, ".dropout", e.LayerNorm(x))
        kb = key_bias(txt_masks)
        p = None
        for lyr in self.lang_encoder.layer:
            x, p = lyr(x, kb)
        return x, p

    def panorama(self, view_fts, loc_fts, nav_types, view_lens):
        ie = self.img_embeddings
        x = ie.img_layer_norm(ie.img_linear(view_fts)) + ie.loc_layer_norm(ie.loc_linear(loc_fts)) \
            + ie.nav_type_embedding(nav_types) + self.embeddings.token_type_embeddings.weight[0]
        x = _drop(ie, ".dropout", ie.layer_norm(x))
        masks = seq_mask(view_lens, x.shape[1])
        kb = key_bias(masks)
        p = None
        for lyr in ie.pano_encoder.layer:
            x, p = lyr(x, kb)
        if self.cfg.adaptive_pano_fusion:
            sc = ie.pano_fuse_linear(x).squeeze(-1) + (~masks).float() * NEG
            fused = (torch.softmax(sc, -1)[..., None] * x).sum(1)
        else:
            m = masks.float()[..., None]
            fused = (x * m).sum(1) / m.sum(1)
        return x, masks, fused, p.mean(1)

    def global_input(self, gmap_img_embeds, gmap_step_ids, gmap_pos_fts):
        g = self.global_encoder
        return gmap_img_embeds + g.gmap_step_embeddings(gmap_step_ids) + g.gmap_pos_embeddings(gmap_pos_fts)

    def global_encode(self, x, gmap_masks, gmap_pair_dists, txt_embeds, txt_masks):
        g = self.global_encoder
        sb = key_bias(gmap_masks)
        if self.cfg.graph_sprels:
            sb = sb + g.sprel_linear(gmap_pair_dists.unsqueeze(3)).squeeze(3).unsqueeze(1)
        cb = key_bias(txt_masks)
        p = None
        for lyr in g.encoder.crossattention:
            x, p = lyr(x, sb, txt_embeds, cb)
        return x, p

    def local_input(self, vp_img_embeds, vp_pos_fts):
        return vp_img_embeds + self.local_encoder.vp_pos_embeddings(vp_pos_fts)

    def local_encode(self, x, vp_masks, txt_embeds, txt_masks):
        sb, cb = key_bias(vp_masks), key_bias(txt_masks)
        p = None
        for lyr in self.local_encoder.encoder.crossattention:
            x, p = lyr(x, sb, txt_embeds, cb)
        return x, p

    def lang2visn(self, txt_embeds, txt_masks, gmap_in, gmap_masks):
        """MLM path: the text attends to the map through the global cross layers with roles swapped
        (use_lang2visn_attn, r2r_magic_model_config.json:27; DESIGN.md open choice O7)."""
        sb, cb = key_bias(txt_masks), key_bias(gmap_masks)
        x, p = txt_embeds, None
        for lyr in self.global_encoder.encoder.crossattention:
            x, p = lyr(x, sb, gmap_in, cb)
        return x, p


def aggregate_gmap(pano_embeds, pano_fused, batch):
    """Map-node image features from per-step panorama embeddings, by viewpoint id
    ([LINEAGE] DUET _aggregate_gmap_features; node semantics agent.py:905-924: visited node <- fused
    panorama embedding, unvisited node <- mean of the candidate-view embeddings that saw it)."""
    B = len(batch["traj_step_lens"])
    K = batch["gmap_step_ids"].shape[1]
    H = pano_embeds.shape[-1]
    out = pano_embeds.new_zeros(B, K, H)
    row = 0
    rows = []
    for b in range(B):
        vis, unv = {}, {}
        for t in range(batch["traj_step_lens"][b]):
            vis[batch["traj_vpids"][b][t]] = pano_fused[row]
            for j, c in enumerate(batch["traj_cand_vpids"][b][t]):
                unv.setdefault(c, []).append(pano_embeds[row, j])
            row += 1
        for k, vp in enumerate(batch["gmap_vpids"][b]):
            if k == 0:
                continue
            rows.append((b, k, vis[vp] if vp in vis else torch.stack(unv[vp]).mean(0)))
    idx_b = torch.tensor([r[0] for r in rows])
    idx_k = torch.tensor([r[1] for r in rows])
    return out.index_put((idx_b, idx_k), torch.stack([r[2] for r in rows]))


def fuse_logits(global_logits, local_logits, batch):
    """Local->global logit fusion ([LINEAGE] DUET forward_sap)."""
    fused = global_logits.clone()
    add = torch.zeros_like(fused)
    add[:, 0] = local_logits[:, 0]
    B = fused.shape[0]
    for i in range(B):
        vm = batch["gmap_visited_masks"][i]
        visited = set(vp for j, vp in enumerate(batch["gmap_vpids"][i]) if vm[j])
        tmp, bw = {}, 0
        for j, c in enumerate(batch["traj_cand_vpids"][i][-1]):
            if c in visited:
                bw = bw + local_logits[i, j + 1]
            else:
                tmp[c] = local_logits[i, j + 1]
        for j, vp in enumerate(batch["gmap_vpids"][i]):
            if j > 0 and vp not in visited:
                add[i, j] = tmp[vp] if vp in tmp else bw
    return fused + add


class RefPretrainModel(nn.Module):
    """Restatement of GlocalTextPathCMTPreTraining.forward(batch, task, compute_loss)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        H, eps = cfg.hidden_size, cfg.layer_norm_eps
        self.bert = RefMagicBert(cfg)
        m = self.mlm_head = nn.Module()
        m.predictions = nn.Module()
        m.predictions.transform = nn.Module()
        m.predictions.transform.dense = nn.Linear(H, H)
        m.predictions.transform.LayerNorm = _ln(H, eps)
        m.predictions.bias = nn.Parameter(torch.zeros(cfg.vocab_size))   # decoder weight tied to word emb
        self.global_sap_head = ClsPrediction(H, eps=eps)
        self.local_sap_head = ClsPrediction(H, eps=eps)
        self.sap_fuse_linear = ClsPrediction(H, 2 * H, eps=eps)
        self.cfp_heads = nn.ModuleDict({k: nn.Linear(H, H) for k in ("gmap", "vp", "fused", "txt")})
        if "mrc" in (getattr(cfg, "pretrain_tasks", None) or ()):      # built only when the task is configured (train_r2r_magic.py:104-107)
            self.image_classifier = nn.Module()                            # RegionClassification(H, image_prob_size)
            self.image_classifier.net = nn.Sequential(nn.Linear(H, H), nn.ReLU(), _ln(H, eps), nn.Linear(H, cfg.image_prob_size))
        # back-door adjustment blocks (do_back_txt / do_back_img; inputs instr_z_* / img_z_* from the collates, tasks.py:156-164, :441-449):
        # the navigation oracle's blocks (oracle/causal_ref.py) at the same two places VLNBert applies them; PARITY UNPINNED like the model
        on = [n for n, f in (("back_txt", "do_back_txt"), ("back_img", "do_back_img")) if getattr(cfg, f, False)]
        if on:
            from .causal_ref import RefCausalBlock
            self.bert.causal = nn.ModuleDict({n: RefCausalBlock(cfg, n, getattr(cfg, f"{n}_dict_size", None) or
                                                                (cfg.image_feat_size if n == "back_img" else H)) for n in on})
        self.apply(self._init)
        name_modules(self)

    def _init(self, m):
        std = self.cfg.initializer_range
        if isinstance(m, (nn.Linear, nn.Embedding)):
            m.weight.data.normal_(0, std)
        if isinstance(m, nn.Linear) and m.bias is not None:
            m.bias.data.zero_()
        if isinstance(m, nn.LayerNorm):
            m.weight.data.fill_(1.0)
            m.bias.data.zero_()

    # ---- shared trunk ---------------------------------------------------------------------
    def trunk(self, batch, need_local=True, need_global=True):
        bert = self.bert
        o = {}
        txt_masks = seq_mask(batch["txt_lens"], batch["txt_ids"].shape[1])
        o["txt_masks"] = txt_masks
        o["txt_embeds"], o["txt_attns"] = bert.text(batch["txt_ids"], txt_masks)
        cz = getattr(bert, "causal", {})
        if "back_txt" in cz and batch.get("instr_z_direction_features") is not None:
            from .causal_ref import cat_instr_dict
            o["txt_embeds"] = cz["back_txt"](o["txt_embeds"], *cat_instr_dict(batch))
        pe, pm, pf, pa = bert.panorama(batch["traj_view_img_fts"], batch["traj_loc_fts"],
                                       batch["traj_nav_types"], batch["traj_vp_view_lens"])
        if "back_img" in cz and batch.get("img_z_features") is not None:      # view embeddings only: the fused embedding stays un-adjusted
            pe = cz["back_img"](pe, batch["img_z_features"], batch["img_z_pzs"])
        o["pano_embeds"], o["pano_fused_embeds"], o["img_attns"] = pe, pf, pa
        gmap_img = aggregate_gmap(pe, pf, batch)
        o["gmap_masks"] = seq_mask(batch["gmap_lens"], batch["gmap_step_ids"].shape[1])
        o["gmap_in"] = bert.global_input(gmap_img, batch["gmap_step_ids"], batch["gmap_pos_fts"])
        if need_global:
            o["gmap_embeds"], o["gmap_attns"] = bert.global_encode(
                o["gmap_in"], o["gmap_masks"], batch["gmap_pair_dists"], o["txt_embeds"], txt_masks)
        if need_local:
            last = torch.tensor(batch["traj_step_lens"]).cumsum(0) - 1
            B, H = len(last), pe.shape[-1]
            vp_lens = batch["traj_vp_view_lens"][last] + 1
            # [stop] + the current panorama's views, cut to the longest CURRENT panorama of the batch ([LINEAGE] DUET
            # LocalVPEncoder.vp_input_embedding `[:, :max_vp_len]`; = vp_pos_fts.shape[1], tasks.py:434-435)
            vp_img = torch.cat([pe.new_zeros(B, 1, H), pe[last]], 1)[:, :int(vp_lens.max())]
            o["vp_masks"] = seq_mask(vp_lens, vp_img.shape[1])
            o["vp_in"] = bert.local_input(vp_img, batch["vp_pos_fts"])
            o["vp_embeds"], o["vp_attns"] = bert.local_encode(o["vp_in"], o["vp_masks"], o["txt_embeds"], txt_masks)
            o["last_rows"] = last
        return o

    def sap_logits(self, o, batch):
        g0, v0 = o["gmap_embeds"][:, 0], o["vp_embeds"][:, 0]
        fw = torch.sigmoid(self.sap_fuse_linear(torch.cat([g0, v0], 1))) if self.cfg.glocal_fuse else 0.5
        gl = self.global_sap_head(o["gmap_embeds"]).squeeze(2) * fw
        gl = gl.masked_fill(batch["gmap_visited_masks"], -float("inf")).masked_fill(~o["gmap_masks"], -float("inf"))
        ll = self.local_sap_head(o["vp_embeds"]).squeeze(2) * (1 - fw)
        nav = batch["traj_nav_types"][o["last_rows"]] == 1
        vp_nav = torch.cat([torch.ones(len(nav), 1, dtype=torch.bool), nav], 1)[:, :ll.shape[1]]
        ll = ll.masked_fill(~vp_nav, -float("inf"))
        fl = fuse_logits(gl, ll, batch)
        return gl, ll, fl

    def forward(self, batch, task, compute_loss=True, teacher_outputs=None, rw=None):
        """rw: the 5 MKRW ability weights softmax(randn(5)/rw_temp)*5 (agent.py:866-871); None -> ones."""
        cfg = self.cfg
        if task == "mlm":
            o = self.trunk(batch, need_local=False, need_global=False)
            x, p = self.bert.lang2visn(o["txt_embeds"], o["txt_masks"], o["gmap_in"], o["gmap_masks"])
            o["gmap_embeds"], o["gmap_attns"] = x, p      # 'global' KD slot for MLM
            sel = batch["txt_labels"] != -1
            t = self.mlm_head.predictions.transform
            hm = t.LayerNorm(F.gelu(t.dense(x[sel])))
            logits = hm @ self.bert.embeddings.word_embeddings.weight.t() + self.mlm_head.predictions.bias
            o["predict"] = logits
            if not compute_loss:
                return {"predict": logits}
            sup = F.cross_entropy(logits, batch["txt_labels"][sel], reduction="mean")
        elif task == "sap":
            o = self.trunk(batch)
            gl, ll, fl = self.sap_logits(o, batch)
            o.update(global_logits=gl, local_logits=ll, fused_logits=fl)
            if not compute_loss:
                return dict(global_logits=gl, local_logits=ll, fused_logits=fl,
                            global_act_labels=batch["global_act_labels"],
                            local_act_labels=batch["local_act_labels"])
            ga, la = batch["global_act_labels"], batch["local_act_labels"]
            per = F.cross_entropy(gl, ga, reduction="none") + F.cross_entropy(ll, la, reduction="none", ignore_index=-100) \
                + F.cross_entropy(fl, ga, reduction="none")
            sup = per.mean()
        elif task == "mrc":
            # masked-region classification on the current viewpoint's views: local branch only (validate_mrc,
            # train_r2r_magic.py:476-500: returns (view_logits, view_targets, obj_logits, obj_targets); obj_prob_size is 0)
            o = self.trunk(batch, need_global=False)
            mm = batch["vp_view_mrc_masks"]
            x = o["vp_embeds"][:, 1:1 + mm.shape[1]][mm]          # [stop] token sits at position 0
            logits = self.image_classifier.net(x)
            tgt = batch["vp_view_probs"][mm].to(logits.dtype)
            o["predict"] = logits
            if not compute_loss:
                return logits, tgt, None, None
            sup = F.kl_div(F.log_softmax(logits, -1), tgt, reduction="none").sum(1).mean()
        elif task == "cfp":
            o = self.trunk(batch)
            h = self.cfp_heads
            g0, v0 = o["gmap_embeds"][:, 0], o["vp_embeds"][:, 0]
            outs = (h["gmap"](g0), h["vp"](v0), h["fused"](g0 + v0), h["txt"](o["txt_embeds"][:, 0]))
            o["cfp"] = outs
            if not compute_loss:
                return outs
            tgt = torch.arange(len(g0))
            sup = 0
            for a in outs[:3]:       # validate_cfp arithmetic, train_r2r_magic.py:548-560, mean over batch
                sim = a @ outs[3].t() / cfg.cfp_temperature
                sup = sup + (F.cross_entropy(sim, tgt, reduction="sum") + F.cross_entropy(sim.t(), tgt, reduction="sum")) / 2.0
            sup = sup / len(g0)
        else:
            raise ValueError(task)
        out = {"supervised_loss": sup, "outputs": o}
        if teacher_outputs is None:
            out["loss"] = sup
            return out
        from . import makd_ref
        kdl = makd_ref.pretrain_makd(self.bert, o, teacher_outputs, batch, task, cfg, rw)
        out["kdl_terms"] = kdl
        out["kdl_loss"] = sum(kdl.values())
        alpha = cfg.kdl["kd_alpha"]
        out["loss"] = alpha * out["kdl_loss"] + (1 - alpha) * sup
        return out
