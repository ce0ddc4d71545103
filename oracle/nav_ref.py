"""ORACLE (test infrastructure): CPU restatement of the navigation-time model `VLNBert.forward(mode, inputs)`
(withheld by the reference; call sites map_nav_src/r2r/agent.py:796,:885,:964; input dicts :83-90,:167-173,
:245-251,:322-328,:936-944).  Built from the same blocks as oracle/model_ref.py; parity unpinned for the
forward as a whole (no reference implementation exists), pinned at block level as described there."""
import torch
import torch.nn as nn

from types import SimpleNamespace

from .causal_ref import RefCausalBlock, cat_instr_dict
from .model_ref import ClsPrediction, RefMagicBert, RefPretrainModel

CAUSAL_FLAGS = {"back_txt": "do_back_txt", "back_img": "do_back_img", "front_txt": "do_front_txt", "front_vp": "do_front_img",
                "front_gmap": "do_front_his"}


class _Inner(RefMagicBert):
    def __init__(self, cfg):
        super().__init__(cfg)
        H, eps = cfg.hidden_size, cfg.layer_norm_eps
        self.global_sap_head = ClsPrediction(H, eps=eps)
        self.local_sap_head = ClsPrediction(H, eps=eps)
        self.sap_fuse_linear = ClsPrediction(H, 2 * H, eps=eps)
        for k in ("txt", "img", "local", "global", "predict"):
            setattr(self, f"kdl_{k}_weight", nn.Parameter(torch.zeros(1)))
        on = [n for n, f in CAUSAL_FLAGS.items() if getattr(cfg, f, False)]
        if on:           # back-door / front-door blocks (f-4); dictionary width: CLIP width for the image dictionary, H otherwise
            self.causal = nn.ModuleDict({n: RefCausalBlock(cfg, n, getattr(cfg, f"{n}_dict_size", None) or
                                                           (cfg.image_feat_size if n == "back_img" else H)) for n in on})


def nav_fuse(gl, ll, b):
    fused = gl.clone()
    add = torch.zeros_like(fused)
    add[:, 0] = ll[:, 0]
    for i in range(gl.shape[0]):
        vm = b["gmap_visited_masks"][i]
        visited = set(vp for j, vp in enumerate(b["gmap_vpids"][i]) if vp is not None and vm[j])
        tmp, bw = {}, 0
        for j, c in enumerate(b["vp_cand_vpids"][i]):
            if c is None or j == 0:
                continue
            if c in visited:
                bw = bw + ll[i, j]
            else:
                tmp[c] = ll[i, j]
        for j, vp in enumerate(b["gmap_vpids"][i]):
            if j > 0 and vp is not None and vp not in visited:
                add[i, j] = tmp[vp] if vp in tmp else bw
    return fused + add


class RefVLNBert(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.vln_bert = _Inner(cfg)
        std = cfg.initializer_range
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                m.weight.data.normal_(0, std)
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()

    def forward(self, mode, b):
        m = self.vln_bert
        cz = getattr(m, "causal", {})
        if mode in ("language", "instr_zdict_update"):
            x, p = m.text(b["txt_ids"], b["txt_masks"]) if mode == "language" else m.text(b["z_txt"], b["z_txt_mask"])
            if "back_txt" in cz and b.get("instr_z_direction_features") is not None:
                x = cz["back_txt"](x, *cat_instr_dict(b))
            if "front_txt" in cz and b.get("front_txt_feats") is not None:
                x = cz["front_txt"](x, b["front_txt_feats"])
            return x, p
        if mode == "panorama":
            x, masks, fused, p = m.panorama(b["view_img_fts"], b["loc_fts"], b["nav_types"], b["view_lens"])
            if "back_img" in cz and b.get("z_img_features") is not None:
                x = cz["back_img"](x, b["z_img_features"], b["z_img_pzs"])
            return x, masks, fused, p
        if mode == "navigation":
            gimg, vimg = b["gmap_img_embeds"], b["vp_img_embeds"]
            if "front_gmap" in cz and b.get("front_gmap_feats") is not None:
                gimg = cz["front_gmap"](gimg, b["front_gmap_feats"])
            if "front_vp" in cz and b.get("front_vp_feats") is not None:
                vimg = cz["front_vp"](vimg, b["front_vp_feats"])
            b = dict(b, gmap_img_embeds=gimg, vp_img_embeds=vimg)
            gin = m.global_input(b["gmap_img_embeds"], b["gmap_step_ids"], b["gmap_pos_fts"])
            g, ga = m.global_encode(gin, b["gmap_masks"], b["gmap_pair_dists"], b["txt_embeds"], b["txt_masks"])
            vin = m.local_input(b["vp_img_embeds"], b["vp_pos_fts"])
            v, va = m.local_encode(vin, b["vp_masks"], b["txt_embeds"], b["txt_masks"])
            fw = torch.sigmoid(m.sap_fuse_linear(torch.cat([g[:, 0], v[:, 0]], 1))) if self.cfg.glocal_fuse else 0.5
            gl = (m.global_sap_head(g).squeeze(2) * fw).masked_fill(b["gmap_visited_masks"], -float("inf")).masked_fill(~b["gmap_masks"], -float("inf"))
            ll = (m.local_sap_head(v).squeeze(2) * (1 - fw)).masked_fill(~b["vp_nav_masks"], -float("inf"))
            return dict(gmap_embeds=g, vp_embeds=v, gmap_attns=ga, vp_attns=va, cls_embeds=g[:, 0] + v[:, 0],
                        global_logits=gl, local_logits=ll, fused_logits=nav_fuse(gl, ll, b))
        if mode == "extract_cfp_features":        # agent.py:1535-1541: first tokens of the whole-trajectory forward (cfp_collate batch)
            o = RefPretrainModel.trunk(SimpleNamespace(bert=m), b)
            return dict(txt_outputs=o["txt_embeds"][:, 0], vp_outputs=o["vp_embeds"][:, 0], gmap_outputs=o["gmap_embeds"][:, 0])
        raise NotImplementedError(mode)
