"""ORACLE (test infrastructure): CPU restatement of the navigation-time model `VLNBert.forward(mode, inputs)`
(withheld by the reference; call sites map_nav_src/r2r/agent.py:796,:885,:964; input dicts :83-90,:167-173,
:245-251,:322-328,:936-944).  Built from the same blocks as oracle/model_ref.py; parity unpinned for the
forward as a whole (no reference implementation exists), pinned at block level as described there."""
import torch
import torch.nn as nn

from types import SimpleNamespace

from .model_ref import ClsPrediction, RefMagicBert, RefPretrainModel


class _Inner(RefMagicBert):
    def __init__(self, cfg):
        super().__init__(cfg)
        H, eps = cfg.hidden_size, cfg.layer_norm_eps
        self.global_sap_head = ClsPrediction(H, eps=eps)
        self.local_sap_head = ClsPrediction(H, eps=eps)
        self.sap_fuse_linear = ClsPrediction(H, 2 * H, eps=eps)
        for k in ("txt", "img", "local", "global", "predict"):
            setattr(self, f"kdl_{k}_weight", nn.Parameter(torch.zeros(1)))


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
        if mode == "language":
            return m.text(b["txt_ids"], b["txt_masks"])
        if mode == "panorama":
            return m.panorama(b["view_img_fts"], b["loc_fts"], b["nav_types"], b["view_lens"])
        if mode == "navigation":
            gin = m.global_input(b["gmap_img_embeds"], b["gmap_step_ids"], b["gmap_pos_fts"])
            g, ga = m.global_encode(gin, b["gmap_masks"], b["gmap_pair_dists"], b["txt_embeds"], b["txt_masks"])
            vin = m.local_input(b["vp_img_embeds"], b["vp_pos_fts"])
            v, va = m.local_encode(vin, b["vp_masks"], b["txt_embeds"], b["txt_masks"])
            fw = torch.sigmoid(m.sap_fuse_linear(torch.cat([g[:, 0], v[:, 0]], 1))) if self.cfg.glocal_fuse else 0.5
            gl = (m.global_sap_head(g).squeeze(2) * fw).masked_fill(b["gmap_visited_masks"], -float("inf")).masked_fill(~b["gmap_masks"], -float("inf"))
            ll = (m.local_sap_head(v).squeeze(2) * (1 - fw)).masked_fill(~b["vp_nav_masks"], -float("inf"))
            return dict(gmap_embeds=g, vp_embeds=v, gmap_attns=ga, vp_attns=va, cls_embeds=g[:, 0] + v[:, 0],
                        global_logits=gl, local_logits=ll, fused_logits=nav_fuse(gl, ll, b))
        if mode == "instr_zdict_update":          # agent.py:1231-1233 (dictionaries off): per-token instruction embeddings
            return m.text(b["z_txt"], b["z_txt_mask"])
        if mode == "extract_cfp_features":        # agent.py:1535-1541: first tokens of the whole-trajectory forward (cfp_collate batch)
            o = RefPretrainModel.trunk(SimpleNamespace(bert=m), b)
            return dict(txt_outputs=o["txt_embeds"][:, 0], vp_outputs=o["vp_embeds"][:, 0], gmap_outputs=o["gmap_embeds"][:, 0])
        raise NotImplementedError(mode)
