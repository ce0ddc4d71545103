"""ORACLE (test infrastructure): CPU restatement of the back-door / front-door adjustment blocks of the navigation model
(SURVEY section 8 f-4).  The reference pins only their inputs (map_nav_src/r2r/agent.py:76-89 language, :162-172 panorama,
:942-944 navigation; switches map_nav_src/r2r/parser.py:129-142) -- the model source is withheld -- so the arithmetic
is this build's restatement of the published adjustments ([LINEAGE] GOAT): PARITY UNPINNED, open choices O14-O16 of DESIGN.md.
Plain torch, per-head loops, no fused ops: the checker for vln-magic_amd/host/causal.py."""
import math

import torch
import torch.nn as nn

HD = 64


class RefCausalBlock(nn.Module):
    def __init__(self, cfg, name, Dz):
        super().__init__()
        H = cfg.hidden_size
        self.name, self.H, self.nh = name, H, H // HD
        self.kind = "back" if name.startswith("back") else "front"
        self.btype = getattr(cfg, "do_back_txt_type", "type_2") if name == "back_txt" else \
            getattr(cfg, "do_back_imgobj_type", getattr(cfg, "do_back_img_type", "type_1")) if name == "back_img" else "type_2"
        self.door = getattr(cfg, "do_add_method", "add") == "door"
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(Dz, H), nn.Linear(Dz, H)
        self.output = nn.Module()
        self.output.dense = nn.Linear(H, H)
        self.output.LayerNorm = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
        if self.door:
            self.gate_x, self.gate_e = nn.Linear(H, 1), nn.Linear(H, 1)

    def forward(self, x, z, pz=None):
        B, N, H = x.shape
        z0 = (z[0] if z.dim() == 3 else z).to(x.dtype)
        k, v = self.key(z0), self.value(z0)                          # [Nz, H]
        if pz is not None:
            v = v * (pz[0] if pz.dim() == 3 else pz).reshape(-1, 1).to(x.dtype)     # [softmax . P(z)] z  ==  softmax . (P(z) z)
        if self.kind == "back" and self.btype == "type_1":
            e = v.sum(0).expand(B, N, H)
        else:
            q = self.query(x)
            heads = []
            for h in range(self.nh):
                sl = slice(h * HD, (h + 1) * HD)
                s = q[..., sl] @ k[:, sl].t() / math.sqrt(HD)         # [B, N, Nz]
                heads.append(torch.softmax(s, -1) @ v[:, sl])
            e = torch.cat(heads, -1)
        e = self.output.dense(e)
        if self.door:
            e = torch.sigmoid(self.gate_x(x) + self.gate_e(e)) * e
        return self.output.LayerNorm(x + e)


def cat_instr_dict(b):
    """direction and landmark entries form ONE instruction dictionary (direction rows first), priors concatenated"""
    z = torch.cat([b["instr_z_direction_features"], b["instr_z_landmark_features"]], 1)
    pz = torch.cat([b["instr_z_direction_pzs"], b["instr_z_landmark_pzs"]], 1)
    return z, pz
