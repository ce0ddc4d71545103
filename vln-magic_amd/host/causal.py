"""Causal-intervention blocks of the navigation model: back-door adjustment over the z-dictionaries and
front-door adjustment over the K-means feature dictionary.

What the reference pins (its model source is withheld, readme.md:75) is the INPUT side only:
  language    instr_z_direction_features / _pzs, instr_z_landmark_features / _pzs [B, Nz, Dz] / [B, Nz, 1] (one dictionary repeated
              over the batch, map_nav_src/r2r/agent.py:76-81) and front_txt_feats [B, Nf, D] (:1212-1227, :89);
  panorama    z_img_features [B, Nz, Dz], z_img_pzs [B, Nz, 1] (:162-164);
  navigation  front_txt_feats / front_vp_feats / front_gmap_feats (:942-944);
  switches    do_back_txt / do_back_img / do_back_txt_type {type_1: p_z, type_2: attention} / do_back_imgobj_type / do_add_method
              {add, door} / do_front_txt / do_front_img / do_front_his / front_n_clusters (map_nav_src/r2r/parser.py:129-142,
              scripts/run_r2r_kdl_valid.sh:59-69, pretrain_src/config/r2r_magic_model_config.json:60-67);
  data        the dictionaries' files (LoadZdict r2r/data_utils.py:45-120: TSV rows with base64 fp32 features and a prior pz;
              KMeansPicker utils/data.py:436-513: one random member per K-means cluster of the extract_cfp_features vectors)
              -- restated in host/zdict.py and pinned there.
The ARITHMETIC is this build's restatement of the published form of the two adjustments ([LINEAGE] GOAT, the authors' earlier
causal-learning navigator): parity unpinned, like the rest of the withheld model (open choices O14-O16).

  back-door   P(Y|do(X)) = sum_z P(Y|X,z) P(z)  ~  f(x, E_z[z]),  E_z[z] = sum_z [softmax_z(q(x) . k(z) / sqrt(d))] P(z) v(z)
              (type_2, attention) or sum_z P(z) v(z) (type_1, prior only);
  front-door  cross-sample attention of every token over the global dictionary of cluster representatives;
  both        one multi-head cross-attention sub-block with the dictionary as its context (heads of 64 like every other attention
              of the model), then  x' = LayerNorm(x + dropout(W_o e))  ('add')  or  LayerNorm(x + g * dropout(W_o e)),
              g = sigmoid(w_x . x + w_e . e + b) per token  ('door': a learned gate decides how much adjustment passes).

Device work: the three token-side products (query projection, attention over the dictionary, output projection) and the
add&norm are the engine's HIP kernels (GEMM, fused attention forward / backward, fused LayerNorm) behind small autograd
wrappers, so the blocks compose with the mode functions of model_nav.py; the dictionary-side projection ([Nz, Dz], batch
independent) is one HIP GEMM per call; the prior scaling and the optional gate are elementwise torch ops on [B, N, H].
"""
import math

import torch
import torch.nn as nn

from . import lanes
from . import ops as O
from .config import cfg_get
from .engine import HD

# block name -> (config switch, dictionary width attribute)
BLOCKS = {
    "back_txt": "do_back_txt", "back_img": "do_back_img",
    "front_txt": "do_front_txt", "front_vp": "do_front_img", "front_gmap": "do_front_his",
}


def dict_width(cfg, name):
    """width of the dictionary entries a block consumes: the CLIP image width for the room-type image dictionary
    (image_z_dict_clip_50.tsv, parser.py:236); the model's own hidden size for everything the model itself produced
    (instruction z-dict = `instr_zdict_update` embeddings, agent.py:1231-1256; front-door features = `extract_cfp_features`
    vectors, parser.py:258 cfp_file_map[student_hidden_size])"""
    over = getattr(cfg, f"{name}_dict_size", None)
    if over:
        return int(over)
    return int(cfg_get(cfg, "image_feat_size")) if name == "back_img" else int(cfg.hidden_size)


def enabled_blocks(cfg):
    return [n for n, flag in BLOCKS.items() if bool(getattr(cfg, flag, False))]


def causal_specs(cfg, p, only=None):
    H = cfg.hidden_size
    s = []
    for n in enabled_blocks(cfg):
        if only is not None and n not in only:
            continue
        q = f"{p}causal.{n}."
        Dz = dict_width(cfg, n)
        s += [(q + "query.weight", (H, H), "normal"), (q + "query.bias", (H,), "zeros"),
              (q + "key.weight", (H, Dz), "normal"), (q + "value.weight", (H, Dz), "normal"),
              (q + "key.bias", (H,), "zeros"), (q + "value.bias", (H,), "zeros"),
              (q + "output.dense.weight", (H, H), "normal"), (q + "output.dense.bias", (H,), "zeros"),
              (q + "output.LayerNorm.weight", (H,), "ones"), (q + "output.LayerNorm.bias", (H,), "zeros")]
        if getattr(cfg, "do_add_method", "add") == "door":
            s += [(q + "gate_x.weight", (1, H), "normal"), (q + "gate_x.bias", (1,), "zeros"),
                  (q + "gate_e.weight", (1, H), "normal"), (q + "gate_e.bias", (1,), "zeros")]
    return s


class _DictAttnFn(torch.autograd.Function):
    """ctx[B,N,H] = MHA(q[B,N,H], k[B,Nz,H], v[B,Nz,H]) without masks, on the engine's attention kernels"""

    @staticmethod
    def forward(ctx, q, k, v, net):
        B, N, H = q.shape
        Nz = k.shape[1]
        T = net.dtype
        qc = q.detach().to(T).reshape(B * N, H).contiguous()
        kv = torch.cat([k.detach().to(T).reshape(B * Nz, H), v.detach().to(T).reshape(B * Nz, H)], 1).contiguous()     # [B*Nz, 2H]
        flops = float(B) * N * Nz * HD * net.nh
        Pm, out, ldp, _ = net._attn_fwd(qc, H, kv, kv[:, H:], 2 * H, B, N, Nz, None, None, None, flops, None)
        ctx.net, ctx.save, ctx.dims, ctx.dt = net, (qc, kv, Pm, ldp, flops), (B, N, Nz, H), (q.dtype, k.dtype, v.dtype)
        return out.view(B, N, H)

    @staticmethod
    def backward(ctx, d_out):
        net = ctx.net
        qc, kv, Pm, ldp, flops = ctx.save
        B, N, Nz, H = ctx.dims
        T = net.dtype
        d = d_out.to(T).reshape(B * N, H).contiguous()
        dq, dkv = net.new(B * N, H), net.new(B * Nz, 2 * H)
        net._attn_bwd(Pm, ldp, d, qc, H, kv, kv[:, H:], 2 * H, dq, H, dkv, dkv[:, H:], 2 * H, B, N, Nz, None, None, None, flops)
        return (dq.view(B, N, H).to(ctx.dt[0]), dkv[:, :H].reshape(B, Nz, H).to(ctx.dt[1]), dkv[:, H:].reshape(B, Nz, H).to(ctx.dt[2]), None)


class _AddNormFn(torch.autograd.Function):
    """y = LayerNorm(x + dropout(e)) (BertSelfOutput shape) on the fused LayerNorm kernels; parameter gradients go to the store"""

    @staticmethod
    def forward(ctx, x, e, net, ln, drop, owner=None):
        ctx.owner, ctx.lane = owner, lanes.cur
        shp = x.shape
        H = shp[-1]
        M = x.numel() // H
        T = net.dtype
        xc, ec = x.detach().to(T).reshape(M, H).contiguous(), e.detach().to(T).reshape(M, H).contiguous()
        y, rstd = net.new(M, H), net.new(M, dtype=torch.float32)
        if drop:
            O.ln_fwd(M, H, y, in0=ec, in1=xc, gamma=ln.g, beta=ln.b, eps=net.eps, rstd=rstd, drop_in0=drop)
        else:
            O.ln_fwd(M, H, y, in0=ec, in1=xc, gamma=ln.g, beta=ln.b, eps=net.eps, rstd=rstd)
        ctx.net, ctx.ln, ctx.drop, ctx.save, ctx.shape, ctx.dt = net, ln, drop, (y, rstd), shp, (x.dtype, e.dtype)
        return y.view(shp)

    @staticmethod
    @lanes.lane_bwd
    def backward(ctx, dy):
        net, ln = ctx.net, ctx.ln
        y, rstd = ctx.save
        M, H = y.shape
        net.S.ensure_grads()
        if ctx.owner is not None:
            from .model_nav import _queue_sync
            _queue_sync(ctx.owner)
        d = dy.to(net.dtype).reshape(M, H).contiguous()
        d_sum = net.new(M, H)
        d_dense = net.new(M, H) if ctx.drop else None
        O.ln_bwd(M, H, d, y=y, gamma=ln.g, beta=ln.b, rstd=rstd, dx=d_sum, dgamma=getattr(ln, "dg", None),
                 dbeta=getattr(ln, "db", None), drop_dx=ctx.drop, dxm=d_dense)
        de = d_dense if ctx.drop else d_sum
        return d_sum.view(ctx.shape).to(ctx.dt[0]), de.view(ctx.shape).to(ctx.dt[1]), None, None, None, None


class _DoorGateFn(torch.autograd.Function):
    """'door' (parser.py:129-142 do_add_method): e * sigmoid(w_x . x + w_e . e + b_x + b_e) per token as one row-gate launch each way
    (csrc/rowops.hip magic_rowgate, mode 1); the four gate parameters' gradients land in their slices of the flat gradient buffer"""

    @staticmethod
    def forward(ctx, x, e, blk):
        ctx.lane = lanes.cur
        net = blk._net
        H = net.H
        names = blk._gate_names
        S = net.S
        xc = x.detach().to(net.dtype).reshape(-1, H).contiguous()
        ec = e.detach().to(net.dtype).reshape(-1, H).contiguous()
        M = xc.shape[0]
        out, g = torch.empty_like(ec), torch.empty(M, dtype=torch.float32, device=xc.device)
        O.rowgate_fwd(1, xc, M, H, S.master(names[0]).view(-1), e=ec, we=S.master(names[2]).view(-1), b0=S.master(names[1]), b1=S.master(names[3]),
                      out=out, gsave=g)
        ctx.blk, ctx.x, ctx.e, ctx.g, ctx.shape, ctx.dt = blk, xc, ec, g, e.shape, (x.dtype, e.dtype)
        return out.view(e.shape).to(e.dtype)

    @staticmethod
    @lanes.lane_bwd
    def backward(ctx, dout):
        blk = ctx.blk
        net, names = blk._net, blk._gate_names
        S, H = net.S, net.H
        S.ensure_grads()
        M = ctx.x.shape[0]
        d = dout.detach().to(net.dtype).reshape(M, H).contiguous()
        dx, de = torch.empty_like(ctx.x), torch.empty_like(ctx.e)
        O.rowgate_bwd(1, ctx.x, M, H, S.master(names[0]).view(-1), e=ctx.e, we=S.master(names[2]).view(-1), gsave=ctx.g, dout=d, dx=dx, de=de,
                      dwx=S.g(names[0]).view(-1), dwe=S.g(names[2]).view(-1), db0=S.g(names[1]), db1=S.g(names[3]))
        return dx.view(ctx.shape).to(ctx.dt[0]), de.view(ctx.shape).to(ctx.dt[1]), None


class CausalBlock(nn.Module):
    """one back-door or front-door adjustment of token features x [B, N, H] against a dictionary z [Nz, Dz] (+ prior pz [Nz])"""

    def __init__(self, model, name):
        super().__init__()
        from .model_nav import HipLinear
        net, p = model.net, f"{model.prefix}causal.{name}."
        self.name, self._net, self._model = name, net, (model,)
        self.kind = "back" if name.startswith("back") else "front"
        cfg = net.cfg
        self.btype = getattr(cfg, "do_back_txt_type", "type_2") if name == "back_txt" else \
            getattr(cfg, "do_back_imgobj_type", getattr(cfg, "do_back_img_type", "type_1")) if name == "back_img" else "type_2"
        self.door = getattr(cfg, "do_add_method", "add") == "door"
        H = net.H

        def hl(w, b, rows=None, cols=None):
            m = HipLinear()
            m._net, m._lin, m._owner = net, net.lin(p + w, p + b, rows=rows, cols=cols), (model,)
            return m
        object.__setattr__(self, "q", hl("query.weight", "query.bias"))
        object.__setattr__(self, "kv", hl("key.weight", "key.bias", rows=2 * H, cols=dict_width(cfg, name)))     # key | value adjacent
        object.__setattr__(self, "o", hl("output.dense.weight", "output.dense.bias"))
        if self.door:
            # the gate is two H-wide dot products per token: one row-gate launch each way (_DoorGateFn) on the store's own parameters
            named = dict(model.named_parameters())
            object.__setattr__(self, "_gate", tuple(named[p + n] for n in ("gate_x.weight", "gate_x.bias", "gate_e.weight", "gate_e.bias")))
            object.__setattr__(self, "_gate_names", tuple(p + n for n in ("gate_x.weight", "gate_x.bias", "gate_e.weight", "gate_e.bias")))
        self._ln_name = p + "output.LayerNorm"
        self._drop_name = p + "output.dropout"

    def forward(self, x, z, pz=None):
        """x [B,N,H]; z [B,Nz,Dz] (the reference repeats one dictionary over the batch: row 0 is used) or [Nz,Dz]; pz likewise"""
        net = self._net
        H = net.H
        B, N, _ = x.shape
        z0 = (z[0] if z.dim() == 3 else z).float()
        Nz = z0.shape[0]
        kv = self.kv(z0.to(x.device))                                  # [Nz, 2H]: batch independent, projected once
        k, v = kv[:, :H], kv[:, H:]
        if pz is not None:                                             # back-door: the prior P(z) weights every entry's value
            p0 = (pz[0] if pz.dim() == 3 else pz).reshape(Nz, 1).to(v.device, v.dtype)
            v = v * p0
        if self.kind == "back" and self.btype == "type_1":             # prior only: E_z[z] = sum_z P(z) v(z), the same for every token
            e = v.sum(0).reshape(1, 1, H).expand(B, N, H)
        else:
            q = self.q(x)
            e = _DictAttnFn.apply(q, k.unsqueeze(0).expand(B, Nz, H), v.unsqueeze(0).expand(B, Nz, H), net)
        e = self.o(e.contiguous())
        if self.door:
            e = _DoorGateFn.apply(x, e, self)
        return _AddNormFn.apply(x, e, net, net.ln(self._ln_name), net._dh(self._drop_name), self._model[0])


def build_blocks(model, only=None):
    """attach `model.causal_blocks[name]` for every block the config switches on (none by default: r2r_magic_model_config.json:60-66)"""
    return {n: CausalBlock(model, n) for n in enabled_blocks(model.config) if only is None or n in only}


def check_inputs(model, mode, batch, keys):
    """a dictionary input that is present while its block is switched off is an error (never silently ignored)"""
    need = {"instr_z_direction_features": "back_txt", "instr_z_landmark_features": "back_txt", "front_txt_feats": "front_txt",
            "z_img_features": "back_img", "front_vp_feats": "front_vp", "front_gmap_feats": "front_gmap"}
    for k in keys:
        if batch.get(k) is not None and k in need and need[k] not in model.causal_blocks:
            raise ValueError(f"VLNBert({mode!r}): input {k!r} given but config.{BLOCKS[need[k]]} is off -- the model has no "
                             f"'{need[k]}' block (map_nav_src/r2r/parser.py:129-133)")
