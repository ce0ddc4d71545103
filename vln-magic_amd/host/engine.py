"""Explicit forward/backward of the MAGIC trunk on the HIP kernels (no autograd tape).

Architecture = the oracle's restatement (oracle/model_ref.py): RoBERTa-token
text encoder (post-LN BERT blocks), 36-view panorama encoder, global (map) and local (viewpoint) METER
co-attention encoders.  Each segment has fwd(...)->Ctx and bwd(Ctx, grads); saved activations are plain
torch tensors owned by the caching allocator; every arithmetic op is a C-ABI kernel launch (host/ops.py).
Parameter gradients are accumulated (fp32 atomics) into the flat gradient buffer of the ParamStore.
"""
import math
import zlib
from types import SimpleNamespace as Ctx

import torch

from . import lanes
from . import ops as O
from .config import cfg_get

import os

HD = 64  # head dim (heads = H/64: train_r2r_magic.py:143,157)
FUSED_ATTN = not os.environ.get("MAGIC_NO_FUSED_ATTN")   # fused QK^T+softmax+PV / 5-product backward when the shape fits LDS


def rup(x, m=8):
    return (x + m - 1) // m * m


def self_layer_specs(p, H, I):
    s = []
    for n in ("query", "key", "value"):
        s.append((f"{p}attention.self.{n}.weight", (H, H), "normal"))
    for n in ("query", "key", "value"):
        s.append((f"{p}attention.self.{n}.bias", (H,), "zeros"))
    s += [(f"{p}attention.output.dense.weight", (H, H), "normal"), (f"{p}attention.output.dense.bias", (H,), "zeros"),
          (f"{p}attention.output.LayerNorm.weight", (H,), "ones"), (f"{p}attention.output.LayerNorm.bias", (H,), "zeros")]
    return s


def ffn_specs(p, H, I):
    return [(f"{p}intermediate.dense.weight", (I, H), "normal"), (f"{p}intermediate.dense.bias", (I,), "zeros"),
            (f"{p}output.dense.weight", (H, I), "normal"), (f"{p}output.dense.bias", (H,), "zeros"),
            (f"{p}output.LayerNorm.weight", (H,), "ones"), (f"{p}output.LayerNorm.bias", (H,), "zeros")]


def cross_layer_specs(p, H, I):
    s = self_layer_specs(p, H, I)
    for n in ("query", "key", "value"):
        s.append((f"{p}crossattention.self.{n}.weight", (H, H), "normal"))
    for n in ("query", "key", "value"):
        s.append((f"{p}crossattention.self.{n}.bias", (H,), "zeros"))
    s += [(f"{p}crossattention.output.dense.weight", (H, H), "normal"), (f"{p}crossattention.output.dense.bias", (H,), "zeros"),
          (f"{p}crossattention.output.LayerNorm.weight", (H,), "ones"), (f"{p}crossattention.output.LayerNorm.bias", (H,), "zeros")]
    return s + ffn_specs(p, H, I)


def cls_specs(p, H, inp=None):
    return [(f"{p}net.0.weight", (H, inp or H), "normal"), (f"{p}net.0.bias", (H,), "zeros"),
            (f"{p}net.2.weight", (H,), "ones"), (f"{p}net.2.bias", (H,), "zeros"),
            (f"{p}net.3.weight", (1, H), "normal"), (f"{p}net.3.bias", (1,), "zeros")]


def trunk_specs(cfg, p="bert."):
    H, I = cfg.hidden_size, cfg.intermediate_size
    af = cfg_get(cfg, "angle_feat_size")
    s = [(f"{p}embeddings.word_embeddings.weight", (cfg.vocab_size, H), "normal"),
         (f"{p}embeddings.position_embeddings.weight", (cfg.max_position_embeddings, H), "normal"),
         (f"{p}embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, H), "normal"),
         (f"{p}embeddings.LayerNorm.weight", (H,), "ones"), (f"{p}embeddings.LayerNorm.bias", (H,), "zeros")]
    for i in range(cfg.num_l_layers):
        s += self_layer_specs(f"{p}lang_encoder.layer.{i}.", H, I) + ffn_specs(f"{p}lang_encoder.layer.{i}.", H, I)
    q = f"{p}img_embeddings."
    s += [(q + "img_linear.weight", (H, cfg.image_feat_size), "normal"), (q + "img_linear.bias", (H,), "zeros"),
          (q + "img_layer_norm.weight", (H,), "ones"), (q + "img_layer_norm.bias", (H,), "zeros"),
          (q + "loc_linear.weight", (H, af + 3), "normal"), (q + "loc_linear.bias", (H,), "zeros"),
          (q + "loc_layer_norm.weight", (H,), "ones"), (q + "loc_layer_norm.bias", (H,), "zeros"),
          (q + "nav_type_embedding.weight", (3, H), "normal"),
          (q + "layer_norm.weight", (H,), "ones"), (q + "layer_norm.bias", (H,), "zeros")]
    for i in range(cfg.num_pano_layers):
        s += self_layer_specs(f"{q}pano_encoder.layer.{i}.", H, I) + ffn_specs(f"{q}pano_encoder.layer.{i}.", H, I)
    s += [(q + "pano_fuse_linear.weight", (1, H), "normal"), (q + "pano_fuse_linear.bias", (1,), "zeros")]
    g = f"{p}global_encoder."
    s += [(g + "gmap_pos_embeddings.0.weight", (H, af + 3), "normal"), (g + "gmap_pos_embeddings.0.bias", (H,), "zeros"),
          (g + "gmap_pos_embeddings.1.weight", (H,), "ones"), (g + "gmap_pos_embeddings.1.bias", (H,), "zeros"),
          (g + "gmap_step_embeddings.weight", (cfg.max_action_steps, H), "normal"),
          (g + "sprel_linear.weight", (1, 1), "normal"), (g + "sprel_linear.bias", (1,), "zeros")]
    for i in range(cfg.num_x_layers):
        s += cross_layer_specs(f"{g}encoder.crossattention.{i}.", H, I)
    l = f"{p}local_encoder."
    s += [(l + "vp_pos_embeddings.0.weight", (H, 2 * af + 6), "normal"), (l + "vp_pos_embeddings.0.bias", (H,), "zeros"),
          (l + "vp_pos_embeddings.1.weight", (H,), "ones"), (l + "vp_pos_embeddings.1.bias", (H,), "zeros")]
    for i in range(cfg.num_x_layers):
        s += cross_layer_specs(f"{l}encoder.crossattention.{i}.", H, I)
    Ht = getattr(cfg, "teacher_hidden_size", None)
    if Ht:
        for n in ("txt_emb_w", "kdl_img_w", "kdl_avg_img_w", "global_cross_w", "local_cross_w"):   # agent_base.py:330
            s += [(f"{p}{n}.weight", (Ht, H), "normal"), (f"{p}{n}.bias", (Ht,), "zeros")]
    return s


class Lin:
    """handle on one nn.Linear living in the ParamStore (optionally a span of fused consecutive tensors)."""

    def __init__(self, store, wname, bname, rows=None, cols=None):
        off, n, shape = store.offsets[wname]
        rows = rows or shape[0]
        cols = cols or shape[1]
        self.W = store.w_span(wname, rows, cols)
        self.b = store.master_span(bname, rows)
        self.Wm = store.master_span(wname, rows * cols).view(rows, cols)
        self._bname, self._g = bname, {}
        self.N, self.K = rows, cols
        self._store, self._wname = store, wname

    def _grads(self):
        """(dW, db) views of the CURRENT lane's flat gradient buffer (host/lanes.py)"""
        k = lanes.cur
        v = self._g.get(k)
        if v is None:
            st = self._store
            if not st.requires_grad:
                return (None, None)
            v = self._g[k] = (st.g_span(self._wname, self.N * self.K).view(self.N, self.K), st.g_span(self._bname, self.N))
        elif k:
            self._store.lane_dirty = True
        return v

    @property
    def dW(self):
        return self._grads()[0]

    @property
    def db(self):
        return self._grads()[1]

    @property
    def WT(self):
        """[K, N] bf16 = W^T (transposed shadow, csrc/encbwd.hip reads it); registers the span on first use"""
        t = getattr(self, "_WT", None)
        if t is None:
            t = self._WT = self._store.t_span(self._wname, self.N, self.K)
        return t


    @property
    def WTf(self):
        """W^T in MFMA-fragment order (flat; csrc/encbwd.hip reads it)"""
        t = getattr(self, "_WTf", None)
        if t is None:
            t = self._WTf = self._store.tf_span(self._wname, self.N, self.K)
        return t

    @property
    def Wf(self):
        """W in MFMA-fragment order (flat; csrc/chain.hip reads it); registers the span on first use"""
        t = getattr(self, "_Wf", None)
        if t is None:
            t = self._Wf = self._store.f_span(self._wname, self.N, self.K)
        return t


class LN:
    def __init__(self, store, wname, bname):
        self.g, self.b = store.master(wname), store.master(bname)
        self._store, self._names, self._g = store, (wname, bname), {}

    def _grads(self):
        k = lanes.cur
        v = self._g.get(k)
        if v is None:
            if not self._store.requires_grad:
                return (None, None)
            v = self._g[k] = (self._store.g(self._names[0]), self._store.g(self._names[1]))
        elif k:
            self._store.lane_dirty = True
        return v

    @property
    def dg(self):
        return self._grads()[0]

    @property
    def db(self):
        return self._grads()[1]


class MagicNet:
    def __init__(self, cfg, store, prefix="bert."):
        self.cfg, self.S, self.p = cfg, store, prefix
        self.H, self.I = cfg.hidden_size, cfg.intermediate_size
        self.nh = cfg.num_attention_heads
        assert self.H == self.nh * HD, "heads = H/64"
        self.eps = cfg.layer_norm_eps
        self.dtype = store.compute_dtype
        self.train = store.requires_grad
        self._cache = {}
        self.drop = None          # (seed uint32[2] device tensor, p_hidden, p_attn) while a training-mode forward is running
        if self.S.device.type == "cuda" and self.rbw_ok():
            # the backward row-block kernel reads W^T: register the transposed-shadow spans of every self-attention block now, so the
            # first sync_shadow() of the first forward fills them in
            for fmt, nl in ((prefix + "lang_encoder.layer.{}.", cfg.num_l_layers), (prefix + "img_embeddings.pano_encoder.layer.{}.", cfg.num_pano_layers)):
                for i in range(nl):
                    lp = fmt.format(i)
                    for lin in (self.lin(lp + "attention.self.query.weight", rows=3 * self.H, cols=self.H), self.lin(lp + "attention.output.dense.weight"),
                                self.lin(lp + "intermediate.dense.weight"), self.lin(lp + "output.dense.weight")):
                        lin.WTf
                        lin.Wf           # (the whole-encoder forward reads W in fragment order)
            for enc in ("global_encoder.", "local_encoder."):
                for i in range(cfg.num_x_layers):
                    lp = f"{prefix}{enc}encoder.crossattention.{i}."
                    for lin in (self.lin(lp + "attention.self.query.weight", rows=3 * self.H, cols=self.H), self.lin(lp + "attention.output.dense.weight"),
                                self.lin(lp + "crossattention.self.query.weight"), self.lin(lp + "crossattention.output.dense.weight"),
                                self.lin(lp + "intermediate.dense.weight"), self.lin(lp + "output.dense.weight")):
                        lin.WTf
                        lin.Wf
                    self.lin(lp + "crossattention.self.key.weight", lp + "crossattention.self.key.bias", rows=2 * self.H, cols=self.H).Wf

    # ---- dropout sites (counter-based masks, csrc/common.hpp) -------------------------------------
    def set_dropout(self, seed=None, p_hidden=0.0, p_attn=0.0):
        """Called by the model before each forward: seed = fresh uint32[2] device tensor (None = eval mode)."""
        self.drop = (seed, float(p_hidden), float(p_attn)) if (seed is not None and (p_hidden > 0 or p_attn > 0)) else None

    @staticmethod
    def site_id(name):
        return (zlib.crc32(name.encode()) & 0xFFFFFFFF) or 1

    def _dh(self, name):
        """hidden-dropout descriptor (seed, p, site) of the module called `name`, or None"""
        if self.drop is None or self.drop[1] <= 0:
            return None
        return (self.drop[0], self.drop[1], self.site_id(name))

    def _da(self, name):
        if self.drop is None or self.drop[2] <= 0:
            return None
        return (self.drop[0], self.drop[2], self.site_id(name))

    # ---- parameter handle cache ---------------------------------------------------------------
    def lin(self, w, b=None, rows=None, cols=None):
        key = ("lin", w, rows, cols)
        if key not in self._cache:
            self._cache[key] = Lin(self.S, w, b or w[:-len("weight")] + "bias", rows, cols)
        return self._cache[key]

    def ln(self, base):
        key = ("ln", base)
        if key not in self._cache:
            self._cache[key] = LN(self.S, base + ".weight", base + ".bias")
        return self._cache[key]

    def new(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dtype, device=self.S.device)

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or self.dtype, device=self.S.device)

    # ---- attention core -----------------------------------------------------------------------
    def _attn_fwd(self, q, ldq, k, v, ldkv, Bn, Nq, Nk, kmask, dist, sprel, flops, drop=None):
        """returns (P clean softmax [kept for the backward], ctx, ldp, P as exposed = dropped probabilities under dropout:
        HF/METER BertSelfAttention returns the attention map AFTER its dropout)."""
        nh, H = self.nh, self.H
        ldp = rup(Nk)
        if FUSED_ATTN and O.attn_supported(self.dtype, Nq, Nk, False):
            Pm, ctx = self.new(Bn, nh, Nq, ldp), self.new(Bn * Nq, H)
            Pd = self.new(Bn, nh, Nq, ldp) if drop else None
            O.attn_fwd(q, ldq, k, v, ldkv, Pm, ldp, ctx, Bn, nh, Nq, Nk, H, 1.0 / math.sqrt(HD), kmask=kmask, dist=dist,
                       sprel_w=sprel[0] if sprel else None, sprel_b=sprel[1] if sprel else None, flops=flops, drop=drop, Pd=Pd)
            return Pm, ctx, ldp, (Pd if drop else Pm)
        S = self.new(Bn, nh, Nq, ldp, dtype=torch.float32)
        O.gemm(0, q, k, S, Nq, Nk, HD, ldq, ldkv, ldp, batch=Bn * nh, nh=nh, sA=(Nq * ldq, HD), sB=(Nk * ldkv, HD),
               sC=(nh * Nq * ldp, Nq * ldp), flop_dims=(1, 1, flops / (nh * Bn)))
        Pm = self.new(Bn, nh, Nq, ldp)
        O.softmax_fwd(S, Pm, Bn, nh, Nq, Nk, ldp, 1.0 / math.sqrt(HD), kmask=kmask, dist=dist,
                      sprel_w=sprel[0] if sprel else None, sprel_b=sprel[1] if sprel else None)
        Pu = Pm
        if drop:
            Pu = self.zeros(Bn, nh, Nq, ldp)
            O.dropout(Pm, Pu, Bn * nh * Nq, Nk, ldp, drop)
        ctx = self.new(Bn * Nq, H)
        O.gemm(1, Pu, v, ctx, Nq, HD, Nk, ldp, ldkv, H, batch=Bn * nh, nh=nh, sA=(nh * Nq * ldp, Nq * ldp), sB=(Nk * ldkv, HD),
               sC=(Nq * H, HD), flop_dims=(1, 1, flops / (nh * Bn)))
        return Pm, ctx, ldp, Pu

    def _attn_bwd(self, Pm, ldp, d_ctx, q, ldq, k, v, ldkv, dq, lddq, dk, dv, lddkv, Bn, Nq, Nk, dist, dsprel, dP_init, flops,
                  drop=None, Pu=None, o=None, acc_kv=False):
        """Pm = clean softmax; under dropout Pu = the dropped probabilities the product used (dP_init = dLoss/dPu).  o = the forward's
        output (long keys: the key-split kernel takes rowsum(P dP) from dO . O); acc_kv: dk / dv are added to (long-key kernel only)."""
        nh, H = self.nh, self.H
        if FUSED_ATTN and o is not None and dist is None and dP_init is None and O.attn_bwd_ks_ok(self.dtype, Nq, Nk):
            O.attn_bwd_ks(q, ldq, k, v, ldkv, Pm, ldp, o, d_ctx, Bn, nh, Nq, Nk, H, 1.0 / math.sqrt(HD), dq, lddq, dk, dv, lddkv,
                          accumulate_kv=acc_kv, flops=flops, drop=drop)
            return
        assert not acc_kv, "accumulating dK / dV needs the key-split attention backward"
        if FUSED_ATTN and O.attn_supported(self.dtype, Nq, Nk, True):
            O.attn_bwd(q, ldq, k, v, ldkv, Pm, ldp, d_ctx, Bn, nh, Nq, Nk, H, 1.0 / math.sqrt(HD), dP_init, dq, lddq, dk, dv, lddkv,
                       dist=dist, dsprel_w=dsprel[0] if dsprel else None, dsprel_b=dsprel[1] if dsprel else None, flops=flops, drop=drop)
            return
        fd = (1, 1, flops / (nh * Bn))      # x (batch = Bn*nh) in the counter -> 2 * sum_b(lq*lk) * 64 * nh
        sP = (nh * Nq * ldp, Nq * ldp)
        if drop and Pu is None:
            Pu = self.zeros(Bn, nh, Nq, ldp)
            O.dropout(Pm, Pu, Bn * nh * Nq, Nk, ldp, drop)
        # dV = Pu^T dO
        O.gemm(2, Pu if drop else Pm, d_ctx, dv, Nk, HD, Nq, ldp, H, lddkv, batch=Bn * nh, nh=nh, sA=sP, sB=(Nq * H, HD), sC=(Nk * lddkv, HD), flop_dims=fd)
        # dP = dO V^T (+ KD gradient already sitting in dP_init)
        dP = dP_init if dP_init is not None else self.new(Bn, nh, Nq, ldp, dtype=torch.float32)
        O.gemm(0, d_ctx, v, dP, Nq, Nk, HD, H, ldkv, ldp, batch=Bn * nh, nh=nh, sA=(Nq * H, HD), sB=(Nk * ldkv, HD), sC=sP,
               residual=dP if dP_init is not None else None, ldr=ldp, flop_dims=fd)
        if drop:                                # gradient wrt the dropped probabilities -> wrt the clean softmax
            O.dropout(dP, dP, Bn * nh * Nq, Nk, ldp, drop)
        dS = self.new(Bn, nh, Nq, ldp)
        O.softmax_bwd(Pm, dP, dS, Bn, nh, Nq, Nk, ldp, 1.0 / math.sqrt(HD), dist=dist,
                      dsprel_w=dsprel[0] if dsprel else None, dsprel_b=dsprel[1] if dsprel else None)
        # dQ = dS K ; dK = dS^T Q
        O.gemm(1, dS, k, dq, Nq, HD, Nk, ldp, ldkv, lddq, batch=Bn * nh, nh=nh, sA=sP, sB=(Nk * ldkv, HD), sC=(Nq * lddq, HD), flop_dims=fd)
        O.gemm(2, dS, q, dk, Nk, HD, Nq, ldp, ldq, lddkv, batch=Bn * nh, nh=nh, sA=sP, sB=(Nq * ldq, HD), sC=(Nk * lddkv, HD), flop_dims=fd)

    # ---- self-attention + add&norm -------------------------------------------------------------
    def chain_ok(self):
        """forward-only row chain (csrc/chain.hip): a frozen network (no backward, no dropout) at the teacher's width"""
        return (not self.train) and self.drop is None and O.chain_ok(self.dtype, self.H, self.I)

    def _ffn_args(self, lp):
        f1, f2, n = self.lin(lp + "intermediate.dense.weight"), self.lin(lp + "output.dense.weight"), self.ln(lp + "output.LayerNorm")
        return (f1.Wf, f1.b, f2.Wf, f2.b, n.g, n.b, f1.N)

    def _next_qkv(self, next_lp, M):
        """(weights of the following block's fused Q|K|V projection, its output buffer) or (None, None)"""
        if next_lp is None:
            return None, None
        ql = self.lin(next_lp + "attention.self.query.weight", rows=3 * self.H, cols=self.H)
        return (ql.Wf, ql.b, 3 * self.H), self.new(M, 3 * self.H)

    def _sa_attn_fwd(self, lp, x, Bn, N, kmask, dist, sprel, rows, aflops, qkv=None, alloc_a=True):
        """Q/K/V projection (unless the previous block's row chain already produced it) + the attention product"""
        H = self.H
        M = Bn * N
        c = Ctx(x=x, Bn=Bn, N=N, rows=rows, aflops=aflops, dist=dist)
        if qkv is None:
            ql = self.lin(lp + "attention.self.query.weight", rows=3 * H, cols=H)
            qkv = O.linear_fwd(x, ql.W, ql.b, M, flop_rows=rows)
        c.qkv = qkv
        c.adrop, c.hdrop = self._da(lp + "attention.self.dropout"), self._dh(lp + "attention.output.dropout")
        c.Ppre, c.ctx, c.ldp, c.P = self._attn_fwd(c.qkv, 3 * H, c.qkv[:, H:], c.qkv[:, 2 * H:], 3 * H, Bn, N, N, kmask, dist, sprel,
                                                   aflops, c.adrop)
        if alloc_a:
            c.a, c.rstd_a = self.new(M, H), self.new(M, dtype=torch.float32)
        return c

    def _sa_out_fwd(self, lp, c):
        o = self.lin(lp + "attention.output.dense.weight")
        n = self.ln(lp + "attention.output.LayerNorm")
        self._dense_add_ln(c.ctx, o, c.x, n, c.Bn * c.N, c.a, c.rstd_a, c.rows, c.hdrop)

    def _sa_fwd(self, lp, x, Bn, N, kmask, dist, sprel, rows, aflops, qkv=None):
        c = self._sa_attn_fwd(lp, x, Bn, N, kmask, dist, sprel, rows, aflops, qkv)
        self._sa_out_fwd(lp, c)
        return c

    def _dense_add_ln(self, x, lin, res, n, M, out, rstd, rows, hdrop):
        """out = LayerNorm(dropout(x W^T + b) + res): BertSelfOutput / BertOutput"""
        H = self.H
        if O.linear_ln_ok(H, lin.K):
            O.linear_ln(x, lin.W, lin.b, M, res, n.g, n.b, self.eps, out, rstd, flop_rows=rows, drop=hdrop)
        elif hdrop:
            d = O.linear_fwd(x, lin.W, lin.b, M, flop_rows=rows)
            O.ln_fwd(M, H, out, in0=d, in1=res, gamma=n.g, beta=n.b, eps=self.eps, rstd=rstd, drop_in0=hdrop)
        else:
            d = O.linear_fwd(x, lin.W, lin.b, M, residual=res, flop_rows=rows)
            O.ln_fwd(M, H, out, in0=d, gamma=n.g, beta=n.b, eps=self.eps, rstd=rstd)

    def _add_ln_bwd(self, n, M, dy, y, rstd, hdrop):
        """backward of _dense_add_ln's LayerNorm: returns (d_sum for the residual branch, d_dense for the dense branch)"""
        d_sum = self.new(M, self.H)
        d_dense = self.new(M, self.H) if hdrop else None
        O.ln_bwd(M, self.H, dy, y=y, gamma=n.g, beta=n.b, rstd=rstd, dx=d_sum, dgamma=n.dg, dbeta=n.db, drop_dx=hdrop, dxm=d_dense)
        return d_sum, (d_dense if hdrop else d_sum)

    # A gradient handed between backward segments is either a plain tensor (wrt a LayerNorm OUTPUT) or a `Pre` pair (already
    # pushed through that LayerNorm by the producer's fused GEMM epilogue: magic_linear_lnbwd).
    def _ln_desc(self, base, y, rstd, hdrop):
        return Ctx(n=self.ln(base), y=y, rstd=rstd, hdrop=hdrop)

    def _through_ln(self, d, desc, M):
        """(d_sum, d_dense) of the LayerNorm described by desc, from d = plain gradient or an already-processed Pre pair"""
        if isinstance(d, tuple):
            return d
        return self._add_ln_bwd(desc.n, M, d, desc.y, desc.rstd, desc.hdrop)

    def _dx_into_ln(self, dy, lin, M, residual, fuse, rows):
        """input gradient dy @ W (+ residual).  With `fuse` (descriptor of the LayerNorm that produced this dense's input) and a
        supported shape the LayerNorm backward runs in the GEMM epilogue and a Pre pair is returned; else the plain tensor."""
        if fuse is not None and O.linear_lnbwd_ok(self.H, lin.W.shape[0]) and lin.W.shape[1] == self.H:
            d_sum = self.new(M, self.H)
            d_dense = self.new(M, self.H) if fuse.hdrop else None
            O.linear_lnbwd(dy, lin.W, M, residual, fuse.y, fuse.n.g, fuse.n.b, fuse.rstd, d_sum, fuse.n.dg, fuse.n.db,
                           drop=fuse.hdrop, dxm=d_dense, flop_rows=rows)
            return (d_sum, d_dense if fuse.hdrop else d_sum)
        return O.linear_dx(dy, lin.W, M, residual=residual, flop_rows=rows)

    def _sa_bwd(self, lp, c, d_a, dsprel=None, dP_init=None, fuse=None):
        """d_a: gradient wrt the attention-output LayerNorm's output (plain) or its Pre pair; fuse: descriptor of the LayerNorm
        that produced this block's input x (the previous block's output LN), or None."""
        H, Bn, N = self.H, c.Bn, c.N
        M = Bn * N
        d_ao, d_aod = self._through_ln(d_a, self._ln_desc(lp + "attention.output.LayerNorm", c.a, c.rstd_a, c.hdrop), M)
        o = self.lin(lp + "attention.output.dense.weight")
        O.linear_dw(d_aod, c.ctx, o.dW, o.db, M, flop_rows=c.rows)
        d_ctx = O.linear_dx(d_aod, o.W, M, flop_rows=c.rows)
        dqkv = self.new(M, 3 * H)
        self._attn_bwd(c.Ppre, c.ldp, d_ctx, c.qkv, 3 * H, c.qkv[:, H:], c.qkv[:, 2 * H:], 3 * H,
                       dqkv, 3 * H, dqkv[:, H:], dqkv[:, 2 * H:], 3 * H, Bn, N, N, c.dist, dsprel, dP_init, c.aflops,
                       c.adrop, c.P if c.adrop else None, o=c.ctx)
        qkv = self.lin(lp + "attention.self.query.weight", rows=3 * H, cols=H)
        O.linear_dw(dqkv, c.x, qkv.dW, qkv.db, M, flop_rows=c.rows)
        return self._dx_into_ln(dqkv, qkv, M, d_ao, fuse, c.rows)

    # ---- FFN + add&norm ------------------------------------------------------------------------
    def _ffn_fwd(self, lp, a, M, rows):
        H, I = self.H, self.I
        c = Ctx(a=a, M=M, rows=rows)
        f1, f2 = self.lin(lp + "intermediate.dense.weight"), self.lin(lp + "output.dense.weight")
        c.z = self.new(M, I)
        c.g = O.linear_fwd(a, f1.W, f1.b, M, epilogue=1, pre=c.z, flop_rows=rows)
        n = self.ln(lp + "output.LayerNorm")
        c.out, c.rstd = self.new(M, H), self.new(M, dtype=torch.float32)
        c.hdrop = self._dh(lp + "output.dropout")
        self._dense_add_ln(c.g, f2, a, n, M, c.out, c.rstd, rows, c.hdrop)
        return c

    def _ffn_bwd(self, lp, c, dout, fuse=None):
        """dout: plain gradient wrt the block output or its Pre pair; fuse: descriptor of the LayerNorm that produced c.a"""
        H, M = self.H, c.M
        d_fo, d_fod = self._through_ln(dout, self._ln_desc(lp + "output.LayerNorm", c.out, c.rstd, c.hdrop), M)
        f1, f2 = self.lin(lp + "intermediate.dense.weight"), self.lin(lp + "output.dense.weight")
        O.linear_dw(d_fod, c.g, f2.dW, f2.db, M, flop_rows=c.rows)
        d_z = O.linear_dx(d_fod, f2.W, M, epilogue=3, aux=c.z, flop_rows=c.rows)
        O.linear_dw(d_z, c.a, f1.dW, f1.db, M, flop_rows=c.rows)
        return self._dx_into_ln(d_z, f1, M, d_fo, fuse, c.rows)

    # ---- layers --------------------------------------------------------------------------------
    def self_layer_fwd(self, lp, x, Bn, N, kmask, rows, aflops, qkv=None, next_lp=None):
        """qkv: this block's Q/K/V projection if the previous block's chain produced it; next_lp: the following block, whose
        projection this block's chain produces (returned as c.next_qkv)."""
        c = Ctx(next_qkv=None)
        M = Bn * N
        if self.chain_ok():
            # frozen network: attention product, then everything per token up to the next block's Q|K|V projection as ONE launch
            c.sa = self._sa_attn_fwd(lp, x, Bn, N, kmask, None, None, rows, aflops, qkv, alloc_a=False)
            o, n1 = self.lin(lp + "attention.output.dense.weight"), self.ln(lp + "attention.output.LayerNorm")
            proj, c.next_qkv = self._next_qkv(next_lp, M)
            c.out = self.new(M, self.H)
            O.chain_fwd(c.sa.ctx, x, M, o.Wf, o.b, n1.g, n1.b, self.eps, ffn=self._ffn_args(lp), y2=c.out, proj=proj, proj_out=c.next_qkv,
                        flop_rows=rows)
            c.ffn = Ctx(out=c.out)
            c.P, c.ldp = c.sa.P, c.sa.ldp
            return c
        c.sa = self._sa_fwd(lp, x, Bn, N, kmask, None, None, rows, aflops, qkv)
        c.ffn = self._ffn_fwd(lp, c.sa.a, M, rows)
        c.out, c.P, c.ldp = c.ffn.out, c.sa.P, c.sa.ldp
        return c

    def self_layer_bwd(self, lp, c, dout, dP_init=None, fuse_in=None):
        """fuse_in: descriptor of the LayerNorm that produced this block's input (then a Pre pair is returned)"""
        d_a = self._ffn_bwd(lp, c.ffn, dout, fuse=self._ln_desc(lp + "attention.output.LayerNorm", c.sa.a, c.sa.rstd_a, c.sa.hdrop))
        return self._sa_bwd(lp, c.sa, d_a, None, dP_init, fuse=fuse_in)

    def _out_ln_desc(self, lp, lc):
        """descriptor of a block's output LayerNorm (BertOutput.LayerNorm)"""
        return self._ln_desc(lp + "output.LayerNorm", lc.ffn.out, lc.ffn.rstd, lc.ffn.hdrop)

    def cross_layer_fwd(self, lp, x, Bn, Nq, kmask, dist, sprel, ctx, Nk, ckmask, rows, crow, sflops, cflops, qkv=None, next_lp=None,
                        kv=None):
        """kv: this layer's key/value projection of the context [Bn*Nk, 2H], computed once by the caller (navigator loop: the
        text does not change between the steps of an episode, model_nav.VLNBert.text_kv); its gradient is then written to
        `dkv` of cross_layer_bwd instead of going through the projection here."""
        H = self.H
        Mq, Mk = Bn * Nq, Bn * Nk
        c = Ctx(Bn=Bn, Nq=Nq, Nk=Nk, ctx=ctx, rows=rows, crow=crow, cflops=cflops, next_qkv=None, kv_given=kv is not None)
        ql = self.lin(lp + "crossattention.self.query.weight")
        kvl = self.lin(lp + "crossattention.self.key.weight", lp + "crossattention.self.key.bias", rows=2 * H, cols=H)
        chain = self.chain_ok()
        if chain:         # self-attention output block + the cross-attention's query projection: one launch
            c.sa = self._sa_attn_fwd(lp, x, Bn, Nq, kmask, dist, sprel, rows, sflops, qkv, alloc_a=False)
            so, sn = self.lin(lp + "attention.output.dense.weight"), self.ln(lp + "attention.output.LayerNorm")
            c.sa.a, c.q = self.new(Mq, H), self.new(Mq, H)
            O.chain_fwd(c.sa.ctx, x, Mq, so.Wf, so.b, sn.g, sn.b, self.eps, y1=c.sa.a, proj=(ql.Wf, ql.b, H), proj_out=c.q, flop_rows=rows)
        else:
            c.sa = self._sa_fwd(lp, x, Bn, Nq, kmask, dist, sprel, rows, sflops, qkv)
            c.q = O.linear_fwd(c.sa.a, ql.W, ql.b, Mq, flop_rows=rows)
        s = c.sa.a
        c.kv = kv if kv is not None else O.linear_fwd(ctx, kvl.W, kvl.b, Mk, flop_rows=crow)
        c.adrop, c.hdrop = self._da(lp + "crossattention.self.dropout"), self._dh(lp + "crossattention.output.dropout")
        c.Ppre, c.cctx, c.ldp, c.P = self._attn_fwd(c.q, H, c.kv, c.kv[:, H:], 2 * H, Bn, Nq, Nk, ckmask, None, None, cflops, c.adrop)
        o = self.lin(lp + "crossattention.output.dense.weight")
        n = self.ln(lp + "crossattention.output.LayerNorm")
        if chain:         # cross-attention output block + FFN + the next block's Q|K|V projection: one launch
            proj, c.next_qkv = self._next_qkv(next_lp, Mq)
            c.out = self.new(Mq, H)
            O.chain_fwd(c.cctx, s, Mq, o.Wf, o.b, n.g, n.b, self.eps, ffn=self._ffn_args(lp), y2=c.out, proj=proj, proj_out=c.next_qkv,
                        flop_rows=rows)
            c.ffn = Ctx(out=c.out)
            return c
        c.c, c.rstd_c = self.new(Mq, H), self.new(Mq, dtype=torch.float32)
        self._dense_add_ln(c.cctx, o, s, n, Mq, c.c, c.rstd_c, rows, c.hdrop)
        c.ffn = self._ffn_fwd(lp, c.c, Mq, rows)
        c.out = c.ffn.out
        return c

    def cross_layer_bwd(self, lp, c, dout, d_ctx_acc, dsprel=None, dP_init=None, fuse_in=None, dkv_out=None, acc_kv=False):
        """returns dx (plain, or a Pre pair when fuse_in is given); accumulates the gradient wrt the context (other modality)
        into d_ctx_acc.  acc_kv (with a cached K/V projection, `dkv_out`): this step's dK / dV are ADDED to dkv_out -- the per-episode
        accumulator of the navigator loop -- instead of written."""
        H, Bn, Nq, Nk = self.H, c.Bn, c.Nq, c.Nk
        Mq, Mk = Bn * Nq, Bn * Nk
        c_ln = self._ln_desc(lp + "crossattention.output.LayerNorm", c.c, c.rstd_c, c.hdrop)
        d_c = self._ffn_bwd(lp, c.ffn, dout, fuse=c_ln)
        d_co, d_cod = self._through_ln(d_c, c_ln, Mq)
        o = self.lin(lp + "crossattention.output.dense.weight")
        O.linear_dw(d_cod, c.cctx, o.dW, o.db, Mq, flop_rows=c.rows)
        d_cctx = O.linear_dx(d_cod, o.W, Mq, flop_rows=c.rows)
        dq = self.new(Mq, H)
        dkv = dkv_out if c.kv_given else self.new(Mk, 2 * H)
        acc = bool(acc_kv and c.kv_given)
        ks = FUSED_ATTN and dP_init is None and O.attn_bwd_ks_ok(self.dtype, Nq, Nk)
        dkv_w = self.new(Mk, 2 * H) if (acc and not ks) else dkv      # (no accumulating form of this shape's kernel: own buffer, one add)
        self._attn_bwd(c.Ppre, c.ldp, d_cctx, c.q, H, c.kv, c.kv[:, H:], 2 * H, dq, H, dkv_w, dkv_w[:, H:], 2 * H,
                       Bn, Nq, Nk, None, None, dP_init, c.cflops, c.adrop, c.P if c.adrop else None, o=c.cctx, acc_kv=acc and ks)
        if acc and not ks:
            O.add_(dkv, dkv_w)
        ql = self.lin(lp + "crossattention.self.query.weight")
        kvl = self.lin(lp + "crossattention.self.key.weight", lp + "crossattention.self.key.bias", rows=2 * H, cols=H)
        O.linear_dw(dq, c.sa.a, ql.dW, ql.db, Mq, flop_rows=c.rows)
        d_s = self._dx_into_ln(dq, ql, Mq, d_co, self._ln_desc(lp + "attention.output.LayerNorm", c.sa.a, c.sa.rstd_a, c.sa.hdrop), c.rows)
        if not c.kv_given:
            O.linear_dw(dkv, c.ctx, kvl.dW, kvl.db, Mk, flop_rows=c.crow)
            O.linear_dx(dkv, kvl.W, Mk, out=d_ctx_acc, residual=d_ctx_acc, flop_rows=c.crow)
        return self._sa_bwd(lp, c.sa, d_s, dsprel, None, fuse=fuse_in)

    # ---- whole-encoder launch (csrc/encoder.hip) ------------------------------------------------------
    def enc_ok(self, N, nlayers):
        return O.encoder_ok(self.dtype, self.H, self.I, self.nh, N, nlayers)

    def _enc_segment(self, prefix_fmt, nl, x, Bn, N, kmask, rows, aflops):
        """allocate what the backward reads for `nl` post-LN self-attention blocks and describe them for O.encoder_fwd;
        returns (segment dict, [per-layer Ctx in the shape self_layer_fwd builds])"""
        H, I, nh = self.H, self.I, self.nh
        M, ldp = Bn * N, rup(N)
        layers, descs = [], []
        for i in range(nl):
            lp = prefix_fmt.format(i)
            sa = Ctx(x=x, Bn=Bn, N=N, rows=rows, aflops=aflops, dist=None, qkv=self.new(M, 3 * H), ldp=ldp,
                     adrop=self._da(lp + "attention.self.dropout"), hdrop=self._dh(lp + "attention.output.dropout"))
            sa.Ppre, sa.ctx, sa.a, sa.rstd_a = self.new(Bn, nh, N, ldp), self.new(M, H), self.new(M, H), self.new(M, dtype=torch.float32)
            sa.P = self.new(Bn, nh, N, ldp) if sa.adrop else sa.Ppre
            ffn = Ctx(a=sa.a, M=M, rows=rows, z=self.new(M, I), g=self.new(M, I), out=self.new(M, H), rstd=self.new(M, dtype=torch.float32),
                      hdrop=self._dh(lp + "output.dropout"))
            ql = self.lin(lp + "attention.self.query.weight", rows=3 * H, cols=H)
            o, n1 = self.lin(lp + "attention.output.dense.weight"), self.ln(lp + "attention.output.LayerNorm")
            f1, f2, n2 = self.lin(lp + "intermediate.dense.weight"), self.lin(lp + "output.dense.weight"), self.ln(lp + "output.LayerNorm")
            descs.append(dict(Wqkv=ql.Wf, bqkv=ql.b, Wo=o.Wf, bo=o.b, g1=n1.g, be1=n1.b, W1=f1.Wf, bi=f1.b, W2=f2.Wf, bo2=f2.b, g2=n2.g, be2=n2.b,
                              qkv=sa.qkv, P=sa.Ppre, Pd=sa.P if sa.adrop else None, ctx=sa.ctx, a=sa.a, z=ffn.z, g=ffn.g, out=ffn.out,
                              rstd_a=sa.rstd_a, rstd_o=ffn.rstd,
                              site_attn=sa.adrop[2] if sa.adrop else 0, site_ao=sa.hdrop[2] if sa.hdrop else 0,
                              site_out=ffn.hdrop[2] if ffn.hdrop else 0))
            layers.append(Ctx(next_qkv=None, sa=sa, ffn=ffn, out=ffn.out, P=sa.P, ldp=ldp))
            x = ffn.out
        flops = nl * (2.0 * rows * (3 * H * H + H * H + 2 * H * I) + 4.0 * aflops)
        return dict(x=layers[0].sa.x, kmask=kmask, nsamp=Bn, N=N, ldp=ldp, layers=descs, flops=flops), layers

    def _enc_launch(self, segs):
        d = self.drop
        O.encoder_fwd(segs, d[0] if d else None, d[2] if d else 0.0, d[1] if d else 0.0, self.eps, 1.0 / math.sqrt(HD))

    def encoders_fwd(self, ct, cp):
        """text (ct) and panorama (cp) contexts prepared with defer=True: run both encoders in ONE launch and finish them"""
        segs = []
        for c in (ct, cp):
            sg, c.layers = self._enc_segment(*c.pending)
            segs.append(sg)
            c.pending = None
        self._enc_launch(segs)
        self._text_tail(ct)
        self._pano_tail(cp, cp.plan)

    # ---- text encoder --------------------------------------------------------------------------
    def _flops_attn(self, lens_q, lens_k):
        return float(sum(a * b for a, b in zip(lens_q, lens_k))) * HD * self.nh

    def _text_embed_args(self, plan):
        """(ctx with the embedding buffers, keyword arguments of the embedding's ln_fwd)"""
        p, H = self.p, self.H
        B, L = plan["B"], plan["L"]
        M = B * L
        c = Ctx(B=B, L=L)
        n = self.ln(p + "embeddings.LayerNorm")
        c.E, c.rstd_e = self.new(M, H), self.new(M, dtype=torch.float32)
        c.edrop = self._dh(p + "embeddings.dropout")
        c.Ed = self.new(M, H) if c.edrop else None
        kw = dict(tabs=((self.S.w(p + "embeddings.word_embeddings.weight"), plan["txt_ids"], 0, 0),
                        (self.S.w(p + "embeddings.position_embeddings.weight"), None, L, 2),
                        (self.S.w(p + "embeddings.token_type_embeddings.weight"), None, 0, 0)),
                  gamma=n.g, beta=n.b, eps=self.eps, rstd=c.rstd_e, drop_out=c.edrop, out_drop=c.Ed)
        return c, M, kw

    def text_fwd(self, plan, defer=False, c=None):
        """defer: stop after the embeddings and leave the layers to `encoders_fwd` (one launch shared with the panorama encoder);
        c: the embeddings are done already (`embeds_fwd`)"""
        p, H = self.p, self.H
        B, L = plan["B"], plan["L"]
        if c is None:
            c, M, kw = self._text_embed_args(plan)
            O.ln_fwd(M, H, c.E, **kw)
        x = c.Ed if c.edrop else c.E
        c.layers = []
        tl = plan["lens"]["txt"]
        af = self._flops_attn(tl, tl)
        nl, qkv = self.cfg.num_l_layers, None
        if self.enc_ok(L, nl):
            c.pending = (p + "lang_encoder.layer.{}.", nl, x, B, L, plan["txt_mask"], plan["txt_tokens"], af)
            if defer:
                return c
            sg, c.layers = self._enc_segment(*c.pending)
            c.pending = None
            self._enc_launch([sg])
            return self._text_tail(c)
        for i in range(nl):
            lc = self.self_layer_fwd(f"{p}lang_encoder.layer.{i}.", x, B, L, plan["txt_mask"], plan["txt_tokens"], af, qkv=qkv,
                                     next_lp=f"{p}lang_encoder.layer.{i + 1}." if i + 1 < nl else None)
            c.layers.append(lc)
            x, qkv = lc.out, lc.next_qkv
        return self._text_tail(c)

    def _text_tail(self, c):
        c.out, c.P, c.ldp = c.layers[-1].out, c.layers[-1].P, c.layers[-1].ldp
        return c

    # ---- backward of whole self-attention stacks on the row-block kernel (csrc/encbwd.hip) --------------------------------
    def rbw_ok(self):
        return self.train and O.rowbwd_ok(self.dtype, self.H, self.I)

    def self_stacks_bwd(self, stacks, on_iter=None):
        """stacks: 1 or 2 tuples (ctx with .layers, layer-prefix format, d_top = plain gradient wrt the stack's output, dP_init for the top block's
        attention map).  Returns the gradients wrt the stacks' inputs.  on_iter(k): called when the top k blocks of every stack are done and their weight
        gradients queued -- the data-parallel exchange cuts its buckets there (trainer.GradSync).

        Two launch forms (csrc/encbwd.hip), chosen per step:
          * ALTERNATING (rounds 2-5): per block the per-token chain (magic_rowbwd: tail of the block above + FFN + both LayerNorm backwards + output
            projection) and the attention backward (magic_attn_bwd, one workgroup per (sample, head)) as two launches, shared by the stacks that are
            still running (text 6 blocks || panorama 2): flat 32-row tiles, the right shape for many short samples;
          * INSIDE (round 6): the attention backward of block j+1 in front of block j's chain in ONE launch, a workgroup per 16-row tile of one sample
            (attn_tile_stage) -- n + 1 launches for n blocks instead of 2 n + 1.  Taken once a single stack is left and its samples are long enough
            to fill 16-row tiles (the text stack from its third block down; a panorama's 36 rows would be three tiles with the last nearly empty:
            measured 146 us for the two shared launches against ~110 for the alternating form, profiles/r06_*).
        MAGIC_RBW_ATTN=0: alternating everywhere; MAGIC_RBW_ATTN=2: inside for every stack from the top (the first round-6 form)."""
        from . import lib as L
        H, I = self.H, self.I
        st = []
        for c, fmt, d_top, dP in stacks:
            lc = c.layers[-1]
            N = lc.sa.N
            can = O.rowbwd_attn_ok(self.dtype, H, I, self.nh, N) and O.attn_supported(self.dtype, N, N, True)
            st.append(Ctx(c=c, fmt=fmt, nl=len(c.layers), j=len(c.layers) - 1, M=lc.sa.Bn * N, N=N, dP=dP, d_top=d_top, dqkv=None, dao=None, dctx=None,
                          attn_due=None, dx0=None, can=can))
        d = self.drop            # (the LayerNorm backward of each stack's last output norm runs inside the first chain: kt = 0)
        everywhere = O.RBW_ATTN_MODE == 2
        done_blocks = 0

        def lins(s, j):
            lp = s.fmt.format(j)
            return (lp, self.lin(lp + "intermediate.dense.weight"), self.lin(lp + "output.dense.weight"), self.lin(lp + "attention.output.dense.weight"),
                    self.lin(lp + "attention.self.query.weight", rows=3 * H, cols=H))

        def chain_seg(s, j):
            """the per-token chain of block j of stack s: segment fields + its output buffers"""
            lp, f1, f2, o, _ = lins(s, j)
            lc = s.c.layers[j]
            sa, ffn, M = lc.sa, lc.ffn, s.M
            n2, n1 = self.ln(lp + "output.LayerNorm"), self.ln(lp + "attention.output.LayerNorm")
            out = Ctx(dz=self.new(M, I), daod=self.new(M, H), dao=self.new(M, H), dctx=self.new(M, H), dfo=self.new(M, H), dfod=self.new(M, H))
            seg = dict(M=M, y2=ffn.out, rstd2=ffn.rstd, g2=n2.g, b2=n2.b, z=ffn.z, W2T=f2.WTf, W1T=f1.WTf, y1=sa.a, rstd1=sa.rstd_a,
                       g1=n1.g, b1=n1.b, dg1=n1.dg, db1=n1.db, WoT=o.WTf, dz=out.dz, daod=out.daod, dao=out.dao, dctx=out.dctx,
                       dfo=out.dfo, dfod=out.dfod, dg2=n2.dg, db2=n2.db,
                       site_out=ffn.hdrop[2] if ffn.hdrop else 0, site_ao=sa.hdrop[2] if sa.hdrop else 0)
            return seg, out, 2.0 * ffn.rows * (2 * H * I + H * H)

        def chain_dw(s, j, out):
            lp, f1, f2, o, _ = lins(s, j)
            lc = s.c.layers[j]
            O.linear_dw(out.dfod, lc.ffn.g, f2.dW, f2.db, s.M, flop_rows=lc.ffn.rows)
            O.linear_dw(out.dz, lc.ffn.a, f1.dW, f1.db, s.M, flop_rows=lc.ffn.rows)
            O.linear_dw(out.daod, lc.sa.ctx, o.dW, o.db, s.M, flop_rows=lc.sa.rows)

        def tail_fields(s, j):
            """the tail of block j + 1 in front of block j's chain: from the plain top gradient, or from dQKV in memory"""
            if s.d_top is not None:
                t = dict(dqkv_n=s.d_top, kt=0, WqkvT_n=lins(s, j)[2].WTf, dao_n=s.d_top)
                s.d_top = None
                return t, 0.0
            return dict(dqkv_n=s.dqkv, kt=12, WqkvT_n=lins(s, j + 1)[4].WTf, dao_n=s.dao), 2.0 * s.c.layers[j].ffn.rows * 3 * H * H

        def attn_fields(s, ja, out):
            """the attention backward of block ja inside a launch (mode 1 / 2): its saved tensors, the d_ctx / d_ao rows the previous launch wrote"""
            sa1 = s.c.layers[ja].sa
            out.dqkv = self.new(s.M, 3 * H)
            return dict(N=sa1.N, ldp=sa1.ldp, qkv_a=sa1.qkv, P_a=sa1.Ppre, o_a=sa1.ctx, dctx_a=s.dctx, dP_init=s.dP if ja == s.nl - 1 else None,
                        dqkv_out=out.dqkv, site_attn=sa1.adrop[2] if sa1.adrop else 0, WqkvT_n=lins(s, ja)[4].WTf, dao_n=s.dao), \
                2.0 * sa1.rows * 3 * H * H + 8.0 * sa1.aflops

        while any(s.dx0 is None for s in st):
            act = [s for s in st if s.dx0 is None]
            jnow = {id(s): s.j for s in act}

            def defer(s):
                """will block s.j's attention backward run INSIDE this stack's next launch (instead of as a launch of its own right after this chain)?"""
                if O.RBW_ATTN_MODE == 0 or not s.can:
                    return False
                if everywhere:
                    return True
                alone_next = all(x is s or x.dx0 is not None or (jnow[id(x)] == 0 and x.attn_due is None) for x in st)      # the others finish in this step
                return s.N > 48 and alone_next
            segs, work = [], []
            for s in act:
                j = s.j
                if s.attn_due is not None:
                    # [attention backward of block attn_due = j + 1] + [chain of block j], or at the bottom [attention backward of block 0] + dx0
                    out = Ctx()
                    af, fl = attn_fields(s, s.attn_due, out)
                    if j >= 0:
                        seg, o2, fl2 = chain_seg(s, j)
                        o2.dqkv = out.dqkv
                        out, fl = o2, fl + fl2
                        seg.update(mode=1, **af)
                    else:
                        out.dx0 = self.new(s.M, H)
                        seg = dict(M=s.M, mode=2, dfo=out.dx0, **af)
                    seg["flops"] = fl
                    work.append((s, "inside", j, out))
                else:
                    # chain of block j with its tail from memory
                    seg, out, fl = chain_seg(s, j)
                    tf, fl2 = tail_fields(s, j)
                    seg.update(tf)
                    seg["flops"] = fl + fl2
                    work.append((s, "chain", j, out))
                segs.append(seg)
            O.rowbwd(segs, d[0] if d else None, d[1] if d else 0.0, p_attn=d[2] if d else 0.0, scale=1.0 / math.sqrt(HD))
            sep = []         # chains whose attention backward follows NOW as a launch of its own (shared by the stacks: one grouped launch)
            for s, kind, j, out in work:
                if kind == "inside":
                    ja = s.attn_due
                    ql = lins(s, ja)[4]
                    O.linear_dw(out.dqkv, s.c.layers[ja].sa.x, ql.dW, ql.db, s.M, flop_rows=s.c.layers[ja].sa.rows)      # block ja is complete
                    if j >= 0:
                        chain_dw(s, j, out)
                        s.dctx, s.dao, s.attn_due, s.j = out.dctx, out.dao, j, j - 1
                    else:
                        s.dx0, s.attn_due = out.dx0, None
                else:
                    chain_dw(s, j, out)
                    s.dctx, s.dao = out.dctx, out.dao
                    if defer(s):
                        s.attn_due, s.j = j, j - 1
                    else:
                        sep.append((s, j, out))
            if sep:
                grp = L.group() if len(sep) > 1 else None
                if grp is not None:
                    grp.__enter__()
                try:
                    for s, j, out in sep:
                        sa = s.c.layers[j].sa
                        out.dqkv = self.new(s.M, 3 * H)
                        self._attn_bwd(sa.Ppre, sa.ldp, out.dctx, sa.qkv, 3 * H, sa.qkv[:, H:], sa.qkv[:, 2 * H:], 3 * H,
                                       out.dqkv, 3 * H, out.dqkv[:, H:], out.dqkv[:, 2 * H:], 3 * H, sa.Bn, sa.N, sa.N, None, None,
                                       s.dP if j == s.nl - 1 else None, sa.aflops, sa.adrop, sa.P if sa.adrop else None)
                finally:
                    if grp is not None:
                        grp.__exit__(None, None, None)
                for s, j, out in sep:
                    ql = lins(s, j)[4]
                    O.linear_dw(out.dqkv, s.c.layers[j].sa.x, ql.dW, ql.db, s.M, flop_rows=s.c.layers[j].sa.rows)
                    s.dqkv = out.dqkv
                    if j == 0:
                        s.dx0 = O.linear_dx(out.dqkv, ql.W, s.M, residual=out.dao, flop_rows=s.c.layers[0].sa.rows)
                    s.j = j - 1
            # blocks complete (all four weight gradients queued) in EVERY stack: counted from the top
            k = min((10 ** 6 if s.dx0 is not None else s.nl - 1 - (s.attn_due if s.attn_due is not None else s.j)) for s in st)      # (attention pending: block attn_due is not complete yet)
            while done_blocks < min(k, max(s.nl for s in st)):
                done_blocks += 1
                if on_iter is not None:
                    on_iter(done_blocks)
        return [s.dx0 for s in st]

    def encoders_bwd(self, ct, cp, plan, d_txt, dP_txt, d_pano, d_fused, dP_pano, on_iter=None):
        """text + panorama encoders' backward together: the two stacks on the row-block kernel, then the embedding backwards"""
        self._pano_head_bwd(cp, d_pano, d_fused)
        p = self.p
        dt, dp = self.self_stacks_bwd([(ct, p + "lang_encoder.layer.{}.", d_txt, dP_txt),
                                       (cp, p + "img_embeddings.pano_encoder.layer.{}.", d_pano, dP_pano)], on_iter=on_iter)
        O.join_dw_early()         # (an early weight-gradient flush may hold the tied decoder's dW: the embedding backward adds into the same table)
        ll = self.lin(p + "img_embeddings.loc_linear.weight")
        if O.embed_in_bwd_ok(self.H, ll.K):          # both embedding backwards -- three LayerNorm backwards of the panorama stage + the text one -- in ONE launch
            return self._embeds_bwd(ct, cp, plan, dt, dp)
        # the two embedding LayerNorm backwards (text; panorama sum) are independent: one paired launch
        from . import lib as _L
        with _L.group():
            self._text_emb_bwd(ct, plan, dt)
            dsum = self._pano_emb_bwd(cp, plan, dp, stage=0)
        self._pano_emb_bwd(cp, plan, dp, stage=1, dsum=dsum)

    def _embeds_bwd(self, ct, cp, plan, dt, dp):
        """_text_emb_bwd + _pano_emb_bwd as one launch (csrc/rowops.hip embed_in_bwd_kernel) + the image projection's (deferred) weight gradient"""
        p, H = self.p + "img_embeddings.", self.H
        M = cp.Np * cp.V
        n1, n2, n3 = self.ln(p + "img_layer_norm"), self.ln(p + "loc_layer_norm"), self.ln(p + "layer_norm")
        ll, il = self.lin(p + "loc_linear.weight"), self.lin(p + "img_linear.weight")
        nt = self.ln(self.p + "embeddings.LayerNorm")
        dP0 = self.new(M, H)
        tok_g = self.S.g(self.p + "embeddings.token_type_embeddings.weight")
        O.embed_in_bwd(H, dict(M=M, Kin=ll.K, dy=dp, drop_dy=cp.edrop, X0=cp.X0, rstd3=cp.rstd_x0, g3=n3.g, b3=n3.b, dg3=n3.dg, db3=n3.db,
                               nav_idx=plan["nav_types"], d_nav=self.S.g(p + "nav_type_embedding.weight"), d_tok=tok_g,
                               A1=cp.A1, rstd1=cp.rstd_a1, g1=n1.g, b1=n1.b, dg1=n1.dg, db1=n1.db, dP0=dP0,
                               A2=cp.A2, rstd2=cp.rstd_a2, g2=n2.g, b2=n2.b, dg2=n2.dg, db2=n2.db, loc=cp.loc, dW=ll.dW, dbl=ll.db),
                       dict(M=ct.B * ct.L, dy=dt, y=ct.E, gamma=nt.g, beta=nt.b, rstd=ct.rstd_e, dx=None, dgamma=nt.dg, dbeta=nt.db, drop_dy=ct.edrop,
                            dtabs=((plan["txt_ids"], 0, 0, self.S.g(self.p + "embeddings.word_embeddings.weight"), 0),
                                   (None, ct.L, 2, self.S.g(self.p + "embeddings.position_embeddings.weight"), 0),
                                   (None, 0, 0, tok_g, 0)), hot0=0))
        O.linear_dw(dP0, cp.feats, il.dW, il.db, M)

    def text_bwd(self, c, plan, d_out, dP_init=None):
        p, H = self.p, self.H
        if self.rbw_ok() and c.layers and hasattr(c.layers[0], "sa") and self.enc_ok(c.L, len(c.layers)):
            return self._text_emb_bwd(c, plan, self.self_stacks_bwd([(c, p + "lang_encoder.layer.{}.", d_out, dP_init)])[0])
        d = d_out
        nl = self.cfg.num_l_layers
        for i in reversed(range(nl)):
            prev = self._out_ln_desc(f"{p}lang_encoder.layer.{i - 1}.", c.layers[i - 1]) if i > 0 else None
            d = self.self_layer_bwd(f"{p}lang_encoder.layer.{i}.", c.layers[i], d, dP_init if i == nl - 1 else None, fuse_in=prev)
        return self._text_emb_bwd(c, plan, d)

    def _text_emb_bwd(self, c, plan, d):
        p, H = self.p, self.H
        M = c.B * c.L
        n = self.ln(p + "embeddings.LayerNorm")
        O.ln_bwd(M, H, d, y=c.E, gamma=n.g, beta=n.b, rstd=c.rstd_e, dx=None, dgamma=n.dg, dbeta=n.db, drop_dy=c.edrop,
                 dtabs=((plan["txt_ids"], 0, 0, self.S.g(p + "embeddings.word_embeddings.weight"), 0),
                        (None, c.L, 2, self.S.g(p + "embeddings.position_embeddings.weight"), 0),
                        (None, 0, 0, self.S.g(p + "embeddings.token_type_embeddings.weight"), 0)),
                 hot0=0)          # id 0 pads every instruction (pretrain_src/data/tasks.py:116 pad_sequence(..., padding_value=0)); a pure performance hint

    # ---- panorama encoder ----------------------------------------------------------------------
    def embeds_fwd(self, plan, feats, loc):
        """both input embeddings with ONE launch behind the image projection (csrc/rowops.hip embed_in_fwd_kernel): the panorama stage's three
        LayerNorms + the text embedding's gathers / LayerNorm / dropout -- four launches in front of the whole-encoder launch before.  Returns
        (text ctx, panorama ctx) as text_fwd / pano_fwd build them up to their embeddings (bit-identical tensors); pass them on as `c=`."""
        p, H = self.p + "img_embeddings.", self.H
        Np, V = plan["Np"], plan["V"]
        M = Np * V
        ct, Mt, kw = self._text_embed_args(plan)
        c = Ctx(Np=Np, V=V, feats=feats, loc=loc)
        il = self.lin(p + "img_linear.weight")
        P0 = O.linear_fwd(feats, il.W, il.b, M)
        n1, n2, n3, ll = self.ln(p + "img_layer_norm"), self.ln(p + "loc_layer_norm"), self.ln(p + "layer_norm"), self.lin(p + "loc_linear.weight")
        c.A1, c.rstd_a1 = self.new(M, H), self.new(M, dtype=torch.float32)
        c.A2, c.rstd_a2 = self.new(M, H), self.new(M, dtype=torch.float32)
        c.X0, c.rstd_x0 = self.new(M, H), self.new(M, dtype=torch.float32)
        c.edrop = self._dh(p + "dropout")
        c.X0d = self.new(M, H) if c.edrop else None
        O.embed_in_fwd(H, dict(M=M, Kin=ll.K, eps=self.eps, P0=P0, g1=n1.g, b1=n1.b, A1=c.A1, rstd1=c.rstd_a1, loc=loc, W=ll.Wm, b=ll.b,
                               g2=n2.g, b2=n2.b, A2=c.A2, rstd2=c.rstd_a2, nav_tab=self.S.w(p + "nav_type_embedding.weight"),
                               nav_idx=plan["nav_types"], tok_tab=self.S.w(self.p + "embeddings.token_type_embeddings.weight"),
                               g3=n3.g, b3=n3.b, X0=c.X0, rstd3=c.rstd_x0, X0d=c.X0d, drop=c.edrop),
                       dict(M=Mt, out=ct.E, **kw))
        return ct, c

    def embed_in_ok(self):
        return O.EMBED_IN and self.H in (128, 256, 384, 768)

    def pano_fwd(self, plan, feats, loc, defer=False, c=None):
        """feats [Np*V, D] compute dtype; loc [Np*V, 7] fp32.  defer: as text_fwd; c: the embeddings are done already (`embeds_fwd`)."""
        p, H = self.p + "img_embeddings.", self.H
        Np, V = plan["Np"], plan["V"]
        M = Np * V
        if c is None:
            c = Ctx(Np=Np, V=V, feats=feats, loc=loc)
            il = self.lin(p + "img_linear.weight")
            from . import lib as _L
            with _L.solo():       # (not offered to a lockstep partner: the text encoder's first groupable launch is its QKV projection, as is our next one)
                P0 = O.linear_fwd(feats, il.W, il.b, M)
            n1 = self.ln(p + "img_layer_norm")
            c.A1, c.rstd_a1 = self.new(M, H), self.new(M, dtype=torch.float32)
            O.ln_fwd(M, H, c.A1, in0=P0, gamma=n1.g, beta=n1.b, eps=self.eps, rstd=c.rstd_a1)
            ll, n2 = self.lin(p + "loc_linear.weight"), self.ln(p + "loc_layer_norm")
            c.A2, c.rstd_a2 = self.new(M, H), self.new(M, dtype=torch.float32)
            O.smallk_ln_fwd(M, H, ll.K, loc, ll.Wm, ll.b, n2.g, n2.b, self.eps, c.A2, c.rstd_a2)
            n3 = self.ln(p + "layer_norm")
            c.X0, c.rstd_x0 = self.new(M, H), self.new(M, dtype=torch.float32)
            c.edrop = self._dh(p + "dropout")
            c.X0d = self.new(M, H) if c.edrop else None
            with _L.solo():       # (the image LayerNorm above pairs with the text embedding's; this one has no partner in the text segment)
                O.ln_fwd(M, H, c.X0, in0=c.A1, in1=c.A2,
                         tabs=((self.S.w(p + "nav_type_embedding.weight"), plan["nav_types"], 0, 0),
                               (self.S.w(self.p + "embeddings.token_type_embeddings.weight"), None, 0, 0), None),
                         gamma=n3.g, beta=n3.b, eps=self.eps, rstd=c.rstd_x0, drop_out=c.edrop, out_drop=c.X0d)
        x = c.X0d if c.edrop else c.X0
        c.layers = []
        af = float(Np) * V * V * HD * self.nh
        nl, qkv = self.cfg.num_pano_layers, None
        c.plan = plan
        if self.enc_ok(V, nl):
            c.pending = (p + "pano_encoder.layer.{}.", nl, x, Np, V, plan["pano_mask"], M, af)
            if defer:
                return c
            sg, c.layers = self._enc_segment(*c.pending)
            c.pending = None
            self._enc_launch([sg])
            return self._pano_tail(c, plan)
        for i in range(nl):
            lc = self.self_layer_fwd(f"{p}pano_encoder.layer.{i}.", x, Np, V, plan["pano_mask"], M, af, qkv=qkv,
                                     next_lp=f"{p}pano_encoder.layer.{i + 1}." if i + 1 < nl else None)
            c.layers.append(lc)
            x, qkv = lc.out, lc.next_qkv
        return self._pano_tail(c, plan)

    def _pano_tail(self, c, plan):
        p, H = self.p + "img_embeddings.", self.H
        Np, V = c.Np, c.V
        c.out, c.P, c.ldp = c.layers[-1].out, c.layers[-1].P, c.layers[-1].ldp
        c.img_attn = self.new(Np, V, c.ldp, dtype=torch.float32)
        c.fused = self.new(Np, H)
        c.fprobs = self.new(Np, V, dtype=torch.float32)
        hm = dict(P=c.P, nh=self.nh, inner=V * c.ldp, pmean=c.img_attn) if c.P.is_contiguous() else {}      # head-mean of the attention map: same launch
        if not hm:
            O.head_mean_fwd(c.P, c.img_attn, Np, self.nh, V * c.ldp)
        if cfg_get(self.cfg, "adaptive_pano_fusion"):
            fl = self.lin(p + "pano_fuse_linear.weight")
            O.pano_fuse_fwd(c.out, plan["view_lens"], fl.Wm, fl.b, c.fused, c.fprobs, Np, V, H, **hm)
        else:
            # masked mean over the valid views (adaptive_pano_fusion=false, r2r_magic_model_config.json:57) = the attention pooling
            # with a zero scoring vector: softmax of equal scores over the unmasked views is 1/n each
            zw, zb = self._zero_fuse()
            O.pano_fuse_fwd(c.out, plan["view_lens"], zw, zb, c.fused, c.fprobs, Np, V, H, **hm)
        return c

    def _zero_fuse(self):
        z = self._cache.get("zero_fuse")
        if z is None:
            z = self._cache["zero_fuse"] = (self.zeros(self.H, dtype=torch.float32), self.zeros(1, dtype=torch.float32))
        return z

    def pano_bwd(self, c, plan, d_pano, d_fused, dP_init=None):
        p = self.p + "img_embeddings."
        self._pano_head_bwd(c, d_pano, d_fused)
        if self.rbw_ok() and self.enc_ok(c.V, len(c.layers)):
            return self._pano_emb_bwd(c, plan, self.self_stacks_bwd([(c, p + "pano_encoder.layer.{}.", d_pano, dP_init)])[0])
        d = d_pano
        nl = self.cfg.num_pano_layers
        for i in reversed(range(nl)):
            prev = self._out_ln_desc(f"{p}pano_encoder.layer.{i - 1}.", c.layers[i - 1]) if i > 0 else None
            d = self.self_layer_bwd(f"{p}pano_encoder.layer.{i}.", c.layers[i], d, dP_init if i == nl - 1 else None, fuse_in=prev)
        return self._pano_emb_bwd(c, plan, d)

    def _pano_head_bwd(self, c, d_pano, d_fused):
        """fused-embedding (attention pooling / masked mean) backward: accumulates into d_pano"""
        p, H = self.p + "img_embeddings.", self.H
        Np, V = c.Np, c.V
        if d_fused is not None:
            if cfg_get(self.cfg, "adaptive_pano_fusion"):
                fl = self.lin(p + "pano_fuse_linear.weight")
                O.pano_fuse_bwd(c.out, c.fprobs, fl.Wm, d_fused, d_pano, fl.dW, fl.db, Np, V, H)
            else:       # masked mean: no scoring parameters (their would-be gradients land in a scratch vector)
                zw, zb = self._zero_fuse()
                O.pano_fuse_bwd(c.out, c.fprobs, zw, d_fused, d_pano, self.new(H, dtype=torch.float32), self.new(1, dtype=torch.float32), Np, V, H)

    def _pano_emb_bwd(self, c, plan, d, stage=None, dsum=None):
        """stage None: everything; 0: the sum LayerNorm's backward only (returns its dx; groupable with the text embedding's); 1: the rest"""
        p, H = self.p + "img_embeddings.", self.H
        Np, V = c.Np, c.V
        M = Np * V
        if stage != 1:
            n3 = self.ln(p + "layer_norm")
            dsum = self.new(M, H)
            O.ln_bwd(M, H, d, y=c.X0, gamma=n3.g, beta=n3.b, rstd=c.rstd_x0, dx=dsum, dgamma=n3.dg, dbeta=n3.db, drop_dy=c.edrop,
                     dtabs=((plan["nav_types"], 0, 0, self.S.g(p + "nav_type_embedding.weight"), 1),
                            (None, 0, 0, self.S.g(self.p + "embeddings.token_type_embeddings.weight"), 0), None))
            if stage == 0:
                return dsum
        n1 = self.ln(p + "img_layer_norm")
        dP0 = self.new(M, H)
        O.ln_bwd(M, H, dsum, y=c.A1, gamma=n1.g, beta=n1.b, rstd=c.rstd_a1, dx=dP0, dgamma=n1.dg, dbeta=n1.db)
        il = self.lin(p + "img_linear.weight")
        O.linear_dw(dP0, c.feats, il.dW, il.db, M)
        ll, n2 = self.lin(p + "loc_linear.weight"), self.ln(p + "loc_layer_norm")
        O.smallk_ln_bwd(M, H, ll.K, c.loc, dsum, c.A2, n2.g, n2.b, c.rstd_a2, ll.dW, ll.db, n2.dg, n2.db)

    # ---- map / viewpoint inputs ----------------------------------------------------------------
    def gmap_in_fwd(self, plan, pano, gmap_pos_fts, gimg=None):
        """gimg given (nav mode: node embeddings kept by the agent's GraphMap) or aggregated from the panoramas."""
        g, H = self.p + "global_encoder.", self.H
        B, K = plan["B"], plan["K"]
        M = B * K
        c = Ctx(pos=gmap_pos_fts)
        if gimg is None:
            gimg = self.new(M, H)
            O.csr_gather(pano.out, *plan["gmap_from_embed"], gimg, M, H)
            O.csr_gather(pano.fused, *plan["gmap_from_fused"], gimg, M, H, accumulate=True)
        pl, pn = self.lin(g + "gmap_pos_embeddings.0.weight"), self.ln(g + "gmap_pos_embeddings.1")
        c.A, c.rstd = self.new(M, H), self.new(M, dtype=torch.float32)
        O.smallk_ln_fwd(M, H, pl.K, gmap_pos_fts, pl.Wm, pl.b, pn.g, pn.b, self.eps, c.A, c.rstd)
        c.out = self.new(M, H)
        O.ln_fwd(M, H, c.out, in0=gimg, in1=c.A, tabs=((self.S.w(g + "gmap_step_embeddings.weight"), plan["gmap_step_ids"], 0, 0), None, None),
                 do_ln=False)
        return c

    def nodes_in_fwd(self, plan, pano, gmap_pos_fts=None, vp_pos_fts=None, gimg=None, vimg=None):
        """gmap_in_fwd and / or vp_in_fwd as ONE launch (magic_node_in_fwd): gathers + position embedding + step embedding per encoder.
        Returns (gmap ctx or None, vp ctx or None) in the shape the two per-op functions build (bit-identical tensors)."""
        H = self.H
        probs, outs = [], [None, None]
        if gmap_pos_fts is not None:
            g = self.p + "global_encoder."
            M = plan["B"] * plan["K"]
            pl, pn = self.lin(g + "gmap_pos_embeddings.0.weight"), self.ln(g + "gmap_pos_embeddings.1")
            c = Ctx(pos=gmap_pos_fts, A=self.new(M, H), rstd=self.new(M, dtype=torch.float32), out=self.new(M, H))
            q = dict(M=M, Kin=pl.K, x=gmap_pos_fts, W=pl.Wm, b=pl.b, gamma=pn.g, beta=pn.b, eps=self.eps, A=c.A, rstd=c.rstd, out=c.out,
                     tab=self.S.w(g + "gmap_step_embeddings.weight"), tab_idx=plan["gmap_step_ids"])
            if gimg is not None:
                q["add0"] = gimg
            else:
                q.update(src1=pano.out, csr1=plan["gmap_from_embed"], src2=pano.fused, csr2=plan["gmap_from_fused"])
            probs.append(q)
            outs[0] = c
        if vp_pos_fts is not None:
            l = self.p + "local_encoder."
            M = plan["B"] * plan["Vp"]
            pl, pn = self.lin(l + "vp_pos_embeddings.0.weight"), self.ln(l + "vp_pos_embeddings.1")
            c = Ctx(pos=vp_pos_fts, A=self.new(M, H), rstd=self.new(M, dtype=torch.float32), out=self.new(M, H))
            q = dict(M=M, Kin=pl.K, x=vp_pos_fts, W=pl.Wm, b=pl.b, gamma=pn.g, beta=pn.b, eps=self.eps, A=c.A, rstd=c.rstd, out=c.out)
            if vimg is not None:
                q["add0"] = vimg
            else:
                q.update(src1=pano.out, csr1=plan["vp_from_embed"])
            probs.append(q)
            outs[1] = c
        O.node_in_fwd(H, probs)
        return outs

    def nodes_in_bwd(self, plan, gin, d_gin, vin, d_vin, d_pano, d_fused):
        """gmap_in_bwd + vp_in_bwd (pretraining form: node embeddings aggregated from the panoramas) in three launches instead of six:
        step-embedding table gradient, both position-embedding backwards, all three transposed gathers (same rounding order as the per-op
        sequence vp -> d_pano, gmap -> d_pano, gmap -> d_fused)."""
        g, l, H = self.p + "global_encoder.", self.p + "local_encoder.", self.H
        Mg, Mv = plan["B"] * plan["K"], plan["B"] * plan["Vp"]
        O.ln_bwd(Mg, H, d_gin, dx=None, do_ln=False,
                 dtabs=((plan["gmap_step_ids"], 0, 0, self.S.g(g + "gmap_step_embeddings.weight"), 0), None, None),
                 hot0=0)          # step id 0 = every unvisited node and every padded map slot (tasks.py:142): reduced per workgroup
        gl_, gn = self.lin(g + "gmap_pos_embeddings.0.weight"), self.ln(g + "gmap_pos_embeddings.1")
        vl_, vn = self.lin(l + "vp_pos_embeddings.0.weight"), self.ln(l + "vp_pos_embeddings.1")
        O.smallk_ln_bwd_pair(H, [dict(M=Mg, Kin=gl_.K, x=gin.pos, dy=d_gin, y=gin.A, gamma=gn.g, beta=gn.b, rstd=gin.rstd, dW=gl_.dW, db=gl_.db,
                                      dgamma=gn.dg, dbeta=gn.db),
                                 dict(M=Mv, Kin=vl_.K, x=vin.pos, dy=d_vin, y=vin.A, gamma=vn.g, beta=vn.b, rstd=vin.rstd, dW=vl_.dW, db=vl_.db,
                                      dgamma=vn.dg, dbeta=vn.db)])
        O.csr_gather_multi(H, [dict(out=d_pano, n_out=plan["Np"] * plan["V"], accumulate=True, src1=d_vin, csr1=plan["vp_from_embed_T"],
                                    src2=d_gin, csr2=plan["gmap_from_embed_T"]),
                               dict(out=d_fused, n_out=plan["Np"], accumulate=True, src1=d_gin, csr1=plan["gmap_from_fused_T"])])

    def gmap_in_bwd(self, c, plan, d_in, d_pano, d_fused):
        g, H = self.p + "global_encoder.", self.H
        M = plan["B"] * plan["K"]
        O.ln_bwd(M, H, d_in, dx=None, do_ln=False,
                 dtabs=((plan["gmap_step_ids"], 0, 0, self.S.g(g + "gmap_step_embeddings.weight"), 0), None, None),
                 hot0=0)          # step id 0 = every unvisited node and every padded map slot (tasks.py:142): reduced per workgroup
        pl, pn = self.lin(g + "gmap_pos_embeddings.0.weight"), self.ln(g + "gmap_pos_embeddings.1")
        O.smallk_ln_bwd(M, H, pl.K, c.pos, d_in, c.A, pn.g, pn.b, c.rstd, pl.dW, pl.db, pn.dg, pn.db)
        if d_pano is not None:      # pretrain path: node embeddings were aggregated from the panoramas
            O.csr_gather(d_in, *plan["gmap_from_embed_T"], d_pano, plan["Np"] * plan["V"], H, accumulate=True)
            O.csr_gather(d_in, *plan["gmap_from_fused_T"], d_fused, plan["Np"], H, accumulate=True)

    def vp_in_fwd(self, plan, pano, vp_pos_fts, vimg=None):
        l, H = self.p + "local_encoder.", self.H
        M = plan["B"] * plan["Vp"]
        c = Ctx(pos=vp_pos_fts)
        if vimg is None:
            vimg = self.new(M, H)
            O.csr_gather(pano.out, *plan["vp_from_embed"], vimg, M, H)
        pl, pn = self.lin(l + "vp_pos_embeddings.0.weight"), self.ln(l + "vp_pos_embeddings.1")
        c.A, c.rstd = self.new(M, H), self.new(M, dtype=torch.float32)
        O.smallk_ln_fwd(M, H, pl.K, vp_pos_fts, pl.Wm, pl.b, pn.g, pn.b, self.eps, c.A, c.rstd)
        c.out = self.new(M, H)
        O.ln_fwd(M, H, c.out, in0=vimg, in1=c.A, do_ln=False)
        return c

    def vp_in_bwd(self, c, plan, d_in, d_pano):
        l, H = self.p + "local_encoder.", self.H
        M = plan["B"] * plan["Vp"]
        pl, pn = self.lin(l + "vp_pos_embeddings.0.weight"), self.ln(l + "vp_pos_embeddings.1")
        O.smallk_ln_bwd(M, H, pl.K, c.pos, d_in, c.A, pn.g, pn.b, c.rstd, pl.dW, pl.db, pn.dg, pn.db)
        if d_pano is not None:
            O.csr_gather(d_in, *plan["vp_from_embed_T"], d_pano, plan["Np"] * plan["V"], H, accumulate=True)

    # ---- cross-modal encoders ------------------------------------------------------------------
    def _sprel(self):
        g = self.p + "global_encoder."
        if not cfg_get(self.cfg, "graph_sprels"):
            return None, None
        w, b = self.S.master(g + "sprel_linear.weight"), self.S.master(g + "sprel_linear.bias")
        dw = (self.S.g(g + "sprel_linear.weight"), self.S.g(g + "sprel_linear.bias")) if self.train else None
        return (w, b), dw

    def cross_fwd(self, which, plan, x, Nq, qmask, qlens, qrows, ctx, Nk, kmask, klens, krows, dist=None, kv=None):
        enc = self.p + ("global_encoder." if which == "global" else "local_encoder.")
        B = plan["B"]
        sprel, _ = self._sprel() if (which == "global" and dist is not None) else (None, None)
        c = Ctx(which=which, layers=[], dist=dist)
        sf, cf = self._flops_attn(qlens, qlens), self._flops_attn(qlens, klens)
        nl, qkv = self.cfg.num_x_layers, None
        for i in range(nl):
            lc = self.cross_layer_fwd(f"{enc}encoder.crossattention.{i}.", x, B, Nq, qmask, dist, sprel, ctx, Nk, kmask, qrows, krows, sf, cf,
                                      qkv=qkv, next_lp=f"{enc}encoder.crossattention.{i + 1}." if i + 1 < nl else None,
                                      kv=None if kv is None else kv[i])
            c.layers.append(lc)
            x, qkv = lc.out, lc.next_qkv
        c.out, c.P, c.ldp = x, c.layers[-1].P, c.layers[-1].ldp
        return c

    # ---- both cross-modal encoders as one launch (csrc/encoder.hip, xencoder_fwd_kernel) -----------------------------------
    def xenc_ok(self, Nq, Nk):
        return O.xencoder_ok(self.dtype, self.H, self.I, self.nh, Nq, Nk, self.cfg.num_x_layers)

    def _xenc_segment(self, which, plan, x, Nq, qmask, qlens, qrows, ctx, Nk, kmask, klens, krows, dist=None):
        """allocate what cross_layer_bwd reads for every layer of one encoder; returns (segment dict for O.xencoder_fwd, cross Ctx)"""
        enc = self.p + ("global_encoder." if which == "global" else "local_encoder.")
        H, I, nh, B = self.H, self.I, self.nh, plan["B"]
        sprel, _ = self._sprel() if (which == "global" and dist is not None) else (None, None)
        sf, cf = self._flops_attn(qlens, qlens), self._flops_attn(qlens, klens)
        Mq, Mk, ldps, ldpc = B * Nq, B * Nk, rup(Nq), rup(Nk)
        cc = Ctx(which=which, layers=[], dist=dist)
        descs = []
        for i in range(self.cfg.num_x_layers):
            lp = f"{enc}encoder.crossattention.{i}."
            sa = Ctx(x=x, Bn=B, N=Nq, rows=qrows, aflops=sf, dist=dist, qkv=self.new(Mq, 3 * H), ldp=ldps,
                     adrop=self._da(lp + "attention.self.dropout"), hdrop=self._dh(lp + "attention.output.dropout"))
            sa.Ppre, sa.ctx, sa.a, sa.rstd_a = self.new(B, nh, Nq, ldps), self.new(Mq, H), self.new(Mq, H), self.new(Mq, dtype=torch.float32)
            sa.P = self.new(B, nh, Nq, ldps) if sa.adrop else sa.Ppre
            c = Ctx(Bn=B, Nq=Nq, Nk=Nk, ctx=ctx, rows=qrows, crow=krows, cflops=cf, next_qkv=None, kv_given=False, sa=sa,
                    q=self.new(Mq, H), kv=self.new(Mk, 2 * H), ldp=ldpc,
                    adrop=self._da(lp + "crossattention.self.dropout"), hdrop=self._dh(lp + "crossattention.output.dropout"))
            c.Ppre, c.cctx, c.c, c.rstd_c = self.new(B, nh, Nq, ldpc), self.new(Mq, H), self.new(Mq, H), self.new(Mq, dtype=torch.float32)
            c.P = self.new(B, nh, Nq, ldpc) if c.adrop else c.Ppre
            c.ffn = Ctx(a=c.c, M=Mq, rows=qrows, z=self.new(Mq, I), g=self.new(Mq, I), out=self.new(Mq, H), rstd=self.new(Mq, dtype=torch.float32),
                        hdrop=self._dh(lp + "output.dropout"))
            c.out = c.ffn.out
            ql = self.lin(lp + "attention.self.query.weight", rows=3 * H, cols=H)
            o, n1 = self.lin(lp + "attention.output.dense.weight"), self.ln(lp + "attention.output.LayerNorm")
            cq = self.lin(lp + "crossattention.self.query.weight")
            ckv = self.lin(lp + "crossattention.self.key.weight", lp + "crossattention.self.key.bias", rows=2 * H, cols=H)
            co, nc = self.lin(lp + "crossattention.output.dense.weight"), self.ln(lp + "crossattention.output.LayerNorm")
            f1, f2, n2 = self.lin(lp + "intermediate.dense.weight"), self.lin(lp + "output.dense.weight"), self.ln(lp + "output.LayerNorm")
            descs.append(dict(Wqkv=ql.Wf, bqkv=ql.b, Wo=o.Wf, bo=o.b, g1=n1.g, be1=n1.b, Wq=cq.Wf, bq=cq.b, Wkv=ckv.Wf, bkv=ckv.b,
                              Woc=co.Wf, boc=co.b, gc=nc.g, bec=nc.b, W1=f1.Wf, bi=f1.b, W2=f2.Wf, bo2=f2.b, g2=n2.g, be2=n2.b,
                              qkv=sa.qkv, P=sa.Ppre, Pd=sa.P if sa.adrop else None, ctx=sa.ctx, a=sa.a, rstd_a=sa.rstd_a,
                              q=c.q, kv=c.kv, Pc=c.Ppre, Pdc=c.P if c.adrop else None, cctx=c.cctx, c=c.c, rstd_c=c.rstd_c,
                              z=c.ffn.z, g=c.ffn.g, out=c.ffn.out, rstd_o=c.ffn.rstd,
                              site_attn=sa.adrop[2] if sa.adrop else 0, site_ao=sa.hdrop[2] if sa.hdrop else 0,
                              site_cattn=c.adrop[2] if c.adrop else 0, site_co=c.hdrop[2] if c.hdrop else 0,
                              site_out=c.ffn.hdrop[2] if c.ffn.hdrop else 0))
            cc.layers.append(c)
            x = c.out
        nl = self.cfg.num_x_layers
        flops = nl * (2.0 * qrows * (3 * H * H + 3 * H * H + 2 * H * I) + 2.0 * krows * 2 * H * H + 4.0 * sf + 4.0 * cf)
        cc.out, cc.P, cc.ldp = cc.layers[-1].out, cc.layers[-1].P, cc.layers[-1].ldp
        seg = dict(x=cc.layers[0].sa.x, cx=ctx, qmask=qmask, cmask=kmask, dist=dist, sprel_w=sprel[0] if sprel else None,
                   sprel_b=sprel[1] if sprel else None, nsamp=B, Nq=Nq, Nk=Nk, ldps=ldps, ldpc=ldpc, layers=descs, flops=flops)
        return seg, cc

    def cross_fwd_fused(self, specs):
        """specs: 1 or 2 argument tuples of cross_fwd (which, plan, x, Nq, qmask, qlens, qrows, ctx, Nk, kmask, klens, krows[, dist]);
        returns their cross contexts, computed by ONE launch"""
        segs, outs = [], []
        for sp in specs:
            sg, cc = self._xenc_segment(*sp)
            segs.append(sg)
            outs.append(cc)
        d = self.drop
        O.xencoder_fwd(segs, d[0] if d else None, d[2] if d else 0.0, d[1] if d else 0.0, self.eps, 1.0 / math.sqrt(HD))
        return outs

    def _grouped(self, fns, ok=True):
        """run the launches of `fns` (independent of each other) as ONE grouped launch when `ok`, else one after the other"""
        from . import lib as L
        if ok and len(fns) > 1:
            with L.group():
                for f in fns:
                    f()
        else:
            for f in fns:
                f()

    def cross_stacks_bwd(self, stacks):
        """Backward of 1 or 2 cross-modal encoders on the row-block kernel.  stacks: tuples (cross ctx, d_out = plain gradient wrt the
        encoder's output, d_ctx_acc = accumulator of the gradient wrt the context rows, dP_init for the top block's cross-attention map).
        Per block five launches shared by the two encoders -- full chain (tail of the block above + FFN + output-norm and
        cross-attention-output-norm backwards + cross output projection), cross-attention backward, key/value input gradient into the
        context accumulator, short chain (dQ Wq + residual -> self-attention-output-norm backward -> self output projection),
        self-attention backward -- instead of nine.  Returns the gradients wrt the encoders' inputs."""
        H, I = self.H, self.I
        st = []
        for c, d_top, d_acc, dP in stacks:
            enc = self.p + ("global_encoder." if c.which == "global" else "local_encoder.")
            _, dsprel = self._sprel() if (c.which == "global" and c.dist is not None) else (None, None)
            lc = c.layers[-1]
            st.append(Ctx(c=c, fmt=enc + "encoder.crossattention.{}.", j=len(c.layers) - 1, M=lc.Bn * lc.Nq, Mk=lc.Bn * lc.Nk, dP=dP, d_top=d_top,
                          d_acc=d_acc, dsprel=dsprel, pre=None, dqkv=None, dao=None, dx0=None, due=None))
        # round 6: a block's SELF-attention backward inside the next launch of its encoder (the full chain of the block below, or at the bottom the
        # input-gradient product) instead of as a launch of its own: csrc/encbwd.hip attn_tile_stage, one workgroup per 16-row tile of a sample's map /
        # viewpoint tokens, the graph-distance bias gradients included
        # MEASURED AND NOT KEPT as the default (MAGIC_RBW_ATTN=3 selects it; `profiles/micro/r06_ab_rbwattn_cross.txt`): 1.447-1.456 vs 1.431-1.438 ms/step -- a
        # 22-node map / 39-token viewpoint sample is two / three 16-row tiles, the last nearly empty, so the launch grows by a quarter while the
        # attention it absorbs is a 13 us launch -- and rowsum(P dP) taken as dO . O (bf16 O) leaves each dS row a rounding residue that the
        # sprel_linear gradients SUM over all rows (12 % off on `sprel_linear.weight` under dropout; the row-wise form's rows sum to zero exactly)
        inside = O.RBW_ATTN_MODE == 3 and all(O.rowbwd_attn_ok(self.dtype, H, I, self.nh, s.c.layers[-1].Nq) and
                                              O.attn_supported(self.dtype, s.c.layers[-1].Nq, s.c.layers[-1].Nq, True) for s in st)

        def attn_fields(s, lc, W, out_prev):
            sa = lc.sa
            out_prev.dqkv = self.new(s.M, 3 * H)
            f = dict(N=sa.N, ldp=sa.ldp, qkv_a=sa.qkv, P_a=sa.Ppre, o_a=sa.ctx, dctx_a=out_prev.dctx, dqkv_out=out_prev.dqkv,
                     site_attn=sa.adrop[2] if sa.adrop else 0, WqkvT_n=W.qkv.WTf, dao_n=out_prev.dao)
            if sa.dist is not None:
                f.update(dist=sa.dist, dsprel_w=s.dsprel[0], dsprel_b=s.dsprel[1])
            return f, 2.0 * sa.rows * 3 * H * H + 8.0 * sa.aflops

        d = self.drop
        seed, ph = (d[0] if d else None), (d[1] if d else 0.0)
        kvdx = []
        for j in reversed(range(len(st[0].c.layers))):
            segs, act = [], []
            for s in st:
                M = s.M
                lp, lc = s.fmt.format(j), s.c.layers[j]
                sa, ffn = lc.sa, lc.ffn
                W = Ctx(f1=self.lin(lp + "intermediate.dense.weight"), f2=self.lin(lp + "output.dense.weight"),
                        co=self.lin(lp + "crossattention.output.dense.weight"), cq=self.lin(lp + "crossattention.self.query.weight"),
                        ckv=self.lin(lp + "crossattention.self.key.weight", lp + "crossattention.self.key.bias", rows=2 * H, cols=H),
                        o=self.lin(lp + "attention.output.dense.weight"), qkv=self.lin(lp + "attention.self.query.weight", rows=3 * H, cols=H))
                n2, nc, n1 = self.ln(lp + "output.LayerNorm"), self.ln(lp + "crossattention.output.LayerNorm"), self.ln(lp + "attention.output.LayerNorm")
                out = Ctx(dz=self.new(M, I), dcod=self.new(M, H), dco=self.new(M, H), dcctx=self.new(M, H), dq=self.new(M, H),
                          dkv=self.new(s.Mk, 2 * H), dao=self.new(M, H), daod=self.new(M, H), dctx=self.new(M, H), dqkv=self.new(M, 3 * H))
                seg = dict(M=M, y2=ffn.out, rstd2=ffn.rstd, g2=n2.g, b2=n2.b, z=ffn.z, W2T=W.f2.WTf, W1T=W.f1.WTf, y1=lc.c, rstd1=lc.rstd_c,
                           g1=nc.g, b1=nc.b, dg1=nc.dg, db1=nc.db, WoT=W.co.WTf, dz=out.dz, daod=out.dcod, dao=out.dco, dctx=out.dcctx,
                           site_out=ffn.hdrop[2] if ffn.hdrop else 0, site_ao=lc.hdrop[2] if lc.hdrop else 0)
                flops = 2.0 * ffn.rows * (2 * H * I + H * H)
                if s.d_top is not None:          # top block: no product, the output norm's backward starts from the plain gradient
                    out.dfo, out.dfod = self.new(M, H), self.new(M, H)
                    seg.update(dqkv_n=s.d_top, kt=0, WqkvT_n=W.f2.WTf, dao_n=s.d_top, dfo=out.dfo, dfod=out.dfod, dg2=n2.dg, db2=n2.db)
                    s.d_top = None
                elif s.due is not None:          # the self-attention backward of block j + 1 runs in front of this chain (its short chain left d_ctx / d_ao)
                    lc1, W1, out1 = s.due
                    out.dfo, out.dfod = self.new(M, H), self.new(M, H)
                    af, fl = attn_fields(s, lc1, W1, out1)
                    seg.update(mode=1, dfo=out.dfo, dfod=out.dfod, dg2=n2.dg, db2=n2.db, **af)
                    flops += fl
                else:
                    qn = self.lin(s.fmt.format(j + 1) + "attention.self.query.weight", rows=3 * H, cols=H)
                    out.dfo, out.dfod = self.new(M, H), self.new(M, H)
                    seg.update(dqkv_n=s.dqkv, kt=12, WqkvT_n=qn.WTf, dao_n=s.dao, dfo=out.dfo, dfod=out.dfod, dg2=n2.dg, db2=n2.db)
                    flops += 2.0 * ffn.rows * 3 * H * H
                seg["flops"] = flops
                segs.append(seg)
                act.append((s, lc, W, n1, out))
            O.rowbwd(segs, seed, ph, p_attn=d[2] if d else 0.0, scale=1.0 / math.sqrt(HD))         # full chain (+ the deferred self-attention backward of the block above)
            for s in st:
                if s.due is not None:
                    lc1, W1, out1 = s.due
                    O.linear_dw(out1.dqkv, lc1.sa.x, W1.qkv.dW, W1.qkv.db, s.M, flop_rows=lc1.sa.rows)
                    s.due = None
            top = j == len(st[0].c.layers) - 1
            self._grouped([lambda s=s, lc=lc, out=out: self._attn_bwd(
                lc.Ppre, lc.ldp, out.dcctx, lc.q, H, lc.kv, lc.kv[:, H:], 2 * H, out.dq, H, out.dkv, out.dkv[:, H:], 2 * H, lc.Bn, lc.Nq, lc.Nk,
                None, None, s.dP if top else None, lc.cflops, lc.adrop, lc.P if lc.adrop else None) for s, lc, W, n1, out in act],
                ok=all(O.attn_supported(self.dtype, lc.Nq, lc.Nk, True) for _, lc, _, _, _ in act) and FUSED_ATTN)
            for s, lc, W, n1, out in act:          # key / value input gradients: independent of the rest of the chain, launched together below
                kvdx.append((s.d_acc, out.dkv, W.ckv.W, s.Mk, lc.crow))
            segs = []
            for s, lc, W, n1, out in act:
                sa = lc.sa
                segs.append(dict(M=s.M, kt=4, dqkv_n=out.dq, WqkvT_n=W.cq.WTf, dao_n=out.dco, y2=sa.a, rstd2=sa.rstd_a, g2=n1.g, b2=n1.b,
                                 dg2=n1.dg, db2=n1.db, WoT=W.o.WTf, dfo=out.dao, dfod=out.daod, dctx=out.dctx,
                                 site_out=sa.hdrop[2] if sa.hdrop else 0, flops=2.0 * sa.rows * 2 * H * H))
            O.rowbwd(segs, seed, ph)                                   # short chain
            if not inside:
              self._grouped([lambda s=s, lc=lc, out=out: self._attn_bwd(
                lc.sa.Ppre, lc.sa.ldp, out.dctx, lc.sa.qkv, 3 * H, lc.sa.qkv[:, H:], lc.sa.qkv[:, 2 * H:], 3 * H,
                out.dqkv, 3 * H, out.dqkv[:, H:], out.dqkv[:, 2 * H:], 3 * H, lc.Bn, lc.Nq, lc.Nq, lc.sa.dist, s.dsprel, None, lc.sa.aflops,
                lc.sa.adrop, lc.sa.P if lc.sa.adrop else None) for s, lc, W, n1, out in act],
                ok=all(O.attn_supported(self.dtype, lc.Nq, lc.Nq, True) for _, lc, _, _, _ in act) and FUSED_ATTN)
            for s, lc, W, n1, out in act:
                sa, ffn, M = lc.sa, lc.ffn, s.M
                O.linear_dw(out.dfod, ffn.g, W.f2.dW, W.f2.db, M, flop_rows=ffn.rows)
                O.linear_dw(out.dz, ffn.a, W.f1.dW, W.f1.db, M, flop_rows=ffn.rows)
                O.linear_dw(out.dcod, lc.cctx, W.co.dW, W.co.db, M, flop_rows=lc.rows)
                O.linear_dw(out.dq, sa.a, W.cq.dW, W.cq.db, M, flop_rows=lc.rows)
                O.linear_dw(out.dkv, lc.ctx, W.ckv.dW, W.ckv.db, s.Mk, flop_rows=lc.crow)
                O.linear_dw(out.daod, sa.ctx, W.o.dW, W.o.db, M, flop_rows=sa.rows)
                if inside:
                    s.due = (lc, W, out)          # (its dQKV -- and dWqkv -- come out of the next launch of this encoder)
                else:
                    O.linear_dw(out.dqkv, sa.x, W.qkv.dW, W.qkv.db, M, flop_rows=sa.rows)
                    s.dqkv, s.dao = out.dqkv, out.dao
            if j == 0 and inside:                # bottom: the self-attention backward of block 0 + the gradient wrt the encoders' inputs, one launch
                segs = []
                for s, lc, W, n1, out in act:
                    af, fl = attn_fields(s, lc, W, out)
                    s.dx0 = self.new(s.M, H)
                    segs.append(dict(M=s.M, mode=2, dfo=s.dx0, flops=fl, **af))
                O.rowbwd(segs, seed, ph, p_attn=d[2] if d else 0.0, scale=1.0 / math.sqrt(HD))
                for s, lc, W, n1, out in act:
                    O.linear_dw(out.dqkv, lc.sa.x, W.qkv.dW, W.qkv.db, s.M, flop_rows=lc.sa.rows)
                    s.due = None
            elif j == 0:
                def dx0(s, lc, W, out):
                    s.dx0 = O.linear_dx(out.dqkv, W.qkv.W, s.M, residual=out.dao, flop_rows=lc.sa.rows)
                self._grouped([lambda s=s, lc=lc, W=W, out=out: dx0(s, lc, W, out) for s, lc, W, n1, out in act])
        # every block's d_context = dKV Wkv (all with respect to the same context rows) as ONE grouped launch into separate buffers, then one
        # fold into the accumulator(s): 2 launches instead of one (paired) accumulating GEMM per block on the chain
        tmps = [self.new(Mk, H) for _, _, _, Mk, _ in kvdx]
        for i0 in range(0, len(kvdx), 8):
            self._grouped([lambda q=q, t=t: O.linear_dx(q[1], q[2], q[3], out=t, flop_rows=q[4]) for q, t in zip(kvdx[i0:i0 + 8], tmps[i0:i0 + 8])])
        accs = {}
        for q, t in zip(kvdx, tmps):
            accs.setdefault(id(q[0]), (q[0], []))[1].append(t)
        for acc, ts in accs.values():
            for i0 in range(0, len(ts), 8):
                O.add_n(acc, ts[i0:i0 + 8])
        return [s.dx0 for s in st]

    def cross_bwd(self, c, d_out, d_ctx_acc, dP_init=None, dkv=None, acc_kv=False):
        if dkv is None and self.rbw_ok() and not any(lc.kv_given for lc in c.layers):
            return self.cross_stacks_bwd([(c, d_out, d_ctx_acc, dP_init)])[0]
        enc = self.p + ("global_encoder." if c.which == "global" else "local_encoder.")
        _, dsprel = self._sprel() if (c.which == "global" and c.dist is not None) else (None, None)
        d = d_out
        nl = self.cfg.num_x_layers
        for i in reversed(range(nl)):
            prev = self._out_ln_desc(f"{enc}encoder.crossattention.{i - 1}.", c.layers[i - 1]) if i > 0 else None
            d = self.cross_layer_bwd(f"{enc}encoder.crossattention.{i}.", c.layers[i], d, d_ctx_acc, dsprel, dP_init if i == nl - 1 else None,
                                     fuse_in=prev, dkv_out=None if dkv is None else dkv[i], acc_kv=acc_kv)
        return d
