"""`GlocalTextPathCMTPreTraining` -- the pretraining model the reference imports from its withheld
`model.pretrain_goat` (pretrain_src/train_r2r_magic.py:40) -- rebuilt on the HIP engine.

Call surface kept: `from_pretrained(None, config=..., state_dict=...)`, `.train()/.eval()`,
`model(batch, task=..., compute_loss=...)`, `state_dict()` / `named_parameters()` with the checkpoint key
names of SURVEY App. A.4.  Return contract per task: train_r2r_magic.py:440-587 (validate_*).
With `compute_loss=True` the forward also evaluates the supervised loss and the in-model MAKD terms
(against `teacher_outputs`) and seeds the explicit backward; `loss.backward()` (or `model.backward()`)
then runs the hand-written backward and leaves the gradients in `param.grad` (views of one flat buffer).
"""
import os

import torch
import torch.nn as nn

from . import ops as O
from .config import cfg_get
from .ddp import refuse_torch_ddp
from .engine import Ctx, MagicNet, cls_specs, rup, trunk_specs
from .params import ParamStore
from .plan import build_plan, check_plan

MID_CUT = 2          # rounds of the shared text / panorama backward after which the data-parallel exchange cuts its middle bucket
LOCKSTEP = not os.environ.get("MAGIC_NO_LOCKSTEP")
LOCKSTEP_EAGER = bool(os.environ.get("MAGIC_LOCKSTEP_EAGER"))
DYN_TERMS = ("txt_emb", "txt_attn", "img_emb", "img_fused", "img_attn", "g_emb", "g_attn", "l_emb", "l_attn")      # plan["dyn"] slots
KD_SLOTS = ("txt_emb_loss", "txt_attn_loss", "img_emb_loss", "avg_img_emb_loss", "img_attn_loss",
            "global_emb_loss", "global_attn_loss", "local_emb_loss", "local_attn_loss", "predict_loss")


PRETRAIN_CAUSAL = ("back_txt", "back_img")       # the pretraining collates carry no front-door dictionaries


def pretrain_specs(cfg):
    H = cfg.hidden_size
    s = trunk_specs(cfg, "bert.")
    s += [("mlm_head.predictions.transform.dense.weight", (H, H), "normal"), ("mlm_head.predictions.transform.dense.bias", (H,), "zeros"),
          ("mlm_head.predictions.transform.LayerNorm.weight", (H,), "ones"), ("mlm_head.predictions.transform.LayerNorm.bias", (H,), "zeros"),
          ("mlm_head.predictions.bias", (cfg.vocab_size,), "zeros")]
    s += cls_specs("global_sap_head.", H) + cls_specs("local_sap_head.", H) + cls_specs("sap_fuse_linear.", H, 2 * H)
    for k in ("gmap", "vp", "fused", "txt"):
        s += [(f"cfp_heads.{k}.weight", (H, H), "normal"), (f"cfp_heads.{k}.bias", (H,), "zeros")]
    if "mrc" in (getattr(cfg, "pretrain_tasks", None) or ()):     # RegionClassification(H, image_prob_size), only when configured
        P = int(cfg_get(cfg, "image_prob_size"))
        s += [("image_classifier.net.0.weight", (H, H), "normal"), ("image_classifier.net.0.bias", (H,), "zeros"),
              ("image_classifier.net.2.weight", (H,), "ones"), ("image_classifier.net.2.bias", (H,), "zeros"),
              ("image_classifier.net.3.weight", (P, H), "normal"), ("image_classifier.net.3.bias", (P,), "zeros")]
    # back-door adjustment blocks (do_back_txt / do_back_img: r2r_magic_model_config.json:60-66, off in the shipped config; their inputs
    # instr_z_* / img_z_* come from the collates, pretrain_src/data/tasks.py:156-164, :441-449) -- the navigation model's blocks
    from .causal import causal_specs
    return s + causal_specs(cfg, "bert.", only=PRETRAIN_CAUSAL)


def _dropout_knobs(model, config):
    """Two nn.Dropout modules that compute nothing: they only HOLD the probabilities, so that the reference's
    `set_dropout(model, p)` (utils/misc.py:19-25: every nn.Dropout module's p := p) and train()/eval() keep working; the
    masks themselves are generated inside the HIP kernels (csrc/common.hpp)."""
    model.dropout = nn.Dropout(float(cfg_get(config, "hidden_dropout_prob")))
    model.attention_dropout = nn.Dropout(float(cfg_get(config, "attention_probs_dropout_prob")))


class _BackwardHook(torch.autograd.Function):
    """Lets `loss.backward()` of an unmodified training loop trigger the explicit HIP backward."""

    @staticmethod
    def forward(ctx, loss_value, anchor, model):
        ctx.model = model
        return loss_value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        ctx.model.backward()
        m = ctx.model
        if m.loss_scale is not None:   # (dynamic scale: the device word the seeds were multiplied by)
            m.store.grad.mul_(grad_out.to(torch.float32) * m.loss_scale[1])
        elif m.grad_scale != 1.0:      # hand autograd's contract back: .grad = grad_out x dLoss/dparam (grad_out: e.g. a GradScaler's factor)
            m.store.grad.mul_(grad_out.to(torch.float32) / m.grad_scale)
        from .trainer import auto_sync
        auto_sync(ctx.model)           # data parallel under an unmodified loop: average the flat gradient buffer here
        return None, None, None


MLM_TAIL_FUSED = os.environ.get("MAGIC_MLM_TAIL_FUSED", "1") != "0"
# K-splits of the vocabulary input gradient d_hm = dlogits Wemb (18 output tiles alone cannot fill 256 CUs): 32 splits.  Default (round 6): each split STORES its
# partial into its own slab (magic_gemm with splitk < 0) and the transform's LayerNorm backward adds the slabs in order (magic_ln_bwd_tail): reproducible, + 13-18 us on
# an mlm step.  MAGIC_MLM_DX_ATOMICS=1: the round-3 form, fp32 atomics into one accumulator -- whenever launch timing shifts ONE element of its bf16 cast flips between
# runs and every gradient below differs at a bf16 ulp (233 of 354 tensors: profiles/micro/r06_det_probe_partial128_b.txt), which is not a property to ship for 5 us a step.
MLM_DX_SPLITK = int(os.environ.get("MAGIC_MLM_DX_SPLITK", "32"))
MLM_DX_ATOMICS = os.environ.get("MAGIC_MLM_DX_ATOMICS", "0") != "0"


class GlocalTextPathCMTPreTraining(nn.Module):
    def __init__(self, config, device="cuda", compute_dtype=torch.bfloat16, seed=0):
        super().__init__()
        self.config = config
        self.device_ = torch.device(device)
        self.compute_dtype = compute_dtype
        trainable = getattr(config, "role", "student") != "teacher" or bool(getattr(config, "train_teacher", False))
        self.store = ParamStore(pretrain_specs(config), device, compute_dtype,
                                init_std=cfg_get(config, "initializer_range"), seed=seed, requires_grad=trainable)
        self.store.attach_to(self)
        self.net = MagicNet(config, self.store, "bert.")
        self.prefix, self.explicit_backward = "bert.", True
        if self.device_.type == "cuda":
            O.dw_counters(self.device_)            # the deterministic weight-gradient launch's counters: allocated outside any capture
        from .causal import build_blocks
        self.causal_blocks = build_blocks(self, only=PRETRAIN_CAUSAL)        # {} unless config.do_back_txt / do_back_img
        _dropout_knobs(self, config)
        self._anchor = torch.zeros(1, device=self.device_, requires_grad=True)
        self._ctx = None
        self._aux = torch.cuda.Stream(device=self.device_) if (self.device_.type == "cuda" and os.environ.get("MAGIC_PAR")) else None   # opt-in: measured slower under HIP-graph replay
        self._kd_idx = torch.tensor([0, 0, 1, 1, 1, 2, 2, 3, 3, 4], device=self.device_)
        self.keep_mlm_logits = False     # True: MLM CE gradient goes to its own buffer instead of overwriting the logits
        self.dropout_seed = None         # optional int32[2] device tensor: when set, forward() keys its dropout masks on it instead of drawing one
        # fp16 compute: every gradient SEED (the coefficient of each fused loss + gradient kernel) is multiplied by `grad_scale`, so the
        # activation gradients stay inside fp16's range (the reference's own fp16 option scales the loss the same way: GradScaler,
        # train_r2r_magic.py:370-371); the flat fp32 gradient buffer then holds grad_scale x the gradient and the optimizer divides
        # (PretrainStep folds 1 / grad_scale into the AdamW kernel's gradient pre-scale; under `loss.backward()` the hook below does).
        # A power of two: exact in every format.  Loss VALUES are never scaled.
        self.grad_scale = 4096.0 if compute_dtype == torch.float16 else 1.0
        # ... or DYNAMIC (enable_dynamic_loss_scale; trainer.PretrainStep turns it on for fp16): fp32[4] device state {S, 1 / S, clean steps in a
        # row, pending}; the loss kernels read S from the device (ops.seed_scale), the host coefficients then carry no scale
        self.loss_scale = None
        self.register_load_state_dict_post_hook(lambda m, k: setattr(m.store, "shadow_clean", False))

    def enable_dynamic_loss_scale(self, init=None):
        """amp.GradScaler semantics on the device (csrc/loss.hip step_rng_kernel + csrc/optim.hip adamw_kernel): returns the state tensor.
        init: the first scale (default: the static `grad_scale`; GradScaler's own default is 65536)"""
        if self.loss_scale is None:
            S = float(init if init is not None else self.grad_scale)
            self.loss_scale = torch.tensor([S, 1.0 / S, 0.0, 0.0], dtype=torch.float32, device=self.device_)
        return self.loss_scale

    # ---- HF-style constructor (train_r2r_magic.py:260-277) ------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, config=None, state_dict=None, **kw):
        model = cls(config, **kw)
        if state_dict is not None:
            own = model.state_dict()
            keep = {k: v for k, v in state_dict.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}
            model.load_state_dict(keep, strict=False)      # unmatched keys are silently ignored, as HF does
        return model

    def _par(self, fn_main, fn_aux):
        """run two independent segments concurrently: fn_aux on this model's auxiliary stream, fn_main on the current one"""
        if LOCKSTEP and self.device_.type == "cuda" and (torch.cuda.is_current_stream_capturing() or LOCKSTEP_EAGER):
            # paired launches (one kernel per two calls).  Only while a HIP graph is being captured: the two host threads
            # rendezvous on every call, which costs more than it saves when launches are issued eagerly (measured: 15 ms/step
            # eager with pairing, 6.5 without; replayed graphs: 3.4 with, 3.9 without)
            from . import lib as L
            return L.lockstep(fn_main, fn_aux)
        if self._aux is None:
            return fn_main(), fn_aux()
        cur = torch.cuda.current_stream()
        self._aux.wait_stream(cur)
        with torch.cuda.stream(self._aux):
            a = fn_aux()
        m = fn_main()
        cur.wait_stream(self._aux)
        return m, a

    def _arm_dropout(self):
        """model.train() (train_r2r_magic.py:358) turns the config's dropouts on (r2r_magic_model_config.json:2-3,6); the seed
        is drawn on the device by torch's graph-safe generator, so a replayed HIP graph gets fresh masks every step."""
        ph, pa = float(self.dropout.p), float(self.attention_dropout.p)
        if self.training and self.store.requires_grad and torch.is_grad_enabled() and (ph > 0 or pa > 0):
            seed = getattr(self, "dropout_seed", None)       # PretrainStep: a device word pair its step prologue refreshes every step (one launch with the MKRW draw)
            if seed is None:
                seed = torch.randint(0, 2 ** 31 - 1, (2,), dtype=torch.int32, device=self.device_)
            self.net.set_dropout(seed, ph, pa)
        else:
            self.net.set_dropout(None)

    def mark_params_dirty(self):
        self.store.shadow_clean = False

    # ---- helpers --------------------------------------------------------------------------------------
    def _dev(self, t, dtype=None):
        t = t.to(self.device_, non_blocking=True)
        return t if dtype is None or t.dtype == dtype else t.to(dtype)

    def _inputs(self, batch, plan):
        Np, V = plan["Np"], plan["V"]
        if batch.get("view_table") is not None:
            # index-only batch (host/feature_table.py): the view features are gathered on the device from
            # the packed HBM table in the reference's token order; nothing but 37 int32 per panorama crossed PCIe
            x = batch["view_table"].gather(batch["traj_vp_row"], batch["traj_view_order"]).reshape(Np * V, -1)
        else:
            x = self._dev(batch["traj_view_img_fts"]).reshape(Np * V, -1)
        if x.dtype != self.compute_dtype:
            x = O.cast_to(x.contiguous(), self.compute_dtype)
        return Ctx(feats=x,
                   loc=self._dev(batch["traj_loc_fts"], torch.float32).reshape(Np * V, -1).contiguous(),
                   gpos=self._dev(batch["gmap_pos_fts"], torch.float32).reshape(plan["B"] * plan["K"], -1).contiguous(),
                   vpos=self._dev(batch["vp_pos_fts"], torch.float32).reshape(plan["B"] * plan["Vp"], -1).contiguous(),
                   dist=self._dev(batch["gmap_pair_dists"], torch.float32).contiguous())

    def _cls(self, prefix, X, M, lda=None):
        """ClsPrediction forward: returns (Y=relu(linear), logit)"""
        n, H = self.net, self.net.H
        l1 = n.lin(prefix + "net.0.weight")
        Y = O.linear_fwd(X, l1.W, l1.b, M, epilogue=2, lda=lda)
        ln = n.ln(prefix + "net.2")
        l2 = n.lin(prefix + "net.3.weight")
        logit = n.new(M, dtype=torch.float32)
        O.lndot_fwd(Y, M, H, ln.g, ln.b, n.eps, l2.Wm, l2.b, logit)
        return Y, logit

    def _cls_tail(self, prefix, Y, M):
        """ClsPrediction after its first projection: LayerNorm + Linear(H, 1)"""
        n = self.net
        ln, l2 = n.ln(prefix + "net.2"), n.lin(prefix + "net.3.weight")
        logit = n.new(M, dtype=torch.float32)
        O.lndot_fwd(Y, M, n.H, ln.g, ln.b, n.eps, l2.Wm, l2.b, logit)
        return logit

    def _cls_bwd(self, prefix, X, Y, dlogit, M, d_acc, lda=None):
        n, H = self.net, self.net.H
        l1, ln, l2 = n.lin(prefix + "net.0.weight"), n.ln(prefix + "net.2"), n.lin(prefix + "net.3.weight")
        dZ = n.new(M, H)
        O.lndot_bwd(Y, M, H, ln.g, ln.b, n.eps, l2.Wm, dlogit, dZ, ln.dg, ln.db, l2.dW, l2.db)
        O.linear_dw(dZ, X, l1.dW, l1.db, M, ldb=lda)
        O.linear_dx(dZ, l1.W, M, out=d_acc, residual=d_acc, ldc=lda)

    # ---- forward ----------------------------------------------------------------------------------------
    def _causal_fwd(self, c, batch, compute_loss):
        """back-door adjustment of the text encoder's output (instruction z-dictionary) and of the panorama encoder's view embeddings (room-type
        image dictionary), where VLNBert('language' / 'panorama') applies them (host/model_nav.py): the blocks of host/causal.py
        are autograd modules, so each runs as a small autograd ISLAND inside the explicit step -- forward on a detached leaf here, its
        backward (`_causal_bwd`) turns the gradient of the adjusted tensor into the gradient of the encoder output and writes the block's
        parameter gradients into the store.  The fused panorama embedding stays the un-adjusted one, as in VLNBert('panorama')."""
        cz, plan, H = self.causal_blocks, c.plan, self.net.H
        need = {"instr_z_direction_features": "back_txt", "instr_z_landmark_features": "back_txt", "img_z_features": "back_img"}
        for k, blk in need.items():
            if batch.get(k) is not None and blk not in cz:
                from .causal import BLOCKS
                raise ValueError(f"batch carries {k!r} but config.{BLOCKS[blk]} is off -- the model has no '{blk}' block "
                                 "(pretrain_src/config/r2r_magic_model_config.json:60-66)")
        track = bool(compute_loss and self.store.requires_grad and torch.is_grad_enabled())
        c.islands = []

        def island(block, x2d, shape, z, pz):
            x0 = x2d.detach().view(shape)
            with torch.enable_grad() if track else torch.no_grad():
                if track:
                    x0.requires_grad_(True)
                y = block(x0, z, pz)
            if track:
                c.islands.append((x0, y))
            return y.detach().reshape(x2d.shape).contiguous()
        dv = lambda t: t.to(self.device_)
        if "back_txt" in cz and (batch.get("instr_z_direction_features") is not None or batch.get("instr_z_landmark_features") is not None):
            parts = [k for k in ("direction", "landmark") if batch.get(f"instr_z_{k}_features") is not None]      # direction rows first
            z = torch.cat([dv(batch[f"instr_z_{k}_features"]) for k in parts], 1)
            pz = torch.cat([dv(batch[f"instr_z_{k}_pzs"]) for k in parts], 1)
            c.txt_out = island(cz["back_txt"], c.txt.out, (plan["B"], plan["L"], H), z, pz)
            c.island_txt = len(c.islands) - 1 if track else None
        if "back_img" in cz and batch.get("img_z_features") is not None:
            out = island(cz["back_img"], c.pano.out, (plan["Np"], plan["V"], H), dv(batch["img_z_features"]), dv(batch["img_z_pzs"]))
            c.pano_adj = Ctx(**vars(c.pano))
            c.pano_adj.out = out
            c.island_pano = len(c.islands) - 1 if track else None

    def _causal_bwd(self, c):
        """gradient of the adjusted tensors -> gradient of the encoder outputs, through the islands of `_causal_fwd`"""
        for name, attr in (("island_txt", "d_txt"), ("island_pano", "d_pano")):
            i = getattr(c, name, None)
            if i is None:
                continue
            x0, y = c.islands[i]
            d = getattr(c, attr)
            torch.autograd.backward(y, d.view(y.shape).to(y.dtype))
            # a COPY: without dropout the island's add&norm hands the same tensor to both of its inputs, the deferred weight-gradient
            # GEMM of the block's output projection keeps reading it until the flush, and the encoder backward below accumulates in place
            setattr(c, attr, x0.grad.reshape(d.shape).to(d.dtype).clone())
            x0.grad = None

    def will_fuse_encoders(self, plan):
        """does a forward on `plan` run the text + panorama encoders as ONE whole-encoder launch (csrc/encoder.hip)?  The one predicate
        behind the model's own choice and the teacher stream's start gate (trainer.capture_split, stream_graph)."""
        n = self.net
        return bool(n.enc_ok(plan["L"], self.config.num_l_layers) and n.enc_ok(plan["V"], self.config.num_pano_layers))

    def forward(self, batch, task, compute_loss=True, teacher_outputs=None, rw=None, plan=None, return_outputs=False, inputs=None):
        n = self.net
        refuse_torch_ddp(self)
        self.store.sync_shadow()
        O.DEFER["queue"].clear(); O.DEFER["bytes"] = 0
        O.RBW_JOBS.clear()
        if self.store.requires_grad:
            O.defer_dw(False)         # a compute_loss forward that was never followed by backward() left deferral armed
        self._arm_dropout()
        plan = plan if plan is not None else build_plan(batch, task, self.device_)
        check_plan(plan, self.config)
        inp = inputs if inputs is not None else self._inputs(batch, plan)
        B, L, K, Vp, H = plan["B"], plan["L"], plan["K"], plan["Vp"], n.H
        c = Ctx(task=task, plan=plan, inp=inp)
        # the text and panorama encoders are independent: run them on two streams
        # both self-attention encoders as ONE launch (csrc/encoder.hip) when the shapes allow: embeddings first (paired), then the launch
        fuse = self.will_fuse_encoders(plan)
        if n.embed_in_ok():        # both input embeddings in one launch behind the image projection, then the layers
            ct, cp = n.embeds_fwd(plan, inp.feats, inp.loc)
            c.txt, c.pano = self._par(lambda: n.text_fwd(plan, defer=fuse, c=ct), lambda: n.pano_fwd(plan, inp.feats, inp.loc, defer=fuse, c=cp))
        else:
            c.txt, c.pano = self._par(lambda: n.text_fwd(plan, defer=fuse), lambda: n.pano_fwd(plan, inp.feats, inp.loc, defer=fuse))
        if fuse:
            n.encoders_fwd(c.txt, c.pano)
        c.txt_out, c.pano_adj = c.txt.out, c.pano
        if self.causal_blocks or any(batch.get(k) is not None for k in ("instr_z_direction_features", "instr_z_landmark_features", "img_z_features")):
            self._causal_fwd(c, batch, compute_loss)
        # map-token and viewpoint-token inputs of the cross-modal encoders: one launch for both (gathers + position / step embeddings)
        c.gin, c.vin = n.nodes_in_fwd(plan, c.pano_adj, inp.gpos if task != "mrc" else None, inp.vpos if task != "mlm" else None)
        tl, gl_, vl = plan["lens"]["txt"], plan["lens"]["gmap"], [Vp] * B
        o = dict(txt_embeds=c.txt_out, txt_attns=c.txt.P, pano_embeds=c.pano_adj.out, pano_fused_embeds=c.pano.fused,
                 img_attns=c.pano.img_attn, plan=plan, inputs=inp)
        if task == "mlm":
            l2v_args = ("global", plan, c.txt_out, L, plan["txt_mask"], tl, plan["txt_tokens"], c.gin.out, K, plan["gmap_mask"], gl_, plan["gmap_nodes"])
            c.l2v = n.cross_fwd_fused([l2v_args])[0] if n.xenc_ok(L, K) else n.cross_fwd(*l2v_args, dist=None)
            o["gmap_embeds"], o["gmap_attns"] = c.l2v.out, c.l2v.P
            nm = plan["n_mask"]
            c.hm_in = n.new(nm, H)
            O.csr_gather(c.l2v.out, *plan["mlm_rows"], c.hm_in, nm, H)
            t = n.lin("mlm_head.predictions.transform.dense.weight")
            c.tz = n.new(nm, H)
            tn = n.ln("mlm_head.predictions.transform.LayerNorm")
            c.hm, c.rstd_hm = n.new(nm, H), n.new(nm, dtype=torch.float32)
            if MLM_TAIL_FUSED and O.linear_ln_ok(H, H):      # dense -> gelu -> LayerNorm as one launch
                O.linear_act_ln(c.hm_in, t.W, t.b, nm, 1, c.tz, tn.g, tn.b, n.eps, c.hm, c.rstd_hm)
            else:
                c.tg = O.linear_fwd(c.hm_in, t.W, t.b, nm, epilogue=1, pre=c.tz)
                O.ln_fwd(nm, H, c.hm, in0=c.tg, gamma=tn.g, beta=tn.b, eps=n.eps, rstd=c.rstd_hm)
            Vv = self.config.vocab_size
            c.ldv = rup(Vv, 8)
            c.logits = n.new(nm, c.ldv)
            if c.ldv > Vv:
                c.logits[:, Vv:].zero_()
            O.linear_fwd(c.hm, self.store.w("bert.embeddings.word_embeddings.weight"), self.store.master("mlm_head.predictions.bias"),
                         nm, out=c.logits, ldc=c.ldv)
            o["predict"] = c.logits[:, :Vv]
        elif task == "mrc":
            # local branch only; RegionClassification on the masked views of the current viewpoint (validate_mrc :476-500)
            loc_args = ("local", plan, c.vin.out, Vp, plan["vp_mask"], vl, B * Vp, c.txt_out, L, plan["txt_mask"], tl, plan["txt_tokens"])
            c.loc = n.cross_fwd_fused([loc_args])[0] if n.xenc_ok(Vp, L) else n.cross_fwd(*loc_args)
            o.update(vp_embeds=c.loc.out, vp_attns=c.loc.P)
            nm = plan["n_mrc"]
            c.mx = n.new(nm, H)
            O.csr_gather(c.loc.out, *plan["mrc_rows"], c.mx, nm, H)
            l1, l2 = n.lin("image_classifier.net.0.weight"), n.lin("image_classifier.net.3.weight")
            ln = n.ln("image_classifier.net.2")
            c.mZ, c.m_rstd = n.new(nm, H), n.new(nm, dtype=torch.float32)
            if MLM_TAIL_FUSED and O.linear_ln_ok(H, H):      # Linear -> ReLU -> LayerNorm as one launch; mY keeps the PRE-activation (same relu' mask)
                c.mY = n.new(nm, H)
                O.linear_act_ln(c.mx, l1.W, l1.b, nm, 2, c.mY, ln.g, ln.b, n.eps, c.mZ, c.m_rstd)
            else:
                c.mY = O.linear_fwd(c.mx, l1.W, l1.b, nm, epilogue=2)
                O.ln_fwd(nm, H, c.mZ, in0=c.mY, gamma=ln.g, beta=ln.b, eps=n.eps, rstd=c.m_rstd)
            c.mlogits = O.linear_fwd(c.mZ, l2.W, l2.b, nm)
            o["predict"] = c.mlogits
        else:
            def _local():
                vin = c.vin
                return vin, n.cross_fwd("local", plan, vin.out, Vp, plan["vp_mask"], vl, B * Vp,
                                        c.txt_out, L, plan["txt_mask"], tl, plan["txt_tokens"])
            # global (map) and local (viewpoint) co-attention encoders are independent too: one launch for both when the shapes allow
            if n.xenc_ok(K, L) and n.xenc_ok(Vp, L):
                c.glob, c.loc = n.cross_fwd_fused([
                    ("global", plan, c.gin.out, K, plan["gmap_mask"], gl_, plan["gmap_nodes"], c.txt_out, L, plan["txt_mask"], tl, plan["txt_tokens"], inp.dist),
                    ("local", plan, c.vin.out, Vp, plan["vp_mask"], vl, B * Vp, c.txt_out, L, plan["txt_mask"], tl, plan["txt_tokens"])])
            else:
              c.glob, (c.vin, c.loc) = self._par(
                lambda: n.cross_fwd("global", plan, c.gin.out, K, plan["gmap_mask"], gl_, plan["gmap_nodes"],
                                    c.txt_out, L, plan["txt_mask"], tl, plan["txt_tokens"], dist=inp.dist), _local)
            o.update(gmap_embeds=c.glob.out, gmap_attns=c.glob.P, vp_embeds=c.loc.out, vp_attns=c.loc.P)
            if task == "sap":
                from . import lib as _lib
                use_gate = bool(cfg_get(self.config, "glocal_fuse"))
                g1, l1_ = n.lin("global_sap_head.net.0.weight"), n.lin("local_sap_head.net.0.weight")
                with _lib.group():                  # the three heads' first projections are independent: one grouped launch
                    c.Yg = O.linear_fwd(c.glob.out, g1.W, g1.b, B * K, epilogue=2)
                    c.Yl = O.linear_fwd(c.loc.out, l1_.W, l1_.b, B * Vp, epilogue=2)
                    if use_gate:
                        f1 = n.lin("sap_fuse_linear.net.0.weight")
                        tmp = O.linear_fwd(c.glob.out, f1.W, None, B, lda=K * H, ldb=2 * H, K=H)
                c.g_raw = self._cls_tail("global_sap_head.", c.Yg, B * K)
                c.l_raw = self._cls_tail("local_sap_head.", c.Yl, B * Vp)
                if use_gate:
                    c.Yf = O.linear_fwd(c.loc.out, f1.W[:, H:], f1.b, B, lda=Vp * H, ldb=2 * H, K=H, residual=tmp, epilogue=0)
                    self._relu(c.Yf)      # ReLU after the two partial products are summed
                    fln, f2 = n.ln("sap_fuse_linear.net.2"), n.lin("sap_fuse_linear.net.3.weight")
                    c.fuse_raw = n.new(B, dtype=torch.float32)
                    O.lndot_fwd(c.Yf, B, H, fln.g, fln.b, n.eps, f2.Wm, f2.b, c.fuse_raw)
                else:
                    c.fuse_raw = n.zeros(B, dtype=torch.float32)
                c.use_gate = use_gate
                c.gl, c.ll, c.fl = n.new(B, K, dtype=torch.float32), n.new(B, Vp, dtype=torch.float32), n.new(B, K, dtype=torch.float32)
                c.sap_fused = bool(compute_loss and O.SAP_LOSS_FUSED and K <= 512 and Vp <= 128)
                if not c.sap_fused:           # (with a loss to follow, the logit fusion rides in the loss launch: _losses / O.sap_fuse_loss)
                    O.sap_fuse_fwd(B, K, Vp, c.g_raw, c.l_raw, c.fuse_raw, plan["gmask"], plan["lmask"], plan["fsrc"], plan["bwmask"],
                                   use_gate, c.gl, c.ll, c.fl)
                o.update(global_logits=c.gl, local_logits=c.ll, fused_logits=c.fl)
            elif task == "cfp":
                c.g0, c.v0, c.t0, c.gv0 = n.new(B, H), n.new(B, H), n.new(B, H), n.new(B, H)
                # the four [CLS]-row selections (fused = map row + viewpoint row) in one launch, the four heads as one grouped launch
                O.csr_gather_multi(H, [dict(out=c.g0, n_out=B, src1=c.glob.out, csr1=plan["g0"]),
                                       dict(out=c.v0, n_out=B, src1=c.loc.out, csr1=plan["v0"]),
                                       dict(out=c.t0, n_out=B, src1=c.txt_out, csr1=plan["t0"]),
                                       dict(out=c.gv0, n_out=B, src1=c.glob.out, csr1=plan["g0"], src2=c.loc.out, csr2=plan["v0"])])
                from . import lib as _lib
                with _lib.group():
                    c.cfp = []
                    for key, src in (("gmap", c.g0), ("vp", c.v0), ("fused", c.gv0), ("txt", c.t0)):
                        hl = n.lin(f"cfp_heads.{key}.weight")
                        c.cfp.append(O.linear_fwd(src, hl.W, hl.b, B))
                o["cfp"] = tuple(c.cfp)
            else:
                raise ValueError(task)
        if not compute_loss:
            if return_outputs:
                return o
            if task == "mlm":
                return {"predict": o["predict"]}
            if task == "mrc":
                return c.mlogits, plan["mrc_targets"], None, None
            if task == "sap":
                return dict(global_logits=c.gl, local_logits=c.ll, fused_logits=c.fl,
                            global_act_labels=self._dev(batch["global_act_labels"]), local_act_labels=self._dev(batch["local_act_labels"]))
            return o["cfp"]
        self._ctx = c
        if callable(teacher_outputs):          # teacher forward running on a side stream: join here
            teacher_outputs = teacher_outputs()
        scaled = self.loss_scale is not None and self.store.requires_grad
        if scaled:
            O.seed_scale(self.loss_scale)          # every gradient-seeding loss launch below multiplies its seed by the device-side scale
        try:
            out = self._losses(c, o, teacher_outputs, rw)
        finally:
            if scaled:
                O.seed_scale(None)
        out["outputs"] = o
        if self.store.requires_grad and torch.is_grad_enabled():
            out["loss"] = _BackwardHook.apply(out["loss"], self._anchor, self)
        return out

    def _relu(self, x):
        # in-place ReLU on a tiny [B,H] head tensor via the activation-derivative kernel: x * relu'(x) == relu(x)
        return O.dact(x, x, 2, out=x)

    # ---- losses + gradient seeds ------------------------------------------------------------------------
    # The distillation terms of a step are independent of each other: their five student->teacher-width projections go out as
    # ONE grouped GEMM launch, all MSE terms (<= 10) as ONE launch, and the five projection input-gradients as ONE grouped launch.
    # Shape-bucketed batches under graph replay (host/stream_graph.py, host/bucket.py): plan["dyn"] = {i: int32[2 n], f: fp32[n], fill: {}}.
    # A distillation term then takes its TRUE (outer, inner) extent and its normaliser from device memory (slot of DYN_TERMS), and registers
    # how the slot is computed from the batch's true sizes -- fill[term](true) -> (outer, inner, norm) -- for StreamStep to evaluate per batch.
    def _dyn(self, c, term, fn, mod=0):
        d = c.plan.get("dyn")
        if d is None:
            return {}
        i = DYN_TERMS.index(term)
        d["fill"][term] = fn
        return dict(valid_dev=d["i"][2 * i:2 * i + 2], norm_dev=d["f"][i:i + 1], norm=1.0, valid_mod=mod)

    def _kd_emb(self, c, slot, s_t, t_t, proj, M, outer, w, coef, d_acc, dyn=None):
        c.kd_emb.append((slot, s_t, t_t, self.net.lin(f"bert.{proj}.weight"), M, outer, w, coef, d_acc, dyn))

    def _kd_attn(self, c, slot, sP, tP, Bn, Nq, Nk, ldp, nh_s, nh_t, w, coef, dyn=None):
        """dyn = (term, 'q' | None = which true size bounds the query rows, which true sizes enter the normaliser as (Nq, Nk))"""
        n = self.net
        hmin = min(nh_s, nh_t)
        dP = None
        if self.store.requires_grad:      # every student head is written when hmin == nh_s (pad columns included): no fill needed
            dP = (n.new if hmin == nh_s else n.zeros)(Bn, nh_s, Nq, ldp, dtype=torch.float32)
        q = dict(s=sP, t=tP, outer=Bn, inner=hmin * Nq * ldp, s_stride=nh_s * Nq * ldp, t_stride=nh_t * Nq * ldp, w=w, rows_per_w=1,
                 norm=1.0 / (Bn * hmin * Nq * Nk), coef=coef[0], coef_dev=coef[1], loss=c.slots[slot:slot + 1], ds=dP,
                 g_stride=nh_s * Nq * ldp)
        if dyn is not None:
            term, qk, kk = dyn           # names of the true sizes of the query / key extents (None: the extent is not padded)
            q.update(self._dyn(c, term, lambda t, Bn=Bn, hmin=hmin, Nq=Nq, Nk=Nk, ldp=ldp, qk=qk, kk=kk:
                               (Bn, (t[qk] if qk else Nq) * ldp, 1.0 / (Bn * hmin * (t[qk] if qk else Nq) * (t[kk] if kk else Nk))), mod=Nq * ldp))
        c.kd_mse.append(q)
        return dP

    def _kd_flush(self, c):
        from . import lib as L
        n, train = self.net, self.store.requires_grad
        jobs = c.kd_emb
        with L.group():
            sps = [O.linear_fwd(s_t, pl.W, pl.b, M) for (_, s_t, _, pl, M, _, _, _, _, _) in jobs]
        dss = []
        for (slot, s_t, t_t, pl, M, outer, w, coef, d_acc, dyn), sp in zip(jobs, sps):
            Ht = pl.N
            inner = (M // outer) * Ht
            ds = n.new(M, Ht) if train else None
            dss.append(ds)
            q = dict(s=sp, t=t_t, outer=outer, inner=inner, s_stride=inner, t_stride=inner, w=w, rows_per_w=1, norm=1.0 / (M * Ht),
                     coef=coef[0], coef_dev=coef[1], loss=c.slots[slot:slot + 1], ds=ds, g_stride=inner)
            if dyn is not None:           # (term, true size bounding the OUTER extent or None, true size bounding the rows of a block or None)
                term, ok, ik = dyn
                rows = M // outer
                q.update(self._dyn(c, term, lambda t, outer=outer, rows=rows, Ht=Ht, ok=ok, ik=ik:
                                   ((t[ok] if ok else outer), (t[ik] if ik else rows) * Ht,
                                    1.0 / ((t[ok] if ok else outer) * (t[ik] if ik else rows) * Ht))))
            c.kd_mse.append(q)
        if c.kd_mse:
            O.mse_multi(c.kd_mse)
        if train:
            for (slot, s_t, t_t, pl, M, outer, w, coef, d_acc, dyn), ds in zip(jobs, dss):
                O.linear_dw(ds, s_t, pl.dW, pl.db, M)
            with L.group():
                for (slot, s_t, t_t, pl, M, outer, w, coef, d_acc, dyn), ds in zip(jobs, dss):
                    O.linear_dx(ds, pl.W, M, out=d_acc, residual=d_acc)
        c.kd_emb, c.kd_mse = [], []

    def _rw_device(self, rw):
        """MKRW ability weights as a DEVICE tensor (a captured HIP graph re-reads them every replay)"""
        if rw is None:
            rw = [1.0] * 5
        if torch.is_tensor(rw) and rw.is_cuda:
            return rw
        cached = getattr(self, "_rwd_cache", None)
        key = tuple(float(x) for x in rw)
        if cached is None or cached[0] != key:
            self._rwd_cache = cached = (key, torch.tensor(list(key), dtype=torch.float32).to(self.device_))
        return cached[1]

    def _losses(self, c, o, t, rw):
        n, cfg, plan, task = self.net, self.config, c.plan, c.task
        B, L, K, Vp, H = plan["B"], plan["L"], plan["K"], plan["Vp"], n.H
        train = self.store.requires_grad
        O.defer_dw(train)         # the distillation heads' weight gradients join the deferred grouped launch of backward()
        kdl = getattr(cfg, "kdl", None)
        kd = t is not None and kdl is not None
        alpha = float(kdl["kd_alpha"]) if kd else 0.0
        sc = 1.0 - alpha
        dyn = train and self.loss_scale is not None      # dynamic loss scale: the kernels multiply their seeds by the device word, not the host
        gs = float(self.grad_scale) if (train and not dyn) else 1.0
        scg = sc * gs                    # coefficient of the supervised gradient seeds (the loss values use sc)
        # every zero-initialised gradient accumulator of the step comes out of ONE zeroed arena per dtype (two fills instead of ~12 tiny ones)
        nm_ = plan["n_mask"] if task == "mlm" else 0
        c.dx_slabs = MLM_DX_SPLITK if (nm_ and MLM_TAIL_FUSED and not MLM_DX_ATOMICS and MLM_DX_SPLITK > 1) else 0
        if c.dx_slabs:                     # [splits, n_mask, H] slabs, every element stored by its split: no zeroing
            f32 = n.zeros(16, dtype=torch.float32)
            c.slots, c.d_hm32 = f32, n.new(c.dx_slabs * nm_, H, dtype=torch.float32)
        else:
            f32 = n.zeros(16 + nm_ * H, dtype=torch.float32)
            c.slots, c.d_hm32 = f32[:16], (f32[16:].view(nm_, H) if nm_ else None)
        if train:
            rows = [B * L, plan["Np"] * plan["V"], plan["Np"]]                               # d_txt, d_pano, d_fused
            rows += {"sap": [B * K, B * Vp, B * L], "cfp": [B * K, B * Vp, B * L, B, B, B], "mlm": [B * L, B * K], "mrc": [B * Vp]}[task]
            arena = n.zeros(sum(rows) * H)
            offs = [0]
            for r in rows:
                offs.append(offs[-1] + r * H)
            pool = [arena[offs[i]:offs[i + 1]].view(rows[i], H) for i in range(len(rows))]

            def zz(*shape):
                t = pool.pop(0)
                assert tuple(t.shape) == tuple(shape), (tuple(t.shape), shape)
                return t
        else:
            zz = lambda *s: None
        c.d_txt, c.d_pano, c.d_fused = zz(B * L, H), zz(plan["Np"] * plan["V"], H), zz(plan["Np"], H)
        c.dP_txt = c.dP_pano = c.dP_g = c.dP_l = None
        c.kd_emb, c.kd_mse = [], []
        res = {}
        # ---- supervised ------------------------------------------------------------------------------
        if task == "sap":
            c.d_gmap, c.d_vp, c.d_txt2 = zz(B * K, H), zz(B * Vp, H), zz(B * L, H)
            c.rows = n.new(3, B, dtype=torch.float32)
            c.dgl, c.dll, c.dfl = (n.new(B, K, dtype=torch.float32), n.new(B, Vp, dtype=torch.float32), n.new(B, K, dtype=torch.float32)) \
                if train else (None, None, None)
            ga, la = plan["global_act_labels"], plan["local_act_labels"]
            c.sap_w = c.kdrows = None
            if getattr(c, "sap_fused", False):
                # logit fusion + the three CE rows + teacher-sample weights + action-distillation rows: ONE launch (six before)
                hard = bool(kd and kdl.get("teacher_sample_hard_mining", False))
                pred = bool(kd and "predict" in kdl["kdl_tasks"])
                if hard:
                    c.sap_w = n.new(B, dtype=torch.float32)
                if pred:
                    c.kdrows = n.new(B, dtype=torch.float32)
                if kd:
                    rwd = self._rw_device(rw)
                O.sap_fuse_loss(B, K, Vp, c.g_raw, c.l_raw, c.fuse_raw, plan["gmask"], plan["lmask"], plan["fsrc"], plan["bwmask"], c.use_gate,
                                c.gl, c.ll, c.fl, ga, la, scg / B, c.rows, dgl=c.dgl, dll=c.dll, dfl=c.dfl,
                                t_fused=t["fused_logits"] if (hard or pred) else None,
                                w_rate=float(kdl["t_sample_preprocess_exp_decay"]) if hard else 0.0, w_out=c.sap_w,
                                T=float(kdl["kd_temperature"]) if kd else 1.0, kd_norm=(1.0 / B if hard else 1.0 / (B * K)),
                                kd_coef=alpha * gs, kd_coef_dev=rwd[4:5] if pred else None, kd_rows=c.kdrows)
            else:
                O.ce_rows(c.gl, B, K, K, ga, coef=scg / B, loss_row=c.rows[0], dlogits=c.dgl, ldd=K)
                O.ce_rows(c.ll, B, Vp, Vp, la, coef=scg / B, loss_row=c.rows[1], dlogits=c.dll, ldd=Vp)
                O.ce_rows(c.fl, B, K, K, ga, coef=scg / B, loss_row=c.rows[2], dlogits=c.dfl, ldd=K)
            sup_rows, sup_w, sup_scale = c.rows, None, 1.0 / B
        elif task == "mlm":
            c.d_x, c.d_gin0 = zz(B * L, H), zz(B * K, H)
            nm = plan["n_mask"]
            c.rows = n.new(nm, dtype=torch.float32)
            c.dlogits = (n.new(nm, c.ldv) if self.keep_mlm_logits else c.logits) if train else None
            roww = plan.get("mlm_row_w")      # shape buckets: nm counts padded (ignored) rows too, the true 1 / n_mask rides in per-row weights
            O.ce_rows(c.logits, nm, cfg.vocab_size, c.ldv, plan["mlm_labels"], ignore_index=-1, coef=scg if roww is not None else scg / nm,
                      row_w=roww, loss_row=c.rows, dlogits=c.dlogits, ldd=c.ldv)
            if train and not self.keep_mlm_logits:
                o["predict"] = None          # overwritten in place by its gradient
            sup_rows, sup_w, sup_scale = c.rows, roww, (1.0 if roww is not None else 1.0 / nm)
        elif task == "mrc":
            c.d_vp = zz(B * Vp, H)
            nm, Pn = plan["n_mrc"], c.mlogits.shape[1]
            c.rows = n.new(nm, dtype=torch.float32)
            c.dmlogits = n.new(nm, Pn) if train else None
            roww = plan.get("mrc_row_w")      # shape buckets (as mlm): nm counts padded rows, the true 1 / n rides in per-row weights
            O.softkl_rows(c.mlogits, nm, Pn, Pn, plan["mrc_targets"], coef=scg if roww is not None else scg / nm, row_w=roww,
                          loss_row=c.rows, dlogits=c.dmlogits, ldd=Pn)
            sup_rows, sup_w, sup_scale = c.rows, roww, (1.0 if roww is not None else 1.0 / nm)
        else:
            c.d_gmap, c.d_vp, c.d_txt2 = zz(B * K, H), zz(B * Vp, H), zz(B * L, H)
            c.d_cls0 = (zz(B, H), zz(B, H), zz(B, H))
            temp = float(cfg_get(cfg, "cfp_temperature"))
            c.rows = n.new(6, B, dtype=torch.float32)
            txt_o = c.cfp[3]
            c.dsim, c.d_cfp = [], None
            ar = plan["arange_b"]
            lds = c.lds = rup(B, 8)
            fused = O.cfp_loss_ok(B, H)
            if fused:                            # the three contrastive terms, forward and backward, as one launch (csrc/loss.hip cfp_loss_kernel)
                c.d_cfp = [n.new(B, H) for _ in range(4)] if train else None
                O.cfp_loss(B, H, c.cfp[:3], txt_o, temp, scg * 0.5 / B, c.rows, d_a=c.d_cfp[:3] if train else None, d_txt=c.d_cfp[3] if train else None)
            for i in (() if fused else range(3)):
                a = c.cfp[i]
                sim, simT = n.new(B, lds, dtype=torch.float32), n.new(B, lds, dtype=torch.float32)
                O.gemm(0, a, txt_o, sim, B, B, H, H, H, lds, alpha=1.0 / temp)
                O.gemm(0, txt_o, a, simT, B, B, H, H, H, lds, alpha=1.0 / temp)
                d1 = n.new(B, lds, dtype=torch.float32) if train else None
                d2 = n.new(B, lds, dtype=torch.float32) if train else None
                O.ce_rows(sim, B, B, lds, ar, coef=scg * 0.5 / B, loss_row=c.rows[2 * i], dlogits=d1, ldd=lds)
                O.ce_rows(simT, B, B, lds, ar, coef=scg * 0.5 / B, loss_row=c.rows[2 * i + 1], dlogits=d2, ldd=lds)
                c.dsim.append((d1, d2))
            c.temp = temp
            sup_rows, sup_w, sup_scale = c.rows, None, 0.5 / B
        kd_rows, rwd = None, None
        # ---- MAKD (pretrain flavour; oracle/makd_ref.pretrain_makd) -------------------------------------
        if kd:
            # MKRW ability weights as a DEVICE tensor (a captured HIP graph re-reads them every replay)
            if rw is None:
                rw = [1.0] * 5
            rwd = self._rw_device(rw)
            rw = [(alpha * gs, rwd[i:i + 1]) for i in range(5)]       # (host factor of the gradient seed, device-side ability weight)
            tasks, types = kdl["kdl_tasks"], kdl["kdl_task_types"]
            emb, att = "emb" in types, "attn" in types
            T = float(kdl["kd_temperature"])
            w = None
            if kdl.get("teacher_sample_hard_mining", False) and task == "sap":
                if getattr(c, "sap_w", None) is not None:
                    w = c.sap_w                  # written by the fused loss launch above
                else:
                    w = n.new(B, dtype=torch.float32)
                    O.ce_rows(t["fused_logits"], B, K, K, plan["global_act_labels"], w_out=w,
                              w_rate=float(kdl["t_sample_preprocess_exp_decay"]))
            nh_s = n.nh
            nh_t = t["txt_attns"].shape[1]
            Np, V = plan["Np"], plan["V"]
            if "txt" in tasks:
                if emb:
                    self._kd_emb(c, 0, o["txt_embeds"], t["txt_embeds"], "txt_emb_w", B * L, B, w, rw[0], c.d_txt, dyn=("txt_emb", None, "L"))
                if att:
                    c.dP_txt = self._kd_attn(c, 1, o["txt_attns"], t["txt_attns"], B, L, L, c.txt.ldp, nh_s, nh_t, w, rw[0], dyn=("txt_attn", "L", "L"))
            # panorama tensors have leading dim sum(T): the sample weights [B] only broadcast (and are only applied by
            # mse_loss, pretrain kd_loss.py:11-16) when every trajectory has exactly one step
            wp = w if (w is not None and Np == B) else None
            if "img" in tasks:
                if emb:
                    self._kd_emb(c, 2, o["pano_embeds"], t["pano_embeds"], "kdl_img_w", Np * V, Np, wp, rw[1], c.d_pano, dyn=("img_emb", "Np", None))
                    self._kd_emb(c, 3, o["pano_fused_embeds"], t["pano_fused_embeds"], "kdl_avg_img_w", Np, Np, wp, rw[1], c.d_fused, dyn=("img_fused", "Np", None))
                if att:
                    ldp = c.pano.ldp
                    g_img = n.new(Np, V, ldp, dtype=torch.float32) if train else None
                    q = dict(s=o["img_attns"], t=t["img_attns"], outer=Np, inner=V * ldp, s_stride=V * ldp, t_stride=V * ldp, w=wp, rows_per_w=1,
                             norm=1.0 / (Np * V * V), coef=rw[1][0], coef_dev=rw[1][1], loss=c.slots[4:5], ds=g_img, g_stride=V * ldp)
                    q.update(self._dyn(c, "img_attn", lambda tr, V=V, ldp=ldp: (tr["Np"], V * ldp, 1.0 / (tr["Np"] * V * V))))
                    c.kd_mse.append(q)
            if "global" in tasks and task != "mrc":
                if task == "mlm":
                    x, Pm, Nq, Nk, ldp, d_acc, qk, kk = c.l2v.out, c.l2v.P, L, K, c.l2v.ldp, c.d_x, "L", "K"
                else:
                    x, Pm, Nq, Nk, ldp, d_acc, qk, kk = c.glob.out, c.glob.P, K, L, c.glob.ldp, c.d_gmap, "K", "L"
                if emb:
                    self._kd_emb(c, 5, x, t["gmap_embeds"], "global_cross_w", B * Nq, B, w, rw[2], d_acc, dyn=("g_emb", None, qk))
                if att:
                    c.dP_g = self._kd_attn(c, 6, Pm, t["gmap_attns"], B, Nq, Nk, ldp, nh_s, nh_t, w, rw[2], dyn=("g_attn", qk, kk))
            if "local" in tasks and task != "mlm":
                if emb:
                    self._kd_emb(c, 7, c.loc.out, t["vp_embeds"], "local_cross_w", B * Vp, B, w, rw[3], c.d_vp, dyn=("l_emb", None, None))
                if att:
                    c.dP_l = self._kd_attn(c, 8, c.loc.P, t["vp_attns"], B, Vp, L, c.loc.ldp, nh_s, nh_t, w, rw[3], dyn=("l_attn", None, "L"))
            self._kd_flush(c)
            if "img" in tasks and att and train:
                c.dP_pano = n.new(Np, nh_s, V, c.pano.ldp, dtype=torch.float32)
                O.head_mean_bwd(g_img, c.dP_pano, Np, nh_s, V * c.pano.ldp)
            if "predict" in tasks and task == "sap":
                if getattr(c, "sap_fused", False):
                    kd_rows = c.kdrows           # (and the gradient is in c.dfl already)
                else:
                    c.kdrows = n.new(B, dtype=torch.float32)
                    O.kd_rows(c.fl, t["fused_logits"], B, K, K, T, w=w, norm=(1.0 / B if w is not None else 1.0 / (B * K)),
                              coef=rw[4][0], coef_dev=rw[4][1], loss_row=c.kdrows, ds=c.dfl, accumulate=True)
                    kd_rows = c.kdrows
        # supervised mean, the action-distillation row sum, the ten ability-weighted MAKD terms, their sum and the total: ONE launch
        lo = O.loss_assemble(sup_rows.reshape(-1), sup_w, sup_scale, kd_rows, c.slots, rwd, alpha, kd, n.new(16, dtype=torch.float32))
        res["supervised_loss"] = lo[0]
        if kd:
            res["kdl_terms"] = {k: lo[1 + i] for i, k in enumerate(KD_SLOTS)}
            res["kdl_loss"] = lo[11]
        res["loss"] = lo[12]
        return res

    # ---- backward ---------------------------------------------------------------------------------------
    def backward(self, on_bucket=None):
        """the explicit backward in two phases (heads + cross-modal encoders | text + panorama encoders).  on_bucket(i, ctx):
        called when gradient bucket i is final -- 0 after phase 1 (its weight-gradient GEMMs are flushed first), 1 at the end
        (trainer.GradSync launches the data-parallel exchange of that bucket from it)."""
        self.backward_phase1()
        c = self._ctx
        if on_bucket is not None:
            # bucket 0's weight gradients are still QUEUED at this point (phase 1 and the forward's distillation heads defer them):
            # they must be on the stream before the hook makes the exchange stream wait on it, or RCCL reduces the range while the
            # grouped dW kernels still add into it (same order as trainer.capture_split: phase 1; flush; cut)
            # round 6: on the weight-gradient stream when there is one (ops.flush_dw_early) -- the bucket's dW launch and, behind it, its exchange
            # (GradSync._on_side waits for that stream too) then run beside the rest of the backward; the main stream never stops for either
            early = self.device_.type == "cuda" and O.flush_dw_early(self.device_)
            if not early:
                O.flush_dw(keep_active=True)
            O.join_side()
            on_bucket(0, c)

            def cut():                      # the top MID_CUT blocks of the text / panorama stacks are done: their slice is final once flushed
                if not (self.device_.type == "cuda" and O.flush_dw_early(self.device_)):
                    O.flush_dw(keep_active=True)
                O.join_side()
                on_bucket(1, c)
            self.backward_phase2(on_cut=cut)
            on_bucket(2, c)
        else:
            # no exchange to cut for: the weight gradients phase 1 queued (heads, cross-modal encoders, distillation projections) go out NOW on the
            # weight-gradient stream and run under phase 2's latency-bound chain (ops.flush_dw_early)
            if self.device_.type == "cuda":
                O.flush_dw_early(self.device_)
            self.backward_phase2()

    @torch.no_grad()
    def backward_phase1(self):
        c = self._ctx
        assert c is not None, "backward() without a compute_loss=True forward"
        self.store.ensure_grads()
        O.defer_dw(True)          # weight-gradient GEMMs are queued and launched grouped at the end
        n, plan, task = self.net, c.plan, c.task
        B, L, K, Vp, H = plan["B"], plan["L"], plan["K"], plan["Vp"], n.H
        if task == "sap":
            dg, dl, df = n.new(B, K, dtype=torch.float32), n.new(B, Vp, dtype=torch.float32), n.new(B, dtype=torch.float32)
            O.sap_fuse_bwd(B, K, Vp, c.g_raw, c.l_raw, c.fuse_raw, plan["gmask"], plan["lmask"], plan["fsrc"], plan["bwmask"],
                           c.use_gate, c.dgl, c.dll, c.dfl, dg, dl, df)
            from . import lib as _lib
            heads = []
            for prefix, X, Y, dlogit, M, d_acc in (("global_sap_head.", c.glob.out, c.Yg, dg, B * K, c.d_gmap),
                                                   ("local_sap_head.", c.loc.out, c.Yl, dl, B * Vp, c.d_vp)):
                l1, ln, l2 = n.lin(prefix + "net.0.weight"), n.ln(prefix + "net.2"), n.lin(prefix + "net.3.weight")
                dZh = n.new(M, H)
                O.lndot_bwd(Y, M, H, ln.g, ln.b, n.eps, l2.Wm, dlogit, dZh, ln.dg, ln.db, l2.dW, l2.db)
                O.linear_dw(dZh, X, l1.dW, l1.db, M)
                heads.append((dZh, l1, M, d_acc))
            with _lib.group():                      # the two heads' input gradients go to different tensors: one grouped launch
                for dZh, l1, M, d_acc in heads:
                    O.linear_dx(dZh, l1.W, M, out=d_acc, residual=d_acc)
            if c.use_gate:
                f1, fln, f2 = n.lin("sap_fuse_linear.net.0.weight"), n.ln("sap_fuse_linear.net.2"), n.lin("sap_fuse_linear.net.3.weight")
                dZ = n.new(B, H)
                O.lndot_bwd(c.Yf, B, H, fln.g, fln.b, n.eps, f2.Wm, df, dZ, fln.dg, fln.db, f2.dW, f2.db)
                # dW[:, :H] += dZ^T g0 ; dW[:, H:] += dZ^T v0 (bias grad once)
                O.linear_dw(dZ, c.glob.out, f1.dW, f1.db, B, N=H, K=H, ldb=K * H, ldc=2 * H)
                O.linear_dw(dZ, c.loc.out, f1.dW[:, H:], None, B, N=H, K=H, ldb=Vp * H, ldc=2 * H)
                with _lib.group():
                    O.linear_dx(dZ, f1.W, B, out=c.d_gmap, residual=c.d_gmap, ldb=2 * H, ldc=K * H, N=H, K=H)
                    O.linear_dx(dZ, f1.W[:, H:], B, out=c.d_vp, residual=c.d_vp, ldb=2 * H, ldc=Vp * H, N=H, K=H)
        elif task == "cfp":
            d_outs = c.d_cfp if c.d_cfp is not None else [n.new(B, H) for _ in range(4)]
            txt_o = c.cfp[3]
            for i in (() if c.d_cfp is not None else range(3)):
                d1, d2 = c.dsim[i]
                a = c.cfp[i]
                it = 1.0 / c.temp
                # sim = a txt^T / temp ; simT = txt a^T / temp
                lds = c.lds
                e1, e2 = self._as(d1), self._as(d2)
                O.gemm(1, e1, txt_o, d_outs[i], B, H, B, lds, H, H, alpha=it)
                O.gemm(2, e2, txt_o, d_outs[i], B, H, B, lds, H, H, alpha=it, residual=d_outs[i], ldr=H)
                if i == 0:
                    O.gemm(2, e1, a, d_outs[3], B, H, B, lds, H, H, alpha=it)
                else:
                    O.gemm(2, e1, a, d_outs[3], B, H, B, lds, H, H, alpha=it, residual=d_outs[3], ldr=H)
                O.gemm(1, e2, a, d_outs[3], B, H, B, lds, H, H, alpha=it, residual=d_outs[3], ldr=H)
            d_g0, d_v0, d_t0 = c.d_cls0
            heads = list(zip((("gmap", c.g0, (d_g0,)), ("vp", c.v0, (d_v0,)), ("fused", c.gv0, (d_g0, d_v0)), ("txt", c.t0, (d_t0,))), d_outs))
            for (key, src, dsts), d_o in heads:
                hl = n.lin(f"cfp_heads.{key}.weight")
                O.linear_dw(d_o, src, hl.dW, hl.db, B)
            # the five input-gradient GEMMs (48 rows each: 6-10 us apiece on the chain) as TWO grouped launches: the three heads with a destination of their
            # own, then the fused head's two (they add into the map / viewpoint rows the first launch wrote)
            from . import lib as _lib
            for sel in (("gmap", "vp", "txt"), ("fused",)):
                with _lib.group():
                    for (key, src, dsts), d_o in heads:
                        if key in sel:
                            hl = n.lin(f"cfp_heads.{key}.weight")
                            for dst in dsts:
                                O.linear_dx(d_o, hl.W, B, out=dst, residual=dst)
            O.csr_gather_multi(H, [dict(out=c.d_gmap, n_out=B * K, accumulate=True, src1=d_g0, csr1=plan["g0_T"]),
                                   dict(out=c.d_vp, n_out=B * Vp, accumulate=True, src1=d_v0, csr1=plan["v0_T"]),
                                   dict(out=c.d_txt, n_out=B * L, accumulate=True, src1=d_t0, csr1=plan["t0_T"])])
        if task == "mlm":
            nm, Vv = plan["n_mask"], self.config.vocab_size
            dlog = c.dlogits                     # CE wrote the gradient (in place unless keep_mlm_logits)
            Wemb = self.store.w("bert.embeddings.word_embeddings.weight")
            O.linear_dw(dlog, c.hm, self.store.g("bert.embeddings.word_embeddings.weight"), self.store.g("mlm_head.predictions.bias"),
                        nm, N=Vv, K=H, lda=c.ldv)
            # dx over the 50k-wide vocabulary: split-K into an fp32 accumulator (18 output tiles alone cannot fill 256 CUs)
            d_hm32 = c.d_hm32
            if c.dx_slabs:
                O.gemm(1, dlog, Wemb, d_hm32, nm, H, Vv, c.ldv, H, H, splitk=-c.dx_slabs)
            else:
                O.gemm(1, dlog, Wemb, d_hm32, nm, H, Vv, c.ldv, H, H, splitk=MLM_DX_SPLITK, accumulate=True)
            tn = n.ln("mlm_head.predictions.transform.LayerNorm")
            if MLM_TAIL_FUSED:           # cast + LayerNorm backward + gelu' as one launch (the fp32 accumulator -- or its slabs -- read as it is)
                d_tz = O.ln_bwd_tail(nm, H, d_hm32, c.hm, tn.g, tn.b, c.rstd_hm, c.tz, 1, n.new(nm, H), tn.dg, tn.db, nslab=max(1, c.dx_slabs))
            else:
                d_hm = O.cast_to(d_hm32, self.compute_dtype)
                d_tg = n.new(nm, H)
                O.ln_bwd(nm, H, d_hm, y=c.hm, gamma=tn.g, beta=tn.b, rstd=c.rstd_hm, dx=d_tg, dgamma=tn.dg, dbeta=tn.db)
                d_tz = O.dact(d_tg, c.tz, 1)
            t = n.lin("mlm_head.predictions.transform.dense.weight")
            O.linear_dw(d_tz, c.hm_in, t.dW, t.db, nm)
            d_hin = O.linear_dx(d_tz, t.W, nm)
            O.csr_gather(d_hin, *plan["mlm_rows_T"], c.d_x, B * L, H, accumulate=True)
            d_gin = c.d_gin0
            d_t2 = n.cross_bwd(c.l2v, c.d_x, d_gin, c.dP_g)
            O.add_(c.d_txt, d_t2)
        elif task == "mrc":
            nm = plan["n_mrc"]
            l1, ln, l2 = n.lin("image_classifier.net.0.weight"), n.ln("image_classifier.net.2"), n.lin("image_classifier.net.3.weight")
            O.linear_dw(c.dmlogits, c.mZ, l2.dW, l2.db, nm)
            dZ = O.linear_dx(c.dmlogits, l2.W, nm)
            dY = n.new(nm, H)
            O.ln_bwd(nm, H, dZ, y=c.mZ, gamma=ln.g, beta=ln.b, rstd=c.m_rstd, dx=dY, dgamma=ln.dg, dbeta=ln.db)
            dpre = O.dact(dY, c.mY, 2)                 # relu'(pre) == relu'(relu(pre))
            O.linear_dw(dpre, c.mx, l1.dW, l1.db, nm)
            dmx = O.linear_dx(dpre, l1.W, nm)
            O.csr_gather(dmx, *plan["mrc_rows_T"], c.d_vp, B * Vp, H, accumulate=True)
            d_vin = n.cross_bwd(c.loc, c.d_vp, c.d_txt, c.dP_l)
            n.vp_in_bwd(c.vin, plan, d_vin, c.d_pano)
        else:
            d_txt2 = c.d_txt2                    # the two encoders accumulate their text gradients separately (no race)
            if n.rbw_ok():                       # both encoders in shared row-block launches (engine.cross_stacks_bwd); one text accumulator: the
                d_gin, d_vin = n.cross_stacks_bwd([(c.glob, c.d_gmap, c.d_txt, c.dP_g), (c.loc, c.d_vp, c.d_txt, c.dP_l)])      # six parts fold at the end
            else:
                d_gin, d_vin = self._par(lambda: n.cross_bwd(c.glob, c.d_gmap, c.d_txt, c.dP_g),
                                         lambda: n.cross_bwd(c.loc, c.d_vp, d_txt2, c.dP_l))
                O.add_(c.d_txt, d_txt2)
            n.nodes_in_bwd(plan, c.gin, d_gin, c.vin, d_vin, c.d_pano, c.d_fused)       # both input stages in shared launches
        if task == "mlm":
            n.gmap_in_bwd(c.gin, plan, d_gin, c.d_pano, c.d_fused)

    @torch.no_grad()
    def backward_phase2(self, on_cut=None):
        """text / panorama encoders + embeddings (the caller flushes phase 1's queued weight-gradient GEMMs first when bucket 0 of the
        exchange must be final before this phase starts).  on_cut(): called once, when the top MID_CUT blocks of both stacks are done
        (trainer.GradSync's middle bucket: text layers [nl - MID_CUT, nl) + the panorama layers); on paths without the shared row-block
        backward it is called right before the end -- the middle bucket is then simply not early."""
        c = self._ctx
        n, plan = self.net, c.plan
        fired = []
        if getattr(c, "islands", None):
            self._causal_bwd(c)

        def on_iter(k):
            if on_cut is not None and k == MID_CUT and not fired:
                fired.append(k)
                on_cut()
        if n.rbw_ok() and n.enc_ok(plan["L"], self.config.num_l_layers) and n.enc_ok(plan["V"], self.config.num_pano_layers):
            n.encoders_bwd(c.txt, c.pano, plan, c.d_txt, c.dP_txt, c.d_pano, c.d_fused, c.dP_pano, on_iter=on_iter)      # both stacks in shared launches
            if on_cut is not None and not fired:
                on_cut()
        elif on_cut is not None:
            self._par(lambda: n.text_bwd(c.txt, plan, c.d_txt, c.dP_txt),
                      lambda: n.pano_bwd(c.pano, plan, c.d_pano, c.d_fused, c.dP_pano))
            on_cut()
        else:
            self._par(lambda: n.text_bwd(c.txt, plan, c.d_txt, c.dP_txt),
                      lambda: n.pano_bwd(c.pano, plan, c.d_pano, c.d_fused, c.dP_pano))
        O.flush_dw()                           # deferred weight-gradient GEMMs, ~8 problems per launch
        O.join_side()                          # (opt-in) weight-gradient GEMMs forked to the side stream
        self._ctx = None

    def _as(self, d):
        """fp32 [B,B] similarity gradient -> compute dtype operand for the MFMA GEMM"""
        return d if d.dtype == self.compute_dtype else O.cast_to(d, self.compute_dtype)
