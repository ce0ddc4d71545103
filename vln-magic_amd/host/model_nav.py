"""`VLNBert(args, role)` -- the navigation-time model the reference imports from its withheld `models.model`
(map_nav_src/r2r/agent.py:30; ctor :36-38) -- on the HIP engine.

Call surface kept:  `vln_bert(mode, inputs)` with mode in
  'language'   -> (txt_embeds [B,L,H], txt_attns [B,h,L,L])                               agent.py:796
  'panorama'   -> (pano_embeds [B,V,H], pano_masks [B,V], pano_fused_embeds [B,H], img_attns [B,V,V])   :885
  'navigation' -> dict(gmap_embeds, vp_embeds, gmap_attns, vp_attns, cls_embeds,
                       global_logits, local_logits, fused_logits)                          :964-967
  'instr_zdict_update'   -> (txt_embeds [B,L,H], txt_attns)  from inputs z_txt / z_txt_mask          :1231-1233
  'extract_cfp_features' -> dict(txt_outputs, vp_outputs, gmap_outputs)  [B,H] each, whole-trajectory forward   :1538-1541
with the causal-intervention inputs of SURVEY section 8 f-4 (instr_z_* / z_img_* back-door dictionaries, front_* front-door
dictionaries) served by host/causal.py when the matching do_back_* / do_front_* switch is on; such an input with its switch
off raises (never ignored)
plus `.vln_bert.<kd head>` callables (`txt_emb_w`, `kdl_img_w`, `kdl_avg_img_w`, `global_cross_w`,
`local_cross_w`; agent.py:568-665, agent_base.py:330), `.drop_env` (:738), `.parameters()`, `.state_dict()`.

Every mode is ONE torch.autograd.Function: forward runs the engine segment, backward runs the hand-written
backward and returns gradients for the tensor inputs (txt_embeds, gmap_img_embeds, vp_img_embeds) so the
agent's Python between the calls (GraphMap bookkeeping, compute_kd_losses, CE) composes with autograd unchanged;
parameter gradients land directly in `param.grad` (views of the flat gradient buffer).
"""
import os
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import lanes
from . import ops as O
from .config import cfg_get, make_config
from .ddp import refuse_torch_ddp
from .engine import Ctx, MagicNet, cls_specs, trunk_specs
from .params import ParamStore

NAV_DEFER_DW = not os.environ.get("MAGIC_NAV_NO_DEFER_DW")
KD_HEADS = ("txt_emb_w", "kdl_img_w", "kdl_avg_img_w", "global_cross_w", "local_cross_w")
# back-door / front-door inputs the agent passes as None unless args.do_back_* / do_front_* are set (agent.py:76-89,:162-172,:942-944,:1212-1227)
CAUSAL_KEYS = ("instr_z_direction_features", "instr_z_direction_pzs", "instr_z_landmark_features", "instr_z_landmark_pzs",
               "front_txt_feats", "front_vp_feats", "front_gmap_feats", "z_img_features", "z_img_pzs")


def nav_specs(cfg, p="vln_bert."):
    H = cfg.hidden_size
    s = trunk_specs(cfg, p)
    s += cls_specs(p + "global_sap_head.", H) + cls_specs(p + "local_sap_head.", H) + cls_specs(p + "sap_fuse_linear.", H, 2 * H)
    for k in ("txt", "img", "local", "global", "predict"):       # learned ability weights, agent.py:1130-1134
        s.append((f"{p}kdl_{k}_weight", (1,), "zeros"))
    from .causal import causal_specs
    return s + causal_specs(cfg, p)


_lane_bwd = lanes.lane_bwd        # backward of a Function that writes parameter gradients re-enters the lane its forward ran in (`ctx.lane`)


class _HipLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, anchor=None):
        ctx.lane = lanes.cur
        net, lin = mod._net, mod._lin
        shp = x.shape
        M = x.numel() // shp[-1]
        xc = x.detach().to(net.dtype).reshape(M, shp[-1]).contiguous()
        y = O.linear_fwd(xc, lin.W, lin.b, M)
        ctx.mod, ctx.x, ctx.shape, ctx.in_dtype = mod, xc, shp, x.dtype
        return y.view(*shp[:-1], lin.N)

    @staticmethod
    @_lane_bwd
    def backward(ctx, dy):
        net, lin = ctx.mod._net, ctx.mod._lin
        M = ctx.x.shape[0]
        d = dy.to(net.dtype).reshape(M, lin.N).contiguous()
        net.S.ensure_grads()
        owner = getattr(ctx.mod, "_owner", None)
        if owner:
            _queue_sync(owner[0])
        if net.train:
            O.linear_dw(d, ctx.x, lin.dW, lin.db, M)
        dx = O.linear_dx(d, lin.W, M)
        return dx.view(ctx.shape).to(ctx.in_dtype), None, None


class _HipLinearMultiFn(torch.autograd.Function):
    """several independent HipLinear modules applied in one autograd node: `ys = _HipLinearMultiFn.apply(mods, *xs)` -- the forward GEMMs leave as
    ONE grouped launch (lib.group), so do the input-gradient GEMMs; the weight gradients join the deferred queue as usual.  The MAKD step applies
    its four or five projection heads this way (host/makd_nav.compute_kd_losses_fused)."""

    @staticmethod
    def forward(ctx, mods, *xs):
        from . import lib as L
        ctx.lane = lanes.cur
        xcs, shapes = [], []
        for mod, x in zip(mods, xs):
            net = mod._net
            shp = x.shape
            M = x.numel() // shp[-1]
            xcs.append(x.detach().to(net.dtype).reshape(M, shp[-1]).contiguous())
            shapes.append(shp)
        with L.group():
            ys = [O.linear_fwd(xc, mod._lin.W, mod._lin.b, xc.shape[0]) for mod, xc in zip(mods, xcs)]
        ctx.mods, ctx.xcs, ctx.shapes, ctx.in_dtypes = mods, xcs, shapes, [x.dtype for x in xs]
        ctx.set_materialize_grads(False)
        return tuple(y.view(*shp[:-1], mod._lin.N) for y, shp, mod in zip(ys, shapes, mods))

    @staticmethod
    @_lane_bwd
    def backward(ctx, *dys):
        from . import lib as L
        mods = ctx.mods
        mods[0]._net.S.ensure_grads()
        owner = getattr(mods[0], "_owner", None)
        if owner:
            _queue_sync(owner[0])
        ds = []
        for mod, xc, dy in zip(mods, ctx.xcs, dys):
            if dy is None:
                ds.append(None)
                continue
            d = dy.to(mod._net.dtype).reshape(xc.shape[0], mod._lin.N).contiguous()
            if mod._net.train:
                O.linear_dw(d, xc, mod._lin.dW, mod._lin.db, xc.shape[0])
            ds.append(d)
        with L.group():
            dxs = [None if d is None else O.linear_dx(d, mod._lin.W, xc.shape[0]) for mod, xc, d in zip(mods, ctx.xcs, ds)]
        return (None,) + tuple(None if dx is None else dx.view(shp).to(dt) for dx, shp, dt in zip(dxs, ctx.shapes, ctx.in_dtypes))


def hip_linear_multi(mods, xs):
    """[mod(x) for mod, x in zip(mods, xs)] for HipLinear modules, as one autograd node with grouped GEMM launches; anything else is applied one by one"""
    if len(mods) > 1 and all(isinstance(m, HipLinear) for m in mods):
        return list(_HipLinearMultiFn.apply(tuple(mods), *xs))
    return [m(x) for m, x in zip(mods, xs)]


class HipLinear(nn.Module):
    """A Linear whose weight/bias live in the ParamStore; callable like nn.Linear (KD projection heads)."""

    def forward(self, x):
        # an input that carries no gradient (a dictionary, a constant) would leave this node out of the autograd graph and the
        # weight gradient unwritten: the owner's anchor (a requires_grad scalar) keeps the node in
        owner = getattr(self, "_owner", None)
        anchor = getattr(owner[0], "_anchor", None) if (owner and not x.requires_grad and torch.is_grad_enabled()) else None
        return _HipLinearFn.apply(x, self, anchor)


class Critic(nn.Module):
    """`Critic(args)` of the withheld models/model.py (agent.py:30,39): the A2C value head the agent constructs, puts in
    `self.models` / `critic_optimizer` (agent_base.py:116-139) and never calls (`train_rl` is False in every shipped script,
    agent_base.py:241-250).  [LINEAGE DUET] state2value = Linear(H,512) -> ReLU -> Dropout(args.dropout) -> Linear(512,1).
    The H->512 projection runs on the HIP GEMM (HipLinear), the 512->1 dot on the row-dot launch (csrc/rowops.hip magic_rowgate, mode 0)."""

    def __init__(self, args, device="cuda", compute_dtype=torch.bfloat16, seed=0):
        super().__init__()
        H = int(getattr(args, "hidden_size", 768))
        specs = [("state2value.0.weight", (512, H), "normal"), ("state2value.0.bias", (512,), "zeros"),
                 ("state2value.3.weight", (1, 512), "normal"), ("state2value.3.bias", (1,), "zeros")]
        self.store = ParamStore(specs, device, compute_dtype, init_std=0.02, seed=seed, requires_grad=True)
        self.store.attach_to(self)
        from .engine import Lin
        from types import SimpleNamespace
        fc = getattr(self.state2value, "0")
        fc.__class__ = HipLinear
        fc._net = SimpleNamespace(dtype=compute_dtype, train=True, S=self.store)
        fc._lin = Lin(self.store, "state2value.0.weight", "state2value.0.bias")
        fc._owner = (self,)
        self.drop = nn.Dropout(p=float(getattr(args, "dropout", 0.5)))
        self.register_load_state_dict_post_hook(lambda m, k: setattr(m.store, "shadow_clean", False))

    def cuda(self, device=None):
        return self

    def forward(self, state):
        self.store.sync_shadow()
        h = self.drop(torch.relu(getattr(self.state2value, "0")(state).float()))
        return _RowDotFn.apply(h, self).squeeze()


class _RowDotFn(torch.autograd.Function):
    """the value head's output Linear(512, 1) as one row-dot launch each way (csrc/rowops.hip magic_rowgate, mode 0); weight / bias gradients land
    in the Critic's flat gradient buffer"""

    @staticmethod
    def forward(ctx, h, critic):
        ctx.lane = lanes.cur
        st = critic.store
        w, b = st.master("state2value.3.weight").view(-1), st.master("state2value.3.bias")
        x = h.detach().reshape(-1, h.shape[-1]).float().contiguous()
        M, H = x.shape
        out = torch.empty(M, dtype=torch.float32, device=x.device)
        O.rowgate_fwd(0, x, M, H, w, b0=b, out_s=out)
        ctx.critic, ctx.x, ctx.shape, ctx.in_dtype = critic, x, h.shape, h.dtype
        return out.view(*h.shape[:-1], 1)

    @staticmethod
    @_lane_bwd
    def backward(ctx, dy):
        st = ctx.critic.store
        st.ensure_grads()
        M, H = ctx.x.shape
        dx = torch.empty_like(ctx.x)
        O.rowgate_bwd(0, ctx.x, M, H, st.master("state2value.3.weight").view(-1), dy=dy.detach().reshape(-1).float().contiguous(), dx=dx,
                      dwx=st.g("state2value.3.weight").view(-1), db0=st.g("state2value.3.bias"))
        return dx.view(ctx.shape).to(ctx.in_dtype), None


class _PassToken:
    """lives exactly as long as the autograd engine keeps this pass's end-of-backward callback"""
    __slots__ = ("__weakref__",)


def _pass_pending(model):
    ref = getattr(model, "_sync_token", None)
    return ref is not None and ref() is not None


def _queue_sync(model):
    """nav mode: many Function.backward calls feed one store per loss.backward(); average the gradients once, when the whole
    autograd pass is over (the engine's end-of-backward callback -- the mechanism DDP's reducer uses too).
    "Already queued for this pass" is a weak reference to a token only the queued callback holds: when a backward pass raises,
    the engine drops its final callbacks, the token dies with them, and the next pass queues afresh (a sticky flag here would
    silently skip the flush and the data-parallel average for every later pass)."""
    if getattr(model, "explicit_backward", False):
        # the pretraining model runs ONE explicit backward per step (model_pretrain.backward): its own deferral / flush / exchange logic
        # owns the gradients; the causal blocks' small autograd islands inside it must not queue an average of their own
        return
    pending = _pass_pending(model)
    if NAV_DEFER_DW:
        # the weight-gradient GEMMs of every Function.backward of this autograd pass are queued (operands kept alive) and leave in a
        # few grouped launches from the end-of-backward callback: ~2 400 single launches per navigator iteration become ~25
        if not pending and not O.DEFER["active"]:
            O.DEFER["queue"].clear(); O.DEFER["bytes"] = 0          # anything left by a pass that raised half-way is stale
            for ref in getattr(model, "_step_graph_sets", ()):
                if ref() is not None:
                    del ref()._cat_used[:]
        O.defer_dw(True)
    if pending:
        return
    tok = _PassToken()
    model._sync_token = weakref.ref(tok)

    def _done(tok=tok):
        model._sync_token = None
        O.join_dw_stream()            # weight-gradient launches of captured step instances (host/step_graphs.py) run on a stream of their own
        lanes.join(getattr(model, "device_", None), forget=False)     # the rollouts' gradient lanes ran on streams of their own (host/lanes.py): this stream waits for them
        for ref in getattr(model, "_step_graph_sets", ()):            # captured step instances: every Linear's weight gradient over all of them, one launch per <= 96
            sg = ref()
            if sg is not None:
                sg.flush_cat()
        if NAV_DEFER_DW:
            O.flush_dw()
        O.flush_rbw_parts()           # partial LayerNorm gradients of the row-block backward launches of this pass (no-op when flush_dw ran)
        model.store.merge_lanes()     # grad += the lanes' buffers, in lane order
        from .trainer import auto_sync
        auto_sync(model)
    torch.autograd.Variable._execution_engine.queue_callback(_done)


def _u8(m):
    """mask -> uint8 without a launch when it is bool / uint8 already (bool and uint8 share their storage layout)"""
    if m.dtype == torch.uint8:
        return m.contiguous()
    if m.dtype == torch.bool:
        return m.contiguous().view(torch.uint8)
    return m.to(torch.uint8).contiguous()


def _i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


def _zeros_like_shape(t, shape, dtype, device):
    return torch.zeros(shape, dtype=dtype, device=device) if t is None else t


class _LanguageFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, txt_ids, txt_masks):
        ctx.lane = lanes.cur
        net = model.net
        B, L = txt_ids.shape
        lens = txt_masks.sum(1).tolist()        # (already a host sync: the range check below rides on it)
        lo, hi = int(txt_ids.min()), int(txt_ids.max())
        if lo < 0 or hi >= net.cfg.vocab_size or L + 2 > net.cfg.max_position_embeddings:
            raise ValueError(f"language input does not fit the model config: ids in [{lo}, {hi}] vs vocab_size {net.cfg.vocab_size}, "
                             f"{L} tokens vs max_position_embeddings {net.cfg.max_position_embeddings}")
        plan = dict(B=B, L=L, txt_ids=txt_ids.reshape(-1).to(torch.int32), txt_mask=txt_masks.to(torch.uint8).contiguous(),
                    lens=dict(txt=lens), txt_tokens=int(sum(lens)))
        c = net.text_fwd(plan)
        ctx.model, ctx.c, ctx.plan, ctx.drop = model, c, plan, net.drop
        ctx.set_materialize_grads(False)
        return c.out.view(B, L, net.H), c.P[..., :L]

    @staticmethod
    @_lane_bwd
    def backward(ctx, d_out, d_attn):
        net, c = ctx.model.net, ctx.c
        net.S.ensure_grads()
        _queue_sync(ctx.model)
        B, L = c.B, c.L
        d = torch.zeros(B * L, net.H, dtype=net.dtype, device=c.out.device) if d_out is None else d_out.to(net.dtype).reshape(B * L, net.H).contiguous()
        dP = None
        if d_attn is not None:
            dP = torch.zeros(B, net.nh, L, c.ldp, dtype=torch.float32, device=c.out.device)
            dP[..., :L] = d_attn.float()
        cur = net.drop
        net.drop = ctx.drop                 # the row-block backward regenerates the masks of THIS call (later forwards re-armed net.drop)
        try:
            net.text_bwd(c, ctx.plan, d, dP)
        finally:
            net.drop = cur
        return None, None, None, None


def pano_forward_body(model, view_img_fts, loc_fts, nav_types, view_lens, pano_masks=None):
    """the panorama segment on the engine: returns (ctx namespace, plan, (pano_embeds, masks, fused, img_attns)).  Plain function: the
    eager autograd Function below and the captured step instances (host/step_graphs.py) both run it."""
    net = model.net
    B, V, D = view_img_fts.shape
    dev = view_img_fts.device
    if pano_masks is None:          # index-plan rollouts hand the mask over with the step's other index arrays (no launches here)
        pano_masks = torch.arange(V, device=dev)[None] < view_lens[:, None]
    plan = dict(Np=B, V=V, nav_types=_i32(nav_types.reshape(-1)), view_lens=_i32(view_lens), pano_mask=_u8(pano_masks))
    if view_img_fts.dtype == net.dtype:     # gathered from the HBM feature table in the compute dtype already
        feats = view_img_fts.detach().reshape(B * V, D).contiguous()
    else:
        feats = O.cast_to(view_img_fts.detach().float().reshape(B * V, D).contiguous(), net.dtype)
    c = net.pano_fwd(plan, feats, loc_fts.detach().float().reshape(B * V, -1).contiguous())
    c.drop = net.drop                       # the backward's row-block launches regenerate THIS call's masks (a later forward re-arms net.drop)
    masks = plan["pano_mask"].view(torch.bool)
    return c, plan, (c.out.view(B, V, net.H), masks, c.fused, c.img_attn[..., :V])


def pano_backward_body(model, c, plan, d_emb, d_fused, d_attn):
    net = model.net
    Np, V, H = c.Np, c.V, net.H
    dev = c.out.device
    d_pano = torch.zeros(Np * V, H, dtype=net.dtype, device=dev) if d_emb is None else d_emb.to(net.dtype).reshape(Np * V, H).clone()
    df = None if d_fused is None else d_fused.to(net.dtype).contiguous()
    dP = None
    if d_attn is not None:
        g = torch.zeros(Np, V, c.ldp, dtype=torch.float32, device=dev)
        g[..., :V] = d_attn.float()
        dP = torch.empty(Np, net.nh, V, c.ldp, dtype=torch.float32, device=dev)
        O.head_mean_bwd(g, dP, Np, net.nh, V * c.ldp)
    cur = net.drop
    net.drop = getattr(c, "drop", cur)
    try:
        net.pano_bwd(c, plan, d_pano, df, dP)
    finally:
        net.drop = cur


class _PanoramaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, view_img_fts, loc_fts, nav_types, view_lens, pano_masks=None):
        ctx.lane = lanes.cur
        c, plan, outs = pano_forward_body(model, view_img_fts, loc_fts, nav_types, view_lens, pano_masks)
        ctx.model, ctx.c, ctx.plan = model, c, plan
        ctx.set_materialize_grads(False)          # an output the loss does not use arrives as None in backward, not as a zero tensor to push through
        ctx.mark_non_differentiable(outs[1])
        return outs

    @staticmethod
    @_lane_bwd
    def backward(ctx, d_emb, _dm, d_fused, d_attn):
        ctx.model.net.S.ensure_grads()
        _queue_sync(ctx.model)
        pano_backward_body(ctx.model, ctx.c, ctx.plan, d_emb, d_fused, d_attn)
        return None, None, None, None, None, None, None


def nav_fusion_plan(gmap_vpids, gmap_visited_masks, vp_cand_vpids, K, Vp):
    """local->global fusion map for the navigation step ([LINEAGE] DUET nav forward): gmap_vpids[b] =
    [None, (None mem)?, visited..., unvisited...]; vp_cand_vpids[b][j] is the viewpoint of local token j (None for
    [stop]/[mem]/non-candidate views)."""
    B = len(gmap_vpids)
    fsrc = np.full((B, K), -1, np.int32)
    bw = np.zeros((B, Vp), np.uint8)
    vm = gmap_visited_masks.cpu().numpy()
    for b in range(B):
        visited = set(vp for j, vp in enumerate(gmap_vpids[b]) if vp is not None and vm[b, j])
        tmp = {}
        for j, c in enumerate(vp_cand_vpids[b]):
            if c is None or j == 0:
                continue
            if c in visited:
                bw[b, j] = 1
            else:
                tmp[c] = j
        fsrc[b, 0] = 0
        for j, vp in enumerate(gmap_vpids[b]):
            if j > 0 and vp is not None and vp not in visited:
                fsrc[b, j] = tmp[vp] if vp in tmp else -2
    return fsrc, bw


def _kv_lins(model):
    net, p, nl = model.net, model.prefix, model.net.cfg.num_x_layers
    H = net.H
    return [net.lin(f"{p}{enc}encoder.crossattention.{i}.crossattention.self.key.weight",
                    f"{p}{enc}encoder.crossattention.{i}.crossattention.self.key.bias", rows=2 * H, cols=H)
            for enc in ("global_encoder.", "local_encoder.") for i in range(nl)]


class _TextKVFn(torch.autograd.Function):
    """key/value projections of the instruction for every cross-attention layer of both encoders, [2 * num_x_layers, B*L, 2H].
    The text embeddings are fixed for a whole episode (agent.py:796 computes them once), but the reference's per-step
    `vln_bert('navigation', ...)` call re-projects them in all six cross layers at every step -- at RxR lengths that is most of
    a step's FLOPs.  Computed once per episode here; each step's navigation backward returns its dK/dV for this tensor,
    autograd sums them, and this backward runs the projection's weight / input gradients ONCE on the sum."""

    @staticmethod
    def forward(ctx, anchor, model, txt_embeds, out=None):
        ctx.lane = lanes.cur
        net = model.net
        B, L, H = txt_embeds.shape
        M = B * L
        txt = txt_embeds.detach().to(net.dtype).reshape(M, H).contiguous()
        lins = _kv_lins(model)
        if out is None:
            out = net.new(len(lins), M, 2 * H)
        else:                              # a static home (host/step_graphs.TextSlot): captured step graphs read the cache at a fixed address
            if tuple(out.shape) != (len(lins), M, 2 * H) or out.dtype != net.dtype or not out.is_contiguous():
                raise ValueError(f"text_kv: out must be a contiguous {net.dtype} tensor of shape {(len(lins), M, 2 * H)}")
            out = out.detach()
        for i, kvl in enumerate(lins):
            O.linear_fwd(txt, kvl.W, kvl.b, M, out=out[i])
        ctx.model, ctx.txt, ctx.shape, ctx.in_dtype = model, txt, (B, L, H), txt_embeds.dtype
        return out

    @staticmethod
    @_lane_bwd
    def backward(ctx, dkv):
        model, net = ctx.model, ctx.model.net
        net.S.ensure_grads()
        _queue_sync(model)
        B, L, H = ctx.shape
        M = B * L
        dkv = dkv.to(net.dtype).contiguous()
        d_txt = torch.zeros(M, H, dtype=net.dtype, device=dkv.device)
        for i, kvl in enumerate(_kv_lins(model)):
            O.linear_dw(dkv[i], ctx.txt, kvl.dW, kvl.db, M)
            O.linear_dx(dkv[i], kvl.W, M, out=d_txt, residual=d_txt)
        return None, None, d_txt.view(B, L, H).to(ctx.in_dtype), None


class _fork:
    """`with _fork(stream) as side: with side: <branch B>; <branch A>`: branch B runs on `stream` (forked from the current stream at entry,
    joined at exit); stream None: both branches on the current stream, in program order"""

    def __init__(self, stream):
        self.s = stream

    def __enter__(self):
        import contextlib
        if self.s is None:
            return contextlib.nullcontext()
        self.s.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self.s)

    def __exit__(self, et, ev, tb):
        if self.s is not None:
            torch.cuda.current_stream().wait_stream(self.s)
        return False


# measured (profiles/micro/r05_ab_nav_pair.txt, same box, bench_nav.py 10 + 10 iterations): MAGIC-L navigator iteration 142.7 -> 130.4 ms, ICoD iteration
# 133.5 -> 108.0 ms against the forked form (the same launches as two parallel branches of the step's graph)
NAV_PAIR = os.environ.get("MAGIC_NAV_PAIR", "1") == "1"


def _pair_branches(fork):
    """pair the two cross-modal encoders' launches (lib.lockstep) instead of forking one onto a side stream: only where a fork was offered (the
    captured step instances: the branches are independent there) and only inside a capture (eagerly the per-call rendezvous of two host threads
    costs more than the launches it saves)"""
    return NAV_PAIR and fork is not None and torch.cuda.is_current_stream_capturing()


def nav_forward_body(model, gmap_img, vp_img, txt_embeds, b, txt_kv=None):
    """the navigation segment (both cross-modal encoders + heads + logit fusion) on the engine: returns (ctx namespace, outputs).  Plain
    function: run by the eager autograd Function below and by the captured step instances (host/step_graphs.py)."""
    net, p = model.net, model.prefix
    H = net.H
    dev = gmap_img.device
    B, K, _ = gmap_img.shape
    Vp = vp_img.shape[1]
    L = txt_embeds.shape[1]
    txt_masks = b["txt_masks"]
    hl = b.get("host_lens")          # index-plan rollout: the host already knows every length (no device->host sync)
    tl, gl_, vl = hl if hl is not None else (txt_masks.sum(1).tolist(), b["gmap_masks"].sum(1).tolist(), b["vp_masks"].sum(1).tolist())
    plan = dict(B=B, K=K, Vp=Vp, L=L, gmap_step_ids=_i32(b["gmap_step_ids"].reshape(-1)))
    c = Ctx(plan=plan, B=B, K=K, Vp=Vp, L=L)
    txt = txt_embeds.detach().to(net.dtype).reshape(B * L, H).contiguous()
    tmask, gmask_u8, vmask_u8 = _u8(txt_masks), _u8(b["gmap_masks"]), _u8(b["vp_masks"])
    c.gin, c.vin = net.nodes_in_fwd(plan, None, b["gmap_pos_fts"].float().reshape(B * K, -1).contiguous(),
                                    b["vp_pos_fts"].float().reshape(B * Vp, -1).contiguous(),
                                    gimg=gmap_img.detach().to(net.dtype).reshape(B * K, H).contiguous(),
                                    vimg=vp_img.detach().to(net.dtype).reshape(B * Vp, H).contiguous())
    nl = net.cfg.num_x_layers
    kv = None if txt_kv is None else txt_kv.detach()
    c.has_kv = kv is not None
    # the two cross-modal encoders are independent until the heads: on a side stream when the caller provides one (`b["fork"]`: the captured
    # step instances -- inside a HIP graph the two become parallel branches, and a 600-row GEMM leaves half of the chip to the other encoder)
    dist_f = b["gmap_pair_dists"].float().contiguous()
    if _pair_branches(b.get("fork")):
        # ... or PAIRED: the two encoders have the same launch sequence, so inside a capture two host threads issue them in lockstep and
        # every groupable launch (GEMM, LayerNorm, attention) of the one goes out together with its twin of the other as ONE kernel
        from . import lib as _L
        c.glob, c.loc = _L.lockstep(
            lambda: net.cross_fwd("global", plan, c.gin.out, K, gmask_u8, gl_, int(sum(gl_)), txt, L, tmask, tl, int(sum(tl)),
                                  dist=dist_f, kv=None if kv is None else kv[:nl]),
            lambda: net.cross_fwd("local", plan, c.vin.out, Vp, vmask_u8, vl, int(sum(vl)), txt, L, tmask, tl, int(sum(tl)),
                                  kv=None if kv is None else kv[nl:]))
    else:
        with _fork(b.get("fork")) as side:
            with side:
                c.loc = net.cross_fwd("local", plan, c.vin.out, Vp, vmask_u8, vl, int(sum(vl)), txt, L, tmask, tl, int(sum(tl)),
                                      kv=None if kv is None else kv[nl:])
            c.glob = net.cross_fwd("global", plan, c.gin.out, K, gmask_u8, gl_, int(sum(gl_)), txt, L, tmask, tl, int(sum(tl)),
                                   dist=dist_f, kv=None if kv is None else kv[:nl])
    # heads
    c.Yg, c.g_raw = model._cls(p + "global_sap_head.", c.glob.out, B * K)
    c.Yl, c.l_raw = model._cls(p + "local_sap_head.", c.loc.out, B * Vp)
    c.use_gate = bool(cfg_get(net.cfg, "glocal_fuse"))
    if c.use_gate:
        f1 = net.lin(p + "sap_fuse_linear.net.0.weight")
        tmp = O.linear_fwd(c.glob.out, f1.W, None, B, lda=K * H, ldb=2 * H, K=H)
        c.Yf = O.linear_fwd(c.loc.out, f1.W[:, H:], f1.b, B, lda=Vp * H, ldb=2 * H, K=H, residual=tmp)
        O.dact(c.Yf, c.Yf, 2, out=c.Yf)
        fln, f2 = net.ln(p + "sap_fuse_linear.net.2"), net.lin(p + "sap_fuse_linear.net.3.weight")
        c.fuse_raw = net.new(B, dtype=torch.float32)
        O.lndot_fwd(c.Yf, B, H, fln.g, fln.b, net.eps, f2.Wm, f2.b, c.fuse_raw)
    else:
        c.fuse_raw = net.zeros(B, dtype=torch.float32)
    if b.get("gmap_logit_masks") is not None:      # ~visited & valid, built with the step's other index arrays (host/nav_plan.py)
        c.gmask = _u8(b["gmap_logit_masks"])
    else:
        c.gmask = _u8((~b["gmap_visited_masks"]) & b["gmap_masks"])
    c.lmask = _u8(b["vp_nav_masks"])
    if b.get("fusion") is not None:   # (fsrc int32 [B,K], bw uint8 [B,Vp]) already on the device (host/nav_plan.fusion_map)
        c.fsrc, c.bw = b["fusion"]
    else:
        fsrc, bw = nav_fusion_plan(b["gmap_vpids"], b["gmap_visited_masks"], b["vp_cand_vpids"], K, Vp)
        c.fsrc, c.bw = torch.from_numpy(fsrc).to(dev), torch.from_numpy(bw).to(dev)
    gl, ll, fl = net.new(B, K, dtype=torch.float32), net.new(B, Vp, dtype=torch.float32), net.new(B, K, dtype=torch.float32)
    O.sap_fuse_fwd(B, K, Vp, c.g_raw, c.l_raw, c.fuse_raw, c.gmask, c.lmask, c.fsrc, c.bw, c.use_gate, gl, ll, fl)
    # cls_embeds: the [stop]-token summary the agent feeds back as the [MEM] token (agent.py:206-210); open choice O9
    cls = net.new(B, H)
    O.csr_gather(c.glob.out, *model._first_rows(B, K, dev), cls, B, H)
    O.csr_gather(c.loc.out, *model._first_rows(B, Vp, dev), cls, B, H, accumulate=True)
    c.drop = net.drop
    return c, (c.glob.out.view(B, K, H), c.loc.out.view(B, Vp, H), c.glob.P[..., :L], c.loc.P[..., :L], cls, gl, ll, fl)


def nav_backward_body(model, c, d_g, d_v, d_ga, d_va, d_cls, dgl, dll, dfl, dkv_acc=None, fork=None):
    """returns (d_gmap_img [B*K, H], d_vp_img [B*Vp, H], d_txt [B*L, H] or None, dkv [2 nl, B*L, 2H] or None).  dkv_acc: the episode's
    accumulator of the cached K/V projection's gradient [2 nl, B*L, 2H]: this step's dK / dV are added to it and None is returned for dkv."""
    net, p = model.net, model.prefix
    B, K, Vp, L, H = c.B, c.K, c.Vp, c.L, net.H
    dev = c.glob.out.device
    T = net.dtype
    d_gmap = torch.zeros(B * K, H, dtype=T, device=dev) if d_g is None else d_g.to(T).reshape(B * K, H).clone()
    d_vp = torch.zeros(B * Vp, H, dtype=T, device=dev) if d_v is None else d_v.to(T).reshape(B * Vp, H).clone()
    d_txt = torch.zeros(B * L, H, dtype=T, device=dev)
    if d_cls is not None:
        dc = d_cls.to(T).contiguous()
        O.csr_gather(dc, *model._first_rows_T(B, K, dev), d_gmap, B * K, H, accumulate=True)
        O.csr_gather(dc, *model._first_rows_T(B, Vp, dev), d_vp, B * Vp, H, accumulate=True)
    if dgl is not None or dll is not None or dfl is not None:
        f32 = lambda t: None if t is None else torch.nan_to_num(t.float(), nan=0.0, posinf=0.0, neginf=0.0).contiguous()
        dg, dl, df = net.new(B, K, dtype=torch.float32), net.new(B, Vp, dtype=torch.float32), net.new(B, dtype=torch.float32)
        O.sap_fuse_bwd(B, K, Vp, c.g_raw, c.l_raw, c.fuse_raw, c.gmask, c.lmask, c.fsrc, c.bw, c.use_gate, f32(dgl), f32(dll), f32(dfl), dg, dl, df)
        model._cls_bwd(p + "global_sap_head.", c.glob.out, c.Yg, dg, B * K, d_gmap)
        model._cls_bwd(p + "local_sap_head.", c.loc.out, c.Yl, dl, B * Vp, d_vp)
        if c.use_gate:
            f1, fln, f2 = net.lin(p + "sap_fuse_linear.net.0.weight"), net.ln(p + "sap_fuse_linear.net.2"), net.lin(p + "sap_fuse_linear.net.3.weight")
            dZ = net.new(B, H)
            O.lndot_bwd(c.Yf, B, H, fln.g, fln.b, net.eps, f2.Wm, df, dZ, fln.dg, fln.db, f2.dW, f2.db)
            O.linear_dw(dZ, c.glob.out, f1.dW, f1.db, B, N=H, K=H, ldb=K * H, ldc=2 * H)
            O.linear_dw(dZ, c.loc.out, f1.dW[:, H:], None, B, N=H, K=H, ldb=Vp * H, ldc=2 * H)
            O.linear_dx(dZ, f1.W, B, out=d_gmap, residual=d_gmap, ldb=2 * H, ldc=K * H, N=H, K=H)
            O.linear_dx(dZ, f1.W[:, H:], B, out=d_vp, residual=d_vp, ldb=2 * H, ldc=Vp * H, N=H, K=H)

    def attn_seed(d, P, Nq):
        if d is None:
            return None
        dP = torch.zeros(B, net.nh, Nq, P.shape[-1], dtype=torch.float32, device=dev)
        dP[..., :L] = d.float()
        return dP
    nl = net.cfg.num_x_layers
    acc = dkv_acc is not None and c.has_kv
    dkv = dkv_acc if acc else (net.new(2 * nl, B * L, 2 * H) if c.has_kv else None)
    cur = net.drop
    net.drop = getattr(c, "drop", cur)      # the row-block backward launches regenerate the masks of THIS call's forward
    try:
        if fork is not None and not c.has_kv:
            fork = None                   # (without the K/V cache both encoders add into d_txt: keep them in order)
        sga, sva = attn_seed(d_ga, c.glob.P, K), attn_seed(d_va, c.loc.P, Vp)
        if _pair_branches(fork):
            from . import lib as _L

            def b_loc():
                d = net.cross_bwd(c.loc, d_vp, d_txt, sva, dkv=None if dkv is None else dkv[nl:], acc_kv=acc)
                net.vp_in_bwd(c.vin, c.plan, d, None)
                return d

            def b_glob():
                d = net.cross_bwd(c.glob, d_gmap, d_txt, sga, dkv=None if dkv is None else dkv[:nl], acc_kv=acc)
                net.gmap_in_bwd(c.gin, c.plan, d, None, None)
                return d
            d_gin, d_vin = _L.lockstep(b_glob, b_loc, side=fork)
        else:
            with _fork(fork) as side:
                with side:
                    d_vin = net.cross_bwd(c.loc, d_vp, d_txt, sva, dkv=None if dkv is None else dkv[nl:], acc_kv=acc)
                    net.vp_in_bwd(c.vin, c.plan, d_vin, None)
                d_gin = net.cross_bwd(c.glob, d_gmap, d_txt, sga, dkv=None if dkv is None else dkv[:nl], acc_kv=acc)
                net.gmap_in_bwd(c.gin, c.plan, d_gin, None, None)
    finally:
        net.drop = cur
    return d_gin, d_vin, (None if c.has_kv else d_txt), (None if acc else dkv)


class _NavigationFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, gmap_img, vp_img, txt_embeds, b, txt_kv=None):
        ctx.lane = lanes.cur
        c, outs = nav_forward_body(model, gmap_img, vp_img, txt_embeds, b, txt_kv)
        ctx.model, ctx.c = model, c
        # an output the loss does not use (the attention maps without attention distillation, the embeddings) arrives as None in backward: with
        # materialised zeros every step pushed a zero dP through the UNFUSED attention backward of both top layers (found in round 5's profile)
        ctx.set_materialize_grads(False)
        ctx.in_dtypes = (gmap_img.dtype, vp_img.dtype, txt_embeds.dtype)
        return outs

    @staticmethod
    @_lane_bwd
    def backward(ctx, d_g, d_v, d_ga, d_va, d_cls, dgl, dll, dfl):
        model, c = ctx.model, ctx.c
        model.net.S.ensure_grads()
        _queue_sync(model)
        B, K, Vp, L, H = c.B, c.K, c.Vp, c.L, model.net.H
        d_gin, d_vin, d_txt, dkv = nav_backward_body(model, c, d_g, d_v, d_ga, d_va, d_cls, dgl, dll, dfl)
        a, b_, t_ = ctx.in_dtypes
        return (None, None, d_gin.view(B, K, H).to(a), d_vin.view(B, Vp, H).to(b_), None if d_txt is None else d_txt.view(B, L, H).to(t_), None, dkv)


class VLNBert(nn.Module):
    def __init__(self, args, role="student", device="cuda", compute_dtype=torch.bfloat16, seed=0, config=None):
        """args: the agent's argparse namespace (map_nav_src/r2r/parser.py) or None with an explicit `config`."""
        super().__init__()
        if config is None:
            hs = getattr(args, "teacher_hidden_size" if role == "teacher" else "hidden_size", 768)
            t_hs = getattr(args, "teacher_hidden_size", None) if role != "teacher" else None
            config = make_config(hs, role=role, teacher_hidden_size=t_hs if getattr(args, "train_kdl", False) else None,
                                 num_l_layers=getattr(args, "num_l_layers", 6), num_x_layers=getattr(args, "num_x_layers", 3),
                                 num_pano_layers=getattr(args, "num_pano_layers", 2), graph_sprels=getattr(args, "graph_sprels", True),
                                 **{k: getattr(args, k) for k in ("do_back_txt", "do_back_img", "do_front_txt", "do_front_img", "do_front_his",
                                                                   "do_back_txt_type", "do_back_imgobj_type", "do_back_img_type", "do_add_method",
                                                                   "front_n_clusters") if hasattr(args, k)})
        self.config, self.role, self.prefix = config, role, "vln_bert."
        self.device_, self.compute_dtype = torch.device(device), compute_dtype
        trainable = role != "teacher" or bool(getattr(args, "train_kdl_teacher", False))
        self.store = ParamStore(nav_specs(config), device, compute_dtype, init_std=cfg_get(config, "initializer_range"),
                                seed=seed, requires_grad=trainable)
        self.store.attach_to(self)
        self.net = MagicNet(config, self.store, self.prefix)
        from .model_pretrain import _dropout_knobs
        _dropout_knobs(self, config)
        self.drop_env = nn.Dropout(p=getattr(args, "feat_dropout", 0.0))      # agent.py:738
        self._anchor = torch.zeros(1, device=self.device_, requires_grad=True)
        self._rows = {}
        if getattr(config, "teacher_hidden_size", None):
            for n in KD_HEADS:
                m = getattr(self.vln_bert, n)
                m.__class__ = HipLinear
                m._net, m._lin = self.net, self.net.lin(f"{self.prefix}{n}.weight")
                m._owner = (self,)          # tuple: not registered as a sub-module
        if self.device_.type == "cuda":
            O.dw_counters(self.device_)
        from .causal import build_blocks
        self.causal_blocks = build_blocks(self)          # plain dict: the blocks' parameters live in the store like every other
        self.register_load_state_dict_post_hook(lambda m, k: setattr(m.store, "shadow_clean", False))

    def cuda(self, device=None):           # `VLNBert(args, role).cuda()` (agent.py:36-38): already resident
        return self

    # shared head helpers (same kernels as the pretraining model)
    from .model_pretrain import GlocalTextPathCMTPreTraining as _P
    _cls, _cls_bwd, _arm_dropout, _dev, _inputs = _P._cls, _P._cls_bwd, _P._arm_dropout, _P._dev, _P._inputs
    del _P

    def _first_rows(self, B, N, dev):
        key = ("f", B, N)
        if key not in self._rows:
            from .plan import csr_pair
            f, t = csr_pair([(b, b * N, 1.0) for b in range(B)], B, B * N)
            self._rows[key] = tuple(torch.from_numpy(a).to(dev) for a in f)
            self._rows[("t", B, N)] = tuple(torch.from_numpy(a).to(dev) for a in t)
        return self._rows[key]

    def _first_rows_T(self, B, N, dev):
        self._first_rows(B, N, dev)
        return self._rows[("t", B, N)]

    def text_kv(self, txt_embeds, out=None):
        """per-episode key/value projections of the instruction for the navigation steps: pass the result as
        `inputs['txt_kv']` of every `vln_bert('navigation', inputs)` call of the episode (optional; identical results)"""
        self.store.sync_shadow()
        return _TextKVFn.apply(self._anchor, self, txt_embeds, out)

    def forward(self, mode, batch):
        refuse_torch_ddp(self)
        from .causal import check_inputs
        check_inputs(self, mode, batch, CAUSAL_KEYS)
        cz = self.causal_blocks
        self.store.sync_shadow()
        self._arm_dropout()           # vln_bert.train() (agent.py:rollout under feedback='sample') -> config dropouts on
        if mode in ("language", "instr_zdict_update"):
            # instr_zdict_update (agent.py:1231-1233, update_z_dict): per-token instruction embeddings under no_grad; the caller indexes `[0][b][j + 1]`
            ids, masks = (batch["txt_ids"], batch["txt_masks"]) if mode == "language" else (batch["z_txt"], batch["z_txt_mask"])
            x, attns = _LanguageFn.apply(self._anchor, self, ids, masks)
            if "back_txt" in cz and (batch.get("instr_z_direction_features") is not None or batch.get("instr_z_landmark_features") is not None):
                # direction rows first, then landmark rows; a landmark-only dictionary (zdict.load_instr_tensor without direction entries) is used as is
                parts = [k for k in ("direction", "landmark") if batch.get(f"instr_z_{k}_features") is not None]
                z = torch.cat([batch[f"instr_z_{k}_features"] for k in parts], 1)
                pz = torch.cat([batch[f"instr_z_{k}_pzs"] for k in parts], 1)
                x = cz["back_txt"](x, z, pz)
            if "front_txt" in cz and batch.get("front_txt_feats") is not None:
                x = cz["front_txt"](x, batch["front_txt_feats"])
            return x, attns
        if mode == "panorama":
            fts = batch["view_img_fts"]
            if not batch.get("already_dropout", True):
                fts = self.drop_env(fts)
            x, masks, fused, attns = _PanoramaFn.apply(self._anchor, self, fts, batch["loc_fts"], batch["nav_types"], batch["view_lens"], batch.get("pano_masks"))
            if "back_img" in cz and batch.get("z_img_features") is not None:
                x = cz["back_img"](x, batch["z_img_features"], batch["z_img_pzs"])
            return x, masks, fused, attns
        if mode == "navigation":
            if "front_gmap" in cz and batch.get("front_gmap_feats") is not None:
                batch = dict(batch, gmap_img_embeds=cz["front_gmap"](batch["gmap_img_embeds"], batch["front_gmap_feats"]))
            if "front_vp" in cz and batch.get("front_vp_feats") is not None:
                batch = dict(batch, vp_img_embeds=cz["front_vp"](batch["vp_img_embeds"], batch["front_vp_feats"]))
            data = {k: batch[k] for k in ("txt_masks", "gmap_masks", "vp_masks", "gmap_step_ids", "gmap_pos_fts", "gmap_pair_dists",
                                          "gmap_visited_masks", "gmap_vpids", "vp_pos_fts", "vp_nav_masks", "vp_cand_vpids")}
            data["host_lens"], data["fusion"], data["gmap_logit_masks"] = batch.get("host_lens"), batch.get("fusion"), batch.get("gmap_logit_masks")
            g, v, ga, va, cls, gl, ll, fl = _NavigationFn.apply(self._anchor, self, batch["gmap_img_embeds"], batch["vp_img_embeds"],
                                                                batch["txt_embeds"], data, batch.get("txt_kv"))
            return dict(gmap_embeds=g, vp_embeds=v, gmap_attns=ga, vp_attns=va, cls_embeds=cls,
                        global_logits=gl, local_logits=ll, fused_logits=fl)
        if mode == "extract_cfp_features":
            return self._extract_cfp_features(batch)
        raise NotImplementedError(f"VLNBert mode {mode!r}")

    @torch.no_grad()
    def _extract_cfp_features(self, batch):
        """Whole-trajectory forward on a `GMapNavAgent.cfp_collate` batch (agent.py:1470-1513; same keys as the pretraining
        cfp collate) -> the first-token features the caller writes to `cfp_features_<iter>.tsv` (agent.py:1535-1541):
        txt_outputs = text [CLS], vp_outputs = local-branch [stop], gmap_outputs = global-branch [stop], each [B, H].
        Same engine segments as the pretraining model's cfp task, without its projection heads (the navigation model has none)."""
        from .plan import build_plan
        n, dev = self.net, self.device_
        big = ("traj_view_img_fts", "traj_loc_fts", "gmap_pos_fts", "gmap_pair_dists", "vp_pos_fts")
        host = {k: (v.cpu() if torch.is_tensor(v) and k not in big else v) for k, v in batch.items()}   # cfp_collate leaves tensors on the GPU
        plan = build_plan(host, "cfp", dev)
        inp = self._inputs(batch, plan)
        B, L, K, Vp, H = plan["B"], plan["L"], plan["K"], plan["Vp"], n.H
        txt = n.text_fwd(plan)
        pano = n.pano_fwd(plan, inp.feats, inp.loc)
        gin = n.gmap_in_fwd(plan, pano, inp.gpos)
        tl, gl_, vl = plan["lens"]["txt"], plan["lens"]["gmap"], [Vp] * B
        glob = n.cross_fwd("global", plan, gin.out, K, plan["gmap_mask"], gl_, plan["gmap_nodes"],
                           txt.out, L, plan["txt_mask"], tl, plan["txt_tokens"], dist=inp.dist)
        vin = n.vp_in_fwd(plan, pano, inp.vpos)
        loc = n.cross_fwd("local", plan, vin.out, Vp, plan["vp_mask"], vl, B * Vp, txt.out, L, plan["txt_mask"], tl, plan["txt_tokens"])
        out = {}
        for name, src, rows in (("txt_outputs", txt.out, "t0"), ("vp_outputs", loc.out, "v0"), ("gmap_outputs", glob.out, "g0")):
            out[name] = n.new(B, H)
            O.csr_gather(src, *plan[rows], out[name], B, H)
        return out
