"""Captured step instances for navigator TRAINING (BASELINE configs 3 and 5).

The fine-tuning iteration (map_nav_src/r2r/agent_base.py:215-296) is two rollouts of up to `max_action_len` steps, each step three model
calls, then ONE `loss.backward()` over all of them: ~9 000 kernel launches of 5-25 us issued from Python one by one -- the loop was bound by
the host (launch + allocation + autograd callbacks: 280 ms of host time for 180 ms of kernels, profiles/micro/nav_kernel_breakdown.py).

Here a step's two heavy segments -- the panorama encoder and the navigation segment (both cross-modal encoders + heads + logit fusion) --
each run as HIP graphs captured once per *instance*:

  * an instance = static input block + forward graph + backward graph(s) + its OWN activation memory (the graphs' private pool).  The
    activations a forward replay leaves behind must survive until that step's backward, and a rollout has up to `max_action_len` steps in
    flight, so the pool holds as many instances per shape as steps were ever alive at once (~40-60 for the iteration's two rollouts: a few
    hundred MB each at H = 768 -- this is what 288 GB of HBM is for); an instance is checked out by a forward call and returned when the
    autograd node that owns it dies (the iteration's graph is freed);
  * shapes are static per instance: B episodes, V = 37 views, map tokens K padded to a bucket, instruction padded to `Lcap` tokens -- padded
    tokens are masked everywhere, so the valid outputs do not change (same contract as host/nav_graph.py);
  * the step's index arrays are written into the instance's pinned staging block and reach its device block in ONE copy; the gathered map /
    viewpoint embeddings are written by the log gather straight into the instance's input buffer; the per-episode instruction K/V cache
    lives in a static `TextSlot`;
  * each instance wraps its replays in a torch.autograd.Function, so the loop's Python between the calls (embedding log, cross entropy,
    MAKD terms, sampling) composes with autograd unchanged; the weight-gradient GEMMs of a step's backward leave in grouped launches INSIDE
    its backward graph;
  * dropout: the seed pair of an instance is redrawn by a launch inside its forward graph (`magic_step_rng`: counter-hashed, device-side
    counter), and its backward graph regenerates the masks from the same words.

Numerics are those of the eager path kernel for kernel (the graphs replay the very launches the eager Functions issue);
tests/test_step_graphs_gpu.py compares losses, trajectories and every parameter gradient of graph-instanced rollouts with the eager ones.
"""
import numpy as np
import torch

from .lib import capture as _capture
from . import lanes
from . import lib as L
from . import ops as O
from .model_nav import _queue_sync, nav_backward_body, nav_forward_body, pano_backward_body, pano_forward_body

import os
_EAGER_BWD = bool(os.environ.get("MAGIC_STEP_GRAPH_EAGER_BWD"))
# a step's weight-gradient launches as their own graph on the weight-gradient stream (ops.dw_stream).  OFF by default: measured neutral on the
# navigator iteration (183 / 171 ms with, 181 ms without: the chip-filling dW launch and the latency-bound chain slow each other down, as on
# the pretraining step in round 3)
DW_SIDE = os.environ.get("MAGIC_STEP_GRAPH_DW_SIDE", "0") != "0"
# the weight gradients of ALL step instances of a backward pass in one launch per <= 96 Linears at the end of the pass (csrc/gemm.hip
# gemm_dw_cat_kernel): the instances keep their dY / X operands, a Linear's ~38 calls per iteration are summed in registers and its fp32 gradient
# is read-modify-written once per iteration instead of once per step (MAGIC-L: 528 MB per step was the per-step launch's whole cost)
DW_CAT = os.environ.get("MAGIC_STEP_GRAPH_DW_CAT", "1") != "0"
FORK = os.environ.get("MAGIC_STEP_GRAPH_FORK", "1") != "0"      # the two cross-modal encoders of a step as parallel branches of its graphs
# a panorama step's backward graph on a stream of its own, beside the lane's chain of navigation-step backwards (needs DW_CAT: no weight-gradient
# launch inside the step graphs)
PANO_SIDE = os.environ.get("MAGIC_PANO_BWD_STREAM", "auto")      # "auto": nav_rollout.NavRollout decides per model (StepGraphs.pano_side); "0" / "1"
K_BUCKET = 16          # map tokens are padded to a multiple of this
V_STATIC = 37          # views per panorama the instances are built for (36, or 37 when two candidates share a discretised view)


def k_bucket(k, step=K_BUCKET):
    return (int(k) + step - 1) // step * step


class _Block:
    """named fixed-shape arrays at fixed offsets of ONE pinned staging buffer -> ONE device buffer (one H2D copy per use)"""

    def __init__(self, spec, dev):
        off, self.layout = 0, {}
        for name, shape, dt in spec:
            off = (off + 15) & ~15
            n = int(np.prod(shape)) * np.dtype(dt).itemsize
            self.layout[name] = (off, n, tuple(shape), np.dtype(dt))
            off += n
        off = max((off + 15) & ~15, 16)
        self.stage = torch.empty(off, dtype=torch.uint8, pin_memory=True)
        self.stage_np = self.stage.numpy()
        self.dbuf = torch.zeros(off, dtype=torch.uint8, device=dev)
        self.d = {}
        for name, (o, n, shape, dt) in self.layout.items():
            self.d[name] = self.dbuf[o:o + n].view(getattr(torch, dt.name)).view(shape)

    def upload(self, arrays):
        st = self.stage_np
        for k, a in arrays.items():
            o, n, shape, dt = self.layout[k]
            if a.__class__ is not np.ndarray:
                a = np.asarray(a)
            if a.dtype == np.bool_:
                a = a.view(np.uint8)
            if a.shape != shape:
                raise ValueError(f"step instance: array {k!r} has shape {a.shape}, the captured shape is {shape}")
            if a.dtype != dt:
                a = a.astype(dt)
            elif not a.flags.c_contiguous:
                a = np.ascontiguousarray(a)
            st[o:o + n] = a.reshape(-1).view(np.uint8)
        self.dbuf.copy_(self.stage, non_blocking=True)


class TextSlot:
    """static home of one rollout's instruction tensors: K/V cache of the 2 x num_x_layers cross-attention layers, key mask, and the
    (unused on the cached path, shape only) embeddings.  `version` counts its refills: an instance checks at backward time that the slot
    still holds the episode its forward read."""

    def __init__(self, model, B, Lcap):
        net, dev = model.net, model.device_
        nl = 2 * net.cfg.num_x_layers
        self.B, self.L = B, Lcap
        self.kv = torch.zeros(nl, B * Lcap, 2 * net.H, dtype=net.dtype, device=dev)
        self.masks = torch.zeros(B, Lcap, dtype=torch.uint8, device=dev)
        self.txt = torch.zeros(B, Lcap, net.H, dtype=net.dtype, device=dev)
        # the gradient of the K/V cache, collected IN PLACE by every step's backward graph (its attention backwards add their dK / dV here);
        # handed to the cache projection's backward once, by `_KVJoin`, when the last step's backward has run
        self.dkv = torch.zeros_like(self.kv)
        self.version = 0


class _KVJoin(torch.autograd.Function):
    """`token = _KVJoin.apply(txt_kv, slot)`: every captured navigation step takes the one-element token as an input (no data: the graphs read the
    slot's cache at its fixed address) and its backward adds the step's dK / dV into `slot.dkv`.  Autograd runs this node's backward only after
    every consumer of the token has run, i.e. when the accumulator is complete: it hands the accumulator to the cache projection's backward.
    (Before: every step returned its own [2 nl, B L, 2H] gradient -- 150 MB at RxR lengths -- and autograd summed them pairwise.)"""

    @staticmethod
    def forward(ctx, txt_kv, slot):
        slot.dkv.zero_()
        ctx.slot, ctx.version = slot, slot.version
        return torch.zeros(1, dtype=torch.float32, device=txt_kv.device)

    @staticmethod
    def backward(ctx, d_token):
        if ctx.slot.version != ctx.version:
            raise RuntimeError("step instance: the instruction slot was refilled before this rollout's backward ran")
        return ctx.slot.dkv, None


class _Releaser:
    """dies with the autograd node that holds it: returns the instance to its pool"""
    __slots__ = ("inst",)

    def __init__(self, inst):
        self.inst = inst

    def __del__(self):
        inst = self.inst
        if inst is not None:
            inst.busy = False


class _Inst:
    def __init__(self, owner, kind, key):
        self.owner, self.kind, self.key = owner, kind, key
        self.lane = lanes.cur      # the gradient lane (host/lanes.py) whose buffers this instance's backward graphs write
        self.busy = False
        self.block = None
        self.g_fwd = None
        self.bwd = {}              # signature -> (static gradient inputs, graph, outputs)
        self.c = self.plan = self.out = None
        self.seed = torch.zeros(2, dtype=torch.int32, device=owner.dev)
        self.slot = self.slot_version = None
        self.gathered = None
        # the graphs of ONE instance share a private memory pool; instances never share one: memory a graph used for temporaries goes back to
        # its pool after the capture, and a LATER capture into the same pool may place tensors that must stay alive there -- the earlier
        # graph's next replay would then overwrite another instance's saved activations (graphs of one pool are only safe when they are
        # replayed in capture order, which steps of interleaved rollouts and their backwards are not)
        self.pool = torch.cuda.graph_pool_handle()


class _CatEntry:
    """the weight-gradient problems one backward graph leaves for the pass-end launch: operand tensors (kept alive: they are allocations of the
    graph's private pool, never reused while this holds them), their addresses and row counts as arrays, and the interned problem keys"""
    __slots__ = ("keep", "dy", "x", "m", "keys", "dtype")

    def __init__(self, queue, intern):
        self.keep = list(queue)
        self.dy = np.array([e[0].data_ptr() for e in queue], np.int64)
        self.x = np.array([e[1].data_ptr() for e in queue], np.int64)
        self.m = np.array([e[4] for e in queue], np.int32)
        self.dtype = queue[0][0].dtype
        keys = tuple((e[2].data_ptr(), e[3].data_ptr() if e[3] is not None else 0, int(e[5]), int(e[6]), int(e[7]), int(e[8]), int(e[9])) for e in queue)
        self.keys = intern.setdefault((self.dtype, keys), (self.dtype, keys))      # one object per distinct problem list: grouping is by identity


class _PanoInstFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, inst):
        inst.g_fwd.replay()
        ctx.inst, ctx.rel = inst, _Releaser(inst)
        ctx.set_materialize_grads(False)
        outs = tuple(t.detach() for t in inst.out)        # fresh tensor objects over the instance's memory (autograd writes its edges onto them)
        ctx.mark_non_differentiable(outs[1])
        return outs

    @staticmethod
    def backward(ctx, d_emb, _dm, d_fused, d_attn):
        inst = ctx.inst
        model = inst.owner.model
        model.net.S.ensure_grads()
        _queue_sync(model)
        owner = inst.owner
        if owner.pano_side and DW_CAT:
            # nothing later in this pass reads what a panorama step's backward writes (its parameter gradients belong to img_embeddings.* alone and
            # are read when the pass ends; its inputs are features), and the next node on this lane -- the PREVIOUS step's navigation backward --
            # touches other rows of the embedding log: the replay goes to the lane's panorama stream and the lane's chain carries on beside it
            s2 = owner.pano_bwd_stream(inst.lane)
            s2.wait_stream(torch.cuda.current_stream())
            for t in (d_emb, d_fused, d_attn):
                if t is not None:
                    t.record_stream(s2)
            with lanes.use(inst.lane, s2):
                owner._run_bwd(inst, ("d_emb", "d_fused", "d_attn"), (d_emb, d_fused, d_attn))
        else:
            with lanes.use(inst.lane):
                owner._run_bwd(inst, ("d_emb", "d_fused", "d_attn"), (d_emb, d_fused, d_attn))
        return None, None


class _NavInstFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, inst, gathered, token):
        inst.g_fwd.replay()
        ctx.inst, ctx.rel = inst, _Releaser(inst)
        ctx.set_materialize_grads(False)
        return tuple(t.detach() for t in inst.out)

    @staticmethod
    def backward(ctx, d_g, d_v, d_ga, d_va, d_cls, dgl, dll, dfl):
        inst = ctx.inst
        model = inst.owner.model
        if inst.slot.version != inst.slot_version:
            raise RuntimeError("step instance: the instruction slot was refilled (a newer rollout started) before this rollout's backward ran; "
                               "back-propagate a rollout before starting the next one on the same slot")
        model.net.S.ensure_grads()
        _queue_sync(model)
        if d_cls is None:                     # (the last step's [cls] feeds nothing: one signature for every step)
            d_cls = inst.owner._zeros_cls(inst)
        with lanes.use(inst.lane):
            bo = inst.owner._run_bwd(inst, ("d_g", "d_v", "d_ga", "d_va", "d_cls", "dgl", "dll", "dfl"), (d_g, d_v, d_ga, d_va, d_cls, dgl, dll, dfl))
        return None, None, bo["d_gathered"], None


class StepGraphs:
    """per-(model, feature table) pools of captured panorama / navigation step instances"""

    def __init__(self, model, table, B, Lcap, max_instances=160, base_seed=0x5EED5EED):
        self.model, self.table, self.B, self.Lcap = model, table, int(B), int(Lcap)
        self.dev = model.device_
        self.max_instances = max_instances
        self.pools = {}            # key -> [instances]
        self.seen = set()          # keys that ran eagerly once (lazy host-side initialisation happens there, never inside a capture)
        self.slots = {}
        self.n_inst = 0
        self.captures = 0
        self.stream = torch.cuda.Stream(device=self.dev)
        self.side = torch.cuda.Stream(device=self.dev) if FORK else None      # second branch of a step graph (global || local cross-modal encoder)
        self.base_seed = int(base_seed)
        self.rng_counter = torch.zeros(1, dtype=torch.int32, device=self.dev)      # lane 0's; other lanes: _rng()
        self._rngs = {0: self.rng_counter}
        self._zc = {}
        self._cat_used, self._cat_intern, self._cat_plan, self._cat_stage = [], {}, {}, None
        self._pbs = {}             # lane -> the stream its panorama steps' backward graphs replay on (PANO_SIDE)
        self.pano_side = PANO_SIDE == "1"
        sets = getattr(model, "_step_graph_sets", None)
        if sets is None:
            sets = model._step_graph_sets = []
        import weakref
        sets.append(weakref.ref(self))

    def pano_bwd_stream(self, lane):
        s = self._pbs.get(lane)
        if s is None:
            s = self._pbs[lane] = torch.cuda.Stream(device=self.dev)
        return s

    # ---- bookkeeping -------------------------------------------------------------------------------------------
    def _mode_key(self):
        m = self.model
        return (bool(m.training), float(m.dropout.p), float(m.attention_dropout.p))

    def usable(self):
        m = self.model
        return (m.store.requires_grad and torch.is_grad_enabled() and not m.causal_blocks and self.dev.type == "cuda"
                and not O.FLOPS["enabled"] and not L.PROFILE["on"])

    def text_slot(self, idx):
        s = self.slots.get(idx)
        if s is None:
            s = self.slots[idx] = TextSlot(self.model, self.B, self.Lcap)
        return s

    def _acquire(self, kind, key, build):
        """a free instance of `key`, a newly captured one, or None (first sight of the key / pool exhausted: the caller runs eagerly)"""
        key = (kind,) + key + (lanes.cur,) + self._mode_key()
        lst = self.pools.setdefault(key, [])
        for inst in lst:
            if not inst.busy:
                inst.busy = True
                return inst
        if key not in self.seen:
            self.seen.add(key)
            return None
        if self.n_inst >= self.max_instances:
            return None
        self._lane_ready()
        inst = _Inst(self, kind, key)
        build(inst)
        lst.append(inst)
        self.n_inst += 1
        inst.busy = True
        return inst

    def _lane_ready(self):
        """what a lane's graphs bake in must exist before the first capture for it: its dropout counter, its gradient buffer, the workspace and
        counters of its weight-gradient launches"""
        k = lanes.cur
        if k not in self._rngs:
            self._rngs[k] = torch.zeros(1, dtype=torch.int32, device=self.dev)
            self.model.store.ensure_lanes(k + 1)
            O.dw_counters(self.dev)

    def _capture(self, inst, body):
        import gc
        g = torch.cuda.CUDAGraph()
        # no automatic garbage collection while the stream is capturing: a collection can run finalizers that talk to the runtime (a dead
        # rollout's graphs, pinned staging buffers) in the middle of the capture -- seen as an abort in round 5 (torch collects once itself,
        # BEFORE the capture begins)
        was = gc.isenabled()
        gc.disable()
        try:
            with _capture(g, pool=inst.pool, stream=self.stream, capture_error_mode="relaxed"):
                body()
        finally:
            if was:
                gc.enable()
        self.captures += 1
        return g

    def _arm(self, inst):
        """inside a forward capture: redraw this instance's seed pair on the device, arm the model's dropout with it"""
        m = self.model
        O.step_rng((self.base_seed + 0x9E3779B1 * inst.lane) & 0x7FFFFFFF, self._rngs[inst.lane], 1.0, seed_out=inst.seed)
        m.dropout_seed = inst.seed
        try:
            m._arm_dropout()
        finally:
            m.dropout_seed = None

    # ---- panorama ------------------------------------------------------------------------------------------------
    def pano_inst(self, V):
        if not self.usable() or V != V_STATIC:
            return None
        return self._acquire("pano", (self.B, V), self._build_pano)

    def _build_pano(self, inst):
        B, V = self.B, V_STATIC
        f32, i32, u8 = np.float32, np.int32, np.uint8
        inst.block = _Block([("vp_rows", (B,), i32), ("view_order", (B, V), i32), ("loc_fts", (B, V, 7), f32), ("nav_types", (B, V), i32),
                             ("view_lens", (B,), i32), ("pano_masks", (B, V), u8)], self.dev)
        d = inst.block.d

        def body():
            self._arm(inst)
            fts = torch.empty(B, V, self.table.shape[2], dtype=self.table.dtype, device=self.dev)
            O.view_gather(self.table, d["vp_rows"], d["view_order"], fts)
            inst.c, inst.plan, inst.out = pano_forward_body(self.model, fts, d["loc_fts"], d["nav_types"], d["view_lens"], d["pano_masks"].view(torch.bool))
        self.model.store.sync_shadow()
        inst.g_fwd = self._capture(inst, body)

    def run_pano(self, inst, arrays):
        inst.block.upload(arrays)
        return _PanoInstFn.apply(self.model._anchor, inst)

    # ---- navigation ----------------------------------------------------------------------------------------------
    def nav_inst(self, K, Vp, slot_idx):
        if not self.usable() or Vp != V_STATIC + 2 or K % K_BUCKET:
            return None
        return self._acquire("nav", (self.B, K, Vp, slot_idx), lambda inst: self._build_nav(inst, K, Vp, slot_idx))

    def _build_nav(self, inst, K, Vp, slot_idx):
        B, L, m = self.B, self.Lcap, self.model
        H = m.net.H
        f32, i32, u8 = np.float32, np.int32, np.uint8
        inst.block = _Block([("gmap_step_ids", (B, K), i32), ("gmap_pos_fts", (B, K, 7), f32), ("gmap_pair_dists", (B, K, K), f32),
                             ("gmap_masks", (B, K), u8), ("gmap_logit_masks", (B, K), u8), ("vp_pos_fts", (B, Vp, 14), f32),
                             ("vp_nav_masks", (B, Vp), u8), ("vp_masks", (B, Vp), u8), ("fsrc", (B, K), i32), ("bw", (B, Vp), u8)], self.dev)
        inst.gathered = torch.zeros(B * K + B * Vp, H, dtype=m.net.dtype, device=self.dev)
        inst.slot = slot = self.text_slot(slot_idx)
        d = inst.block.d
        m._first_rows(B, K, self.dev)
        m._first_rows(B, Vp, self.dev)
        lens = ([L] * B, [K] * B, [Vp] * B)                  # (FLOP accounting only: not counted under graph replay)

        def body():
            self._arm(inst)
            b = dict(txt_masks=slot.masks, gmap_masks=d["gmap_masks"], vp_masks=d["vp_masks"], gmap_step_ids=d["gmap_step_ids"],
                     gmap_pos_fts=d["gmap_pos_fts"], gmap_pair_dists=d["gmap_pair_dists"], gmap_logit_masks=d["gmap_logit_masks"],
                     vp_pos_fts=d["vp_pos_fts"], vp_nav_masks=d["vp_nav_masks"], host_lens=lens, fusion=(d["fsrc"], d["bw"]), fork=self.side)
            inst.c, inst.out = nav_forward_body(m, inst.gathered[:B * K].view(B, K, H), inst.gathered[B * K:].view(B, Vp, H), slot.txt, b, slot.kv)
        m.store.sync_shadow()
        inst.g_fwd = self._capture(inst, body)

    def kv_token(self, slot_idx, txt_kv):
        """once per rollout, after `txt_kv = model.text_kv(txt_embeds, out=slot.kv)`: the token every step of the rollout passes to run_nav"""
        return _KVJoin.apply(txt_kv, self.text_slot(slot_idx))

    def run_nav(self, inst, arrays, gathered, token):
        """gathered: the log gather's output, written into inst.gathered (carries the autograd history of the embeddings); token: kv_token()"""
        if gathered.data_ptr() != inst.gathered.data_ptr():
            inst.gathered.copy_(gathered.detach())
        inst.block.upload(arrays)
        inst.slot_version = inst.slot.version
        g, v, ga, va, cls, gl, ll, fl = _NavInstFn.apply(self.model._anchor, inst, gathered, token)
        return dict(gmap_embeds=g, vp_embeds=v, gmap_attns=ga, vp_attns=va, cls_embeds=cls, global_logits=gl, local_logits=ll, fused_logits=fl)

    def _zeros_cls(self, inst):
        z = self._zc.get(inst.key)
        if z is None:
            z = self._zc[inst.key] = torch.zeros(self.B, self.model.net.H, dtype=self.model.net.dtype, device=self.dev)
        return z

    # ---- backward --------------------------------------------------------------------------------------------------
    def _run_bwd(self, inst, names, grads):
        if _EAGER_BWD:                # debugging aid: the instance's forward replayed from its graph, its backward launched eagerly
            m = self.model
            if inst.kind == "pano":
                pano_backward_body(m, inst.c, inst.plan, *grads)
                return {}
            d_gin, d_vin, _, _ = nav_backward_body(m, inst.c, *grads, dkv_acc=inst.slot.dkv)
            return dict(d_gathered=torch.cat([d_gin, d_vin], 0))
        sig = tuple(g is not None for g in grads)
        ent = inst.bwd.get(sig)
        if ent is None:
            ent = inst.bwd[sig] = self._capture_bwd(inst, names, grads)
        bi, g, bo, g_dw, cat = ent
        if getattr(inst, "lane", 0):  # the replayed graph writes the lane's gradient buffer through pointers baked in at capture: no handle access marks it
            self.model.store.lane_dirty = True
        for n, t in zip(names, grads):
            if t is not None:
                dst = bi[n]
                dst.copy_(t.reshape(dst.shape))
        if g_dw is None and not DW_CAT:
            O.dw_guard(self.dev)      # (the backward graph holds the weight-gradient launch: shared workspace + counters, ops.dw_guard)
        g.replay()
        if g_dw is not None:          # this step's weight gradients: on the device's weight-gradient stream, under the next step's backward chain
            ds = O.dw_stream(self.dev)
            ds.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ds):
                O.dw_guard(self.dev)
                g_dw.replay()
            O.dw_stream_used(self.dev)
        if cat is not None:
            self._cat_used.append(cat)
        return bo

    def _capture_bwd(self, inst, names, grads):
        m = self.model
        bi = {n: torch.zeros(t.shape, dtype=t.dtype, device=self.dev) for n, t in zip(names, grads) if t is not None}
        bo = {}
        # the weight-gradient GEMMs of this step leave in grouped launches inside its backward graph: isolate the launch layer's queues from
        # whatever the surrounding autograd pass has queued eagerly
        saved = (O.DEFER["queue"], O.DEFER["active"], O.DEFER.get("bytes", 0), list(O.RBW_JOBS), list(O.PART_JOBS))
        O.DEFER["queue"], O.DEFER["bytes"] = [], 0
        O.RBW_JOBS[:] = []
        O.PART_JOBS[:] = []

        def body():
            O.defer_dw(True)
            if inst.kind == "pano":
                pano_backward_body(m, inst.c, inst.plan, bi.get("d_emb"), bi.get("d_fused"), bi.get("d_attn"))
            else:
                d_gin, d_vin, _, _ = nav_backward_body(m, inst.c, *[bi.get(n) for n in names], dkv_acc=inst.slot.dkv, fork=self.side)
                bo["d_gathered"] = torch.cat([d_gin, d_vin], 0)
            if DW_CAT:
                O.flush_rbw_parts()       # the column sums of partial parameter-gradient rows stay inside the graph; the dW queue is left for flush_cat
            elif not DW_SIDE:
                O.flush_dw()
        g_dw = cat = None
        try:
            g = self._capture(inst, body)
            if DW_CAT and O.DEFER["queue"]:
                cat = _CatEntry(O.DEFER["queue"], self._cat_intern)
            elif DW_SIDE and (O.DEFER["queue"] or O.PART_JOBS or O.RBW_JOBS):
                # the step's weight-gradient launches (grouped dW GEMMs over the operands the backward graph leaves in the instance's memory +
                # the column sums of its partial parameter-gradient rows) as a graph of their own, replayed on the weight-gradient stream
                g_dw = self._capture(inst, O.flush_dw)
        finally:
            O.DEFER["queue"], O.DEFER["active"], O.DEFER["bytes"] = saved[0], saved[1], saved[2]
            O.RBW_JOBS[:] = saved[3]
            O.PART_JOBS[:] = saved[4]
        return bi, g, bo, g_dw, cat

    def flush_cat(self):
        """end of a backward pass (model_nav._queue_sync's callback, after the lanes were joined): every Linear's weight gradient over all the
        step instances that ran in this pass, <= 96 Linears per launch.  Segment order = the order the instances' backwards ran in."""
        used, self._cat_used = self._cat_used, []
        if not used:
            return
        if self.model.store.lane_grads:       # (the concatenated launch writes dW of whichever lane its entries were captured on)
            self.model.store.lane_dirty = True
        groups = {}
        for c in used:
            groups.setdefault(id(c.keys), []).append(c)
        launches, total = [], 0
        for ents in groups.values():
            dtype, keys = ents[0].keys
            plan = self._cat_plan.get(id(ents[0].keys))
            if plan is None:              # unique dW pointers of this problem list, and which queue entries feed each (a Linear used twice in a step)
                uniq, cols = {}, []
                for j, k in enumerate(keys):
                    if k[0] in uniq:
                        if keys[uniq[k[0]]][1:] != k[1:]:
                            raise RuntimeError("step instances: one dW queued with two different shapes")
                        cols[uniq[k[0]]].append(j)
                    else:
                        uniq[k[0]] = len(cols)
                        cols.append([j])
                plan = self._cat_plan[id(ents[0].keys)] = ([keys[c[0]] for c in cols], cols, max(len(c) for c in cols))
            probs, cols, width = plan
            E = len(ents)
            DY, X, M = np.stack([c.dy for c in ents]), np.stack([c.x for c in ents]), np.stack([c.m for c in ents])      # [E, n_entries]
            n_seg = E * width
            if width == 1:
                sel = [c[0] for c in cols]
                dy_t, x_t, m_t = DY[:, sel].T, X[:, sel].T, M[:, sel].T
            else:
                dy_t, x_t, m_t = np.zeros((len(cols), n_seg), np.int64), np.zeros((len(cols), n_seg), np.int64), np.zeros((len(cols), n_seg), np.int32)
                for j, c in enumerate(cols):
                    k = E * len(c)
                    dy_t[j, :k], x_t[j, :k], m_t[j, :k] = DY[:, c].reshape(-1), X[:, c].reshape(-1), M[:, c].reshape(-1)
                dy_t[m_t == 0] = dy_t[0, 0]               # (skipped segments: any valid address)
                x_t[m_t == 0] = x_t[0, 0]
            # wide problems (both dimensions >= 128) go in launches of their own: the kernel then works on 128 x 128 tiles
            wide = [j for j, k in enumerate(probs) if k[2] >= 128 and k[3] >= 128]
            narrow = [j for j, k in enumerate(probs) if not (k[2] >= 128 and k[3] >= 128)]
            for idx in (wide, narrow):
                for a in range(0, len(idx), 96):
                    sel = idx[a:a + 96]
                    launches.append((dtype, [probs[j] for j in sel], n_seg, np.ascontiguousarray(dy_t[sel]), np.ascontiguousarray(x_t[sel]),
                                     np.ascontiguousarray(m_t[sel]), total))
                    total += len(sel) * n_seg * 20
                    total = (total + 15) & ~15
        st = self._cat_stage
        if st is not None:
            st[2].synchronize()           # the previous pass's table copy has left the pinned buffer
        if st is None or st[0].numel() < total:
            cap = max(total * 2, 1 << 20)
            st = self._cat_stage = (torch.empty(cap, dtype=torch.uint8, pin_memory=True), torch.empty(cap, dtype=torch.uint8, device=self.dev), torch.cuda.Event())
        host = st[0].numpy()
        for dtype, probs, n_seg, dy_t, x_t, m_t, off in launches:
            n = len(probs) * n_seg
            host[off:off + 8 * n] = dy_t.reshape(-1).view(np.uint8)
            host[off + 8 * n:off + 16 * n] = x_t.reshape(-1).view(np.uint8)
            host[off + 16 * n:off + 20 * n] = m_t.reshape(-1).view(np.uint8)
        st[1][:total].copy_(st[0][:total], non_blocking=True)
        st[2].record()
        base = st[1].data_ptr()
        for dtype, probs, n_seg, dy_t, x_t, m_t, off in launches:
            n = len(probs) * n_seg
            O.dw_cat(dtype, [(k[0], k[1], k[2], k[3], k[4], k[5], k[6]) for k in probs], n_seg, base + off, base + off + 8 * n, base + off + 16 * n)

    def report(self):
        return {"instances": self.n_inst, "captures": self.captures,
                "by_key": {"/".join(str(x) for x in k[:6]): len(v) for k, v in self.pools.items() if v}}
