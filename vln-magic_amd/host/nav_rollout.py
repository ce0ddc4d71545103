"""The navigator step loop on index plans (BASELINE config 5).

Same computation as `GMapNavAgent.rollout` (map_nav_src/r2r/agent.py:722-1160) for feedback 'teacher' | 'argmax' |
'sample', with or without MAKD against a frozen teacher -- but the per-sample Python over device tensors is gone:
host/nav_plan.NavPlanner turns each step's observations into index arrays (ONE packed host->device copy per step), the
panorama features are gathered on the device from the HBM-resident table (`magic_view_gather`), every step's embeddings
are appended to a device log, and `gmap_img_embeds` / `vp_img_embeds` are one CSR gather over that log whose backward is
the transposed gather (so `loss.backward()` still reaches every earlier step's panorama encoder, as `pad_tensors_wgrad`
does in the reference, :234).

Device->host traffic: none under teacher forcing (the stop probabilities the reference reads with `.item()` per sample
and step, :986-996, are kept on the device and read once after the loop); one [B] action vector per step for 'argmax' /
'sample' (the stepper needs it).
"""
import contextlib
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import lanes
from . import ops as O
from .makd_nav import compute_kd_losses, compute_kd_losses_fused, kd_terms
from .nav_plan import IGNORE, NavPlanner
from .kd_loss import ce_rows_loss, exponential_decay


_SLOT = 1 << 16
_RING = {"slots": [], "turn": 0, "used": []}


def to_device(arrays, dev):
    """ONE pinned staging buffer + ONE async copy for a dict of numpy arrays; returns device tensors (views of the copy)"""
    metas, off = [], 0
    for k, a in arrays.items():
        a = np.ascontiguousarray(a)
        if a.dtype == np.bool_:
            a = a.view(np.uint8)
        off = (off + 15) & ~15
        metas.append((k, a, off))
        off += a.nbytes
    ev = None
    if dev.type == "cuda" and off <= _SLOT:
        # a slot of ONE pinned ring instead of a pinned allocation per call (the step loop stages a handful of small arrays per step: the host
        # allocator's bookkeeping was most of this function); a slot is reused 128 calls later, behind the event of its last copy
        if not _RING["slots"]:
            big = torch.empty(128 * _SLOT, dtype=torch.uint8, pin_memory=True)
            _RING["slots"] = [(big[i * _SLOT:(i + 1) * _SLOT], big[i * _SLOT:(i + 1) * _SLOT].numpy(), torch.cuda.Event()) for i in range(128)]
            _RING["used"] = [False] * 128
        i = _RING["turn"] = (_RING["turn"] + 1) % 128
        slot, sn, ev = _RING["slots"][i]
        if _RING["used"][i]:
            ev.synchronize()
        _RING["used"][i] = True
        stage = slot[:max(off, 16)]
    else:
        stage = torch.empty(max(off, 16), dtype=torch.uint8, pin_memory=(dev.type == "cuda"))
        sn = stage.numpy()
    for k, a, o in metas:
        sn[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    buf = stage.to(dev, non_blocking=True)
    if ev is not None:
        ev.record()
    out = {}
    for k, a, o in metas:
        t = buf[o:o + a.nbytes].view(getattr(torch, str(a.dtype))).view(a.shape)
        out[k] = t
    out["_stage"] = stage           # keep the pinned buffer alive until the copy has run
    return out


def _extent(x, *ext):
    """x[:, :ext0, :ext1, ...] (None = the whole dimension): the valid extent of a tensor computed on padded static shapes"""
    idx = (slice(None),) + tuple(slice(None) if e is None else slice(0, e) for e in ext)
    return x[idx]


def _nav_extent(outs, K, Vp, L):
    return dict(outs, gmap_embeds=outs["gmap_embeds"][:, :K], vp_embeds=outs["vp_embeds"][:, :Vp], gmap_attns=outs["gmap_attns"][:, :, :K, :L],
                vp_attns=outs["vp_attns"][:, :, :Vp, :L])


class _GradRows:
    """holder of an EmbeddingLog's gradient rows: what the autograd nodes keep (never the log itself -- the log holds the tokens, a token's node
    holding the log would be a reference cycle, and the rollout's autograd graph, with the step instances it owns, would wait for the
    garbage collector instead of dying with the iteration)"""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


class _LogSrc(torch.autograd.Function):
    """`tok = _LogSrc.apply(x, log, row0, n)`: x was copied into log rows [row0, row0 + n).  Every later gather of the rollout takes the
    one-element token as an input (no data) and ADDS its transposed gather into `log.rows.buf` in place; autograd runs this node's backward
    only after every gather that holds the token has run -- the rows are complete then -- and hands them to x's producer ONCE.
    (Before: every gather returned a gradient for every earlier step tensor and autograd summed them pairwise: ~T^2 / 2 bf16 add
    launches per rollout and tensor kind -- 1 226 per iteration at RxR lengths, 4.6 ms of kernel time -- and every gather's backward
    wrote a full [n_src, H] gradient log, zeros included.)"""

    @staticmethod
    def forward(ctx, x, rows, row0, n):
        ctx.rows, ctx.span, ctx.shape = rows, (row0, row0 + n), x.shape
        return torch.empty(1, dtype=torch.float32, device=x.device)

    @staticmethod
    def backward(ctx, _d):
        if getattr(ctx, "done", False):
            raise RuntimeError("a second backward pass through the same rollout's embedding log: its gradient rows are accumulated in place and "
                               "were handed over by the first pass (run the rollout again, or back-propagate the summed loss once)")
        ctx.done = True
        a, b = ctx.span
        return ctx.rows.buf[a:b].view(ctx.shape), None, None, None


class _LogChain(torch.autograd.Function):
    """`chain_t = _LogChain.apply(chain_{t-1}, *new tokens)`: ONE token per gather instead of one per logged tensor (a gather late in the
    rollout held ~60 inputs: autograd's per-input bookkeeping was a measurable share of a step's host time).  gather_t holds chain_t; chain_t
    holds chain_{t-1} and the tokens of the tensors logged since: a tensor's `_LogSrc` node still runs only after every later gather has."""

    @staticmethod
    def forward(ctx, *toks):
        ctx.n = len(toks)
        ctx.set_materialize_grads(False)
        return torch.empty(1, dtype=torch.float32, device=toks[0].device)

    @staticmethod
    def backward(ctx, _d):
        return (None,) * ctx.n


class _LogGather(torch.autograd.Function):
    """out[n_out, H] = CSR(ptr, idx, w) x log; backward = the transposed CSR added into the log's gradient rows (`log.rows.buf`); the step tensors
    the rows were copied from receive them through their `_LogSrc` tokens."""

    @staticmethod
    def forward(ctx, buf, rows, csr, csr_t, n_out, n_src, out, *toks):
        H = buf.shape[1]
        # out: a static buffer to gather into (the input of a captured step instance, host/step_graphs.py) or None
        out = torch.empty(n_out, H, dtype=buf.dtype, device=buf.device) if out is None else out.detach()
        O.csr_gather(buf, csr[0], csr[1], csr[2], out, n_out, H)
        ctx.rows, ctx.csr_t, ctx.n_src, ctx.H = rows, csr_t, n_src, H
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, d_out):
        if d_out is not None:
            O.csr_gather(d_out.contiguous(), ctx.csr_t[0], ctx.csr_t[1], ctx.csr_t[2], ctx.rows.buf, ctx.n_src, ctx.H, accumulate=True)
        return (None,) * len(ctx.needs_input_grad)


# host-side section timers of `steps` (MAGIC_NAV_TIMERS=1; profiles/micro/nav_kernel_breakdown.py prints them): where a step's host time goes,
# without a profiler's per-call overhead.  _tk(name) charges the time since the previous mark to `name`.
_T = {"on": bool(os.environ.get("MAGIC_NAV_TIMERS")), "acc": {}, "last": 0.0}


def _tk(name):
    if _T["on"]:
        import time
        now = time.perf_counter()
        _T["acc"][name] = _T["acc"].get(name, 0.0) + now - _T["last"]
        _T["last"] = now


FUSED_MAKD = os.environ.get("MAGIC_NAV_FUSED_MAKD", "1") != "0"      # a step's mse distillation terms in one launch (makd_nav.compute_kd_losses_fused)
PANO_SIDE = os.environ.get("MAGIC_PANO_BWD_STREAM", "auto")      # step_graphs.PANO_SIDE: "auto" = on for the wide models (H >= 384)
LANES = os.environ.get("MAGIC_NAV_LANES", "1") != "0"      # the rollouts of `run_interleaved` as gradient lanes on streams of their own


class EmbeddingLog:
    """append-only [rows, H] device buffer of one model's per-step outputs (`buf`), its gradient rows (`rows.buf`: the gathers' backwards add into
    them in place) and the autograd tokens of the tracked rows"""

    def __init__(self, H, dtype, dev, rows=4096):
        self.buf = torch.zeros(rows, H, dtype=dtype, device=dev)
        self.rows = _GradRows()                # .buf allocated (zero) with the first tracked rows
        self.toks = []                         # tokens of the tensors logged since the last gather
        self.chain = None                      # the token the next gather holds (_LogChain)

    def _grow(self, rows):
        nb = torch.zeros(max(2 * self.buf.shape[0], rows), self.buf.shape[1], dtype=self.buf.dtype, device=self.buf.device)
        nb[:self.buf.shape[0]] = self.buf
        self.buf = nb
        if self.rows.buf is not None:          # (forward only: no gradient has been written yet)
            self.rows.buf = torch.zeros_like(nb)

    def put(self, row0, x, track=True):
        x2 = x.detach().reshape(-1, self.buf.shape[1])
        n = x2.shape[0]
        if row0 + n > self.buf.shape[0]:
            self._grow(row0 + n)
        self.buf[row0:row0 + n].copy_(x2)
        if track and x.requires_grad:
            if self.rows.buf is None:
                self.rows.buf = torch.zeros_like(self.buf)
            self.toks.append(_LogSrc.apply(x, self.rows, row0, n))

    def gather(self, csr, csr_t, n_out, n_src, grad=True, out=None):
        if grad and (self.toks or self.chain is not None):
            if self.toks:
                self.chain = _LogChain.apply(*(([self.chain] if self.chain is not None else []) + self.toks))
                self.toks = []
            return _LogGather.apply(self.buf, self.rows, csr, csr_t, n_out, n_src, out, self.chain)
        if out is None:
            out = torch.empty(n_out, self.buf.shape[1], dtype=self.buf.dtype, device=self.buf.device)
        O.csr_gather(self.buf, csr[0], csr[1], csr[2], out, n_out, self.buf.shape[1])
        return out


class PlanAhead:
    """The index plans of a TEACHER-FORCED rollout computed ahead of time on a helper thread (`NavRollout.plan_ahead`).  Under teacher forcing
    nothing the planner does depends on the model -- the actions are the expert's -- so the whole sequence of step plans of the NEXT batch can be
    built while the GPU runs the current iteration's backward (the main thread is parked inside `loss.backward()` then, the GIL free), the way the
    reference's PrefetchLoader prepares the next batch (pretrain_src/data/loader.py:78-120).  `steps(..., ahead=...)` consumes the plans."""

    def __init__(self, make_planner, thread=True):
        """thread=False: the plans are built right here, on the calling thread -- placed after the iteration's backward and optimizer launches
        (nothing in between waits for the GPU), the host plans the next batch while the GPU drains them; no second Python thread, no GIL fight"""
        import threading
        self.planner = self.plans = self.err = None
        self._make = make_planner
        self._th = None
        if thread:
            self._th = threading.Thread(target=self._run, name="magic-plan-ahead", daemon=True)
            self._th.start()
        else:
            self._run()

    def _run(self):
        try:
            pl = self._make()
            plans = []
            while True:
                plan = pl.begin_pano()
                live = int((~pl.ended).sum())
                plan.update(pl.begin_nav())
                done = pl.end_step(None)
                plan.update(_live=live, _done=done, _actions=list(pl.actions))
                plans.append(plan)
                if done or len(plans) >= pl.T:
                    break
            self.planner, self.plans = pl, plans
        except BaseException as e:          # noqa: BLE001 - re-raised by result() on the caller's thread
            self.err = e

    def result(self):
        if self._th is not None:
            self._th.join()
        if self.err is not None:
            raise self.err
        return self.planner, self.plans


class NavRollout:
    def __init__(self, student, feature_table, teacher=None, kd=None, max_action_len=15, expert_policy="spl", cache_text_kv=True,
                 train_teacher=False, graphs=False, Lcap=None):
        """feature_table: [n_viewpoints, 36, D] device tensor in the student's compute dtype (packed once, SURVEY f-2).
        kd: dict(alpha, temperature, decay[, ability_weight]) -- MAKD hyper-parameters (run_r2r_kdl_valid.sh:97-104); ability_weight =
        args.kdl_adaptive_ability_weight_type: 'RW' (default: the `rw_seq` scalars a caller passes), 'learned_weight' (softplus of the
        learner model's kdl_*_weight parameters, agent.py:583-586) or None.
        graphs: run the panorama / navigation segments of every TRAINING step as captured HIP graphs on per-step instances
        (host/step_graphs.py); Lcap = the instruction length the static shapes are built for (args.max_instr_len; longer batches run eagerly)."""
        self.student, self.teacher, self.kd = student, teacher, kd
        self.table = feature_table
        self.T, self.expert = max_action_len, expert_policy
        self.cache_text_kv = cache_text_kv
        self.train_teacher = bool(train_teacher) and teacher is not None      # ICoD co-training (args.train_kdl_teacher)
        self.dev = feature_table.device
        self.graphs, self.Lcap, self._sg = bool(graphs) and Lcap is not None and cache_text_kv, Lcap, {}
        if teacher is not None:
            self.heads = {n: getattr(student.vln_bert, n) for n in ("txt_emb_w", "kdl_img_w", "kdl_avg_img_w", "global_cross_w", "local_cross_w")}

    def step_graphs(self, model, B):
        """the model's pools of captured step instances for batches of B episodes (created on first use)"""
        from .step_graphs import StepGraphs
        sg = self._sg.get(id(model))
        if sg is None or sg.B != B:
            sg = self._sg[id(model)] = StepGraphs(model, self.table, B, self.Lcap)
            # panorama backwards beside the lane's chain: a WIDE model gains (MAGIC-L navigator iteration 134.9 -> 124.6 ms; ICoD with only its MAGIC-L
            # teacher's panoramas there 120.1 -> 116.7), the MAGIC-S student's few-microsecond kernels lose (ICoD 120 -> 124-145 with the student's, 114.8 ->
            # 124.9 with both): profiles/micro/r05_ab_pano_side.txt
            sg.pano_side = (PANO_SIDE == "1" or (PANO_SIDE == "auto" and model.net.H >= 384)
                            or (PANO_SIDE == "teacher" and model is self.teacher) or (PANO_SIDE == "student" and model is not self.teacher))
        return sg

    def _use_graphs(self, obs, grad, text_copies):
        return bool(self.graphs and grad and max(len(ob["instr_encoding"]) for ob in obs) <= self.Lcap)

    def _planner(self, env, obs, feedback, grad, use_g):
        from .step_graphs import K_BUCKET, V_STATIC
        return NavPlanner(env, obs, feedback=feedback, max_action_len=self.T, expert_policy=self.expert, train=grad,
                          pad_V=V_STATIC if use_g else 0, k_bucket=K_BUCKET if use_g else 1)

    def plan_ahead(self, env, obs, grad=True, text_copies=1, thread=True):
        """plan a teacher-forced rollout over (env, obs) ahead of time -- on a helper thread, or (thread=False) here and now; pass the handle
        as `ahead=` to `steps` / `run` with the SAME env and obs (the planner steps `env` through the episodes: do not touch it until the
        rollout has consumed the plans)"""
        use_g = self._use_graphs(obs, grad, text_copies)
        return PlanAhead(lambda: self._planner(env, obs, "teacher", grad, use_g), thread=thread)

    def graph_report(self):
        return {("teacher" if m is self.teacher else "student"): self._sg[id(m)].report() for m in (self.student, self.teacher)
                if m is not None and id(m) in self._sg}

    def _pano_inputs(self, d, plan):
        B, V = plan["B"], plan["V"]
        fts = torch.empty(B, V, self.table.shape[2], dtype=self.table.dtype, device=self.dev)
        O.view_gather(self.table, d["vp_rows"], d["view_order"], fts)
        return dict(view_img_fts=fts, loc_fts=d["loc_fts"], nav_types=d["nav_types"], view_lens=d["view_lens"], already_dropout=True,
                    pano_masks=d["pano_masks"].view(torch.bool))

    def _nav_inputs(self, d, plan, gathered, txt_embeds, txt_masks, txt_lens, txt_kv=None):
        B, K, Vp, H = plan["B"], plan["K"], plan["Vp"], gathered.shape[1]
        return dict(gmap_img_embeds=gathered[:B * K].view(B, K, H), vp_img_embeds=gathered[B * K:].view(B, Vp, H),
                    txt_embeds=txt_embeds, txt_kv=txt_kv, txt_masks=txt_masks, gmap_masks=d["gmap_masks"].view(torch.bool), vp_masks=d["vp_masks"].view(torch.bool),
                    gmap_step_ids=d["gmap_step_ids"], gmap_pos_fts=d["gmap_pos_fts"], gmap_pair_dists=d["gmap_pair_dists"],
                    gmap_visited_masks=d["gmap_visited_masks"].view(torch.bool), gmap_logit_masks=d["gmap_logit_masks"], gmap_vpids=plan["gmap_vpids"], vp_pos_fts=d["vp_pos_fts"],
                    vp_nav_masks=d["vp_nav_masks"].view(torch.bool), vp_cand_vpids=plan["vp_cand_vpids"],
                    host_lens=(txt_lens, [int(x) - 1 for x in plan["gmap_lens"]], [int(x) + 2 for x in plan["view_lens"]]),
                    fusion=(d["fsrc"], d["bw"]))

    def run(self, *args, **kw):
        """One batch of episodes (see `steps` for the arguments).  Returns dict(loss, ml_loss, kdl, traj, n_steps, decisions[, steps])."""
        g = self.steps(*args, **kw)
        try:
            while True:
                next(g)
        except StopIteration as e:
            return e.value

    def run_interleaved(self, jobs):
        """Several rollouts advanced round-robin, one step each: jobs = [(args, kwargs) of `steps`, ...] -> list of results.
        Every rollout yields right after it has LAUNCHED a step, before it needs that step's actions on the host; the host then plans
        and launches the next rollout's step while the GPU works, so the per-step action copy of a 'sample' rollout and its planning no
        longer leave the GPU idle (the iteration's teacher-forced and DAgger rollouts: agent_base.py:243-250).  Needs one stepper per
        rollout; the rollouts are independent, so the results equal those of running them one after the other."""
        gens = [self.steps(*a, **dict(k, slot=k.get("slot", i))) for i, (a, k) in enumerate(jobs)]
        res = [None] * len(gens)
        live = list(range(len(gens)))
        # gradient lanes (host/lanes.py): with captured step instances every rollout works on its own stream into its own gradient buffer, so
        # the rollouts' latency-bound chains overlap on the GPU in the forward AND in the one backward pass that follows
        laned = self.graphs and LANES and len(gens) > 1 and self.dev.type == "cuda" and all(k.get("grad", True) for _, k in jobs)
        # ... and only when the step instances ARE usable: a model with causal-intervention blocks runs its steps eagerly (StepGraphs.usable() is False),
        # and eager autograd nodes on a lane stream are isolated only if every one of them re-enters its lane in backward (model_nav._lane_bwd)
        if laned and (getattr(self.student, "causal_blocks", None) or (self.train_teacher and getattr(self.teacher, "causal_blocks", None))):
            laned = False
        ctx = [contextlib.nullcontext] * len(gens)
        if laned:
            lanes.fork(self.dev, range(len(gens)))
            ctx = [(lambda i=i: lanes.use(i, lanes.stream(self.dev, i) if i else None)) for i in range(len(gens))]
        while live:
            for i in list(live):
                try:
                    with ctx[i]():
                        next(gens[i])
                except StopIteration as e:
                    res[i] = e.value
                    live.remove(i)
        if laned:
            lanes.join(self.dev, forget=False)        # the results (losses, logits) are the caller's to use on ITS stream
        return res

    def steps(self, env, obs, feedback="teacher", train_ml=1.0, rw_seq=None, sample_draws=None, grad=True, record=False, text_copies=1, slot=0,
              ahead=None):
        """Generator form of one batch of episodes: yields after launching each step, returns the result dict.

        feedback / train_ml may be per-episode lists: the two rollouts of a fine-tuning iteration (teacher-forced with ml_weight,
        then 'sample' with weight 1 on the SAME episodes; agent_base.py:243-250) are independent per episode, so they can run as one
        batch of 2B episodes -- half the launches, twice the rows per launch.  text_copies = k: the batch is k copies of the same
        B / k instructions (that case): the text encoder and the K/V projections run once on B / k and are tiled.  The loss is
        divided by B / text_copies, i.e. it equals the sum of the separate rollouts' losses.
        slot: which static instruction slot the captured step instances of this rollout read (`graphs=True`): rollouts whose autograd
        graphs are alive at the same time (the iteration's two) take different slots.
        ahead: a `plan_ahead` handle for this (env, obs) -- teacher forcing only: the step plans were built ahead of time."""
        st, te, dev = self.student, self.teacher, self.dev
        B = len(obs)
        per_episode = not isinstance(feedback, str)
        fb = list(feedback) if per_episode else [feedback] * B
        needs_action = any(f != "teacher" for f in fb)
        Bn = B // text_copies                                    # loss normaliser (the reference's batch_size of ONE rollout)
        w_host = [float(train_ml)] * B if isinstance(train_ml, (int, float)) else [float(x) for x in train_ml]
        w_ml = torch.tensor(w_host, dtype=torch.float32, device=dev)
        tt_grad = self.train_teacher and grad
        use_g = self._use_graphs(obs, grad, text_copies)
        plans = None
        if ahead is not None:
            if needs_action:
                raise ValueError("plan_ahead serves teacher-forced rollouts only (any other feedback needs the model's actions step by step)")
            pl, plans = ahead.result()
        else:
            pl = self._planner(env, obs, feedback, grad, use_g)
        lang = pl.language()
        Lt = lang["txt_ids"].shape[1]   # the batch's own padded extent: what the distillation terms are reduced over (kd_loss.py: sums / means run over the padded batch)
        if use_g:                       # static shapes: the instruction padded to Lcap tokens (masked)
            ids = np.zeros((B, self.Lcap), np.int64)
            ids[:, :lang["txt_ids"].shape[1]] = lang["txt_ids"]
            lang = dict(lang, txt_ids=ids)
        ld = to_device(dict(txt_ids=lang["txt_ids"][:Bn]), dev)
        L = lang["txt_ids"].shape[1]
        txt_lens = [int(x) for x in lang["txt_lens"]]
        # per-rollout host -> device traffic in ONE pinned copy (a pageable `as_tensor(..., device=)` per step is a synchronous copy: 10 ms per iteration)
        once = dict(txt_lens=np.asarray(lang["txt_lens"][:Bn], np.int64), is_smp=np.array([f == "sample" for f in fb], np.bool_))
        if sample_draws is not None:
            once["draws"] = np.asarray(sample_draws, np.float64)
        od = to_device(once, dev)
        u_masks = (torch.arange(L, device=dev)[None] < od["txt_lens"][:, None])
        txt_masks = u_masks.repeat(text_copies, 1) if text_copies > 1 else u_masks
        lin = dict(txt_ids=ld["txt_ids"], txt_masks=u_masks)
        tile = (lambda x, dim=0: torch.cat([x] * text_copies, dim)) if text_copies > 1 else (lambda x, dim=0: x)
        ctxg = torch.enable_grad() if grad else torch.no_grad()
        sgs = sgt = None
        with ctxg:
            if use_g:
                sgs = self.step_graphs(st, B)
                sgs = sgs if sgs.usable() else None
            txt_embeds, txt_attns = st("language", lin)
            if sgs is not None:
                sl = sgs.text_slot(slot)
                sl.version += 1
                sl.masks.copy_(txt_masks)
                txt_embeds, txt_attns = tile(txt_embeds), tile(txt_attns)
                txt_kv = st.text_kv(txt_embeds, out=sl.kv)            # (text_copies > 1: projected on the tiled rows, into the slot's static cache)
                s_tok = sgs.kv_token(slot, txt_kv)
            else:
                txt_kv = st.text_kv(txt_embeds) if self.cache_text_kv else None      # once per episode, not once per step
                txt_embeds, txt_attns = tile(txt_embeds), tile(txt_attns)
                txt_kv = tile(txt_kv, 1) if txt_kv is not None else None
        kdv = (lambda x, *ext: x) if not use_g else _extent      # distillation sees the batch's own extent, not the static shapes' padding
        s_out = dict(txt_embeds=kdv(txt_embeds, Lt), txt_attns=kdv(txt_attns, None, Lt, Lt))
        t_out = {}
        tctx = torch.enable_grad if tt_grad else torch.no_grad
        t_ml_loss, t_kdl = torch.zeros((), dtype=torch.float32, device=dev), {}
        if te is not None:
            with tctx():
                if use_g and tt_grad:
                    sgt = self.step_graphs(te, B)
                    sgt = sgt if sgt.usable() else None
                t_txt, t_txt_attns = te("language", lin)
                if sgt is not None:
                    tsl = sgt.text_slot(slot)
                    tsl.version += 1
                    tsl.masks.copy_(txt_masks)
                    t_txt, t_txt_attns = tile(t_txt), tile(t_txt_attns)
                    t_kv = te.text_kv(t_txt, out=tsl.kv)
                    t_tok = sgt.kv_token(slot, t_kv)
                else:
                    t_kv = te.text_kv(t_txt) if self.cache_text_kv else None
                    t_txt, t_txt_attns = tile(t_txt), tile(t_txt_attns)
                    t_kv = tile(t_kv, 1) if t_kv is not None else None
            t_out = dict(txt_embeds=kdv(t_txt, Lt), txt_attns=kdv(t_txt_attns, None, Lt, Lt))
            t_log = EmbeddingLog(te.net.H, te.net.dtype, dev)
        s_log = EmbeddingLog(st.net.H, st.net.dtype, dev)
        ml_loss = torch.zeros((), dtype=torch.float32, device=dev)
        kdl = {}
        stop_probs, steps = [], []
        decisions = 0
        with ctxg:
            for t in range(self.T):
                _tk("(outside)")
                if plans is not None:
                    plan = plans[t]
                    decisions += plan["_live"]
                else:
                    plan = pl.begin_pano()
                    decisions += int((~pl.ended).sum())
                _tk("begin_pano")
                vl = np.asarray(plan["view_lens"])
                pano_arrays = dict(vp_rows=plan["vp_rows"], view_order=plan["view_order"], loc_fts=plan["loc_fts"],
                                   nav_types=np.asarray(plan["nav_types"]).astype(np.int32), view_lens=vl.astype(np.int32),
                                   pano_masks=np.arange(plan["V"])[None] < vl[:, None])      # dtypes the kernels read: no casts on the device
                pi = sgs.pano_inst(plan["V"]) if sgs is not None else None
                tpi = None
                if te is not None and sgt is not None:
                    with tctx():
                        tpi = sgt.pano_inst(plan["V"])
                d, pin = {}, None
                if pi is None or (te is not None and tpi is None):
                    d = to_device(pano_arrays, dev)
                    pin = self._pano_inputs(d, plan)
                pe, pm, pf, pa = sgs.run_pano(pi, pano_arrays) if pi is not None else st("panorama", pin)
                Vt = plan["V_valid"]
                s_out.update(pano_embeds=kdv(pe, Vt), pano_fused_embeds=pf, img_attns=kdv(pa, Vt, Vt))
                if te is not None:
                    with tctx():
                        tpe, _, tpf, tpa = sgt.run_pano(tpi, pano_arrays) if tpi is not None else te("panorama", pin)
                _tk("pano arrays + launch")
                # the GPU is busy with the panorama encoder(s): build the second half of the plan now
                if plans is None:
                    plan.update(pl.begin_nav())
                _tk("begin_nav")
                fixed = {k: plan[k] for k in ("gmap_pos_fts", "gmap_pair_dists", "gmap_masks", "vp_pos_fts", "vp_nav_masks", "vp_masks", "fsrc", "bw")}
                fixed["gmap_step_ids"] = np.asarray(plan["gmap_step_ids"]).astype(np.int32)
                fixed["gmap_logit_masks"] = ~np.asarray(plan["gmap_visited_masks"], bool) & np.asarray(plan["gmap_masks"], bool)
                arrays = dict(targets=plan["targets"], csr_ptr=plan["csr"][0], csr_idx=plan["csr"][1], csr_w=plan["csr"][2])
                if plan["csr_t"] is not None:
                    arrays.update(csrt_ptr=plan["csr_t"][0], csrt_idx=plan["csr_t"][1], csrt_w=plan["csr_t"][2])
                ni = sgs.nav_inst(plan["K"], plan["Vp"], slot) if sgs is not None else None
                tni = None
                if te is not None and sgt is not None:
                    with tctx():
                        tni = sgt.nav_inst(plan["K"], plan["Vp"], slot)
                if ni is None or (te is not None and tni is None):      # somebody runs this step eagerly: the fixed-shape arrays travel with the rest
                    arrays.update(fixed, gmap_visited_masks=plan["gmap_visited_masks"])
                d2 = to_device(arrays, dev)
                d2["_stage2"] = d2.pop("_stage")
                d.update(d2)
                csr = (d["csr_ptr"], d["csr_idx"], d["csr_w"])
                csr_t = (d["csrt_ptr"], d["csrt_idx"], d["csrt_w"]) if plan["csr_t"] is not None else None
                _tk("nav arrays + to_device")
                s_log.put(plan["log_base"], pe)
                s_log.put(plan["log_fused"], pf)
                gathered = s_log.gather(csr, csr_t, plan["n_out"], plan["log_cls"], grad=grad, out=ni.gathered if ni is not None else None)
                if ni is not None:
                    outs = sgs.run_nav(ni, fixed, gathered, s_tok)
                else:
                    outs = st("navigation", self._nav_inputs(d, plan, gathered, txt_embeds, txt_masks, txt_lens, txt_kv))
                _tk("log put/gather + nav launch")
                s_log.put(plan["log_cls"], outs["cls_embeds"])
                logits = outs["fused_logits"]
                Kt = plan["K_valid"]
                s_out.update(nav_outs=_nav_extent(outs, Kt, Vt + 2, Lt) if use_g else outs, nav_logits=kdv(logits, Kt))
                targets = d["targets"]
                ce = ce_rows_loss(logits, targets, IGNORE)
                ml_loss = ml_loss + (ce * w_ml).sum()
                stop_probs.append(torch.softmax(logits.detach(), 1)[:, 0])
                if te is not None:
                    with tctx():
                        t_out.update(pano_embeds=kdv(tpe, Vt), pano_fused_embeds=tpf, img_attns=kdv(tpa, Vt, Vt))
                        t_log.put(plan["log_base"], tpe, track=tt_grad)
                        t_log.put(plan["log_fused"], tpf, track=tt_grad)
                        tg = t_log.gather(csr, csr_t, plan["n_out"], plan["log_cls"], grad=tt_grad, out=tni.gathered if tni is not None else None)
                        if tni is not None:
                            t_outs = sgt.run_nav(tni, fixed, tg, t_tok)
                        else:
                            t_outs = te("navigation", self._nav_inputs(d, plan, tg, t_txt, txt_masks, txt_lens, t_kv))
                        t_log.put(plan["log_cls"], t_outs["cls_embeds"], track=tt_grad)
                        t_out.update(nav_outs=_nav_extent(t_outs, Kt, Vt + 2, Lt) if use_g else t_outs, nav_logits=kdv(t_outs["fused_logits"], Kt))
                        t_ce = ce_rows_loss(t_outs["fused_logits"], targets, IGNORE)
                        t_out["sample_weights"] = exponential_decay(t_ce.detach(), self.kd["decay"])
                    if grad:
                        learned = self.kd.get("ability_weight") == "learned_weight"
                        rw_t = None if (rw_seq is None or learned) else rw_seq[t]
                        if learned or not FUSED_MAKD:
                            kdl = compute_kd_losses(t, s_out, t_out, self.heads, kdl, role="t2s", temperature=self.kd["temperature"], weights=rw_t,
                                                    learned=st.vln_bert if learned else None)
                        else:            # the step's nine mse terms in one launch / one autograd node, the running sums as one vector
                            kdl = compute_kd_losses_fused(t, s_out, t_out, self.heads, kdl, role="t2s", temperature=self.kd["temperature"], weights=rw_t)
                        if tt_grad:      # reverse direction (agent.py:1026): teacher tensors vs the student's, projected by the student's heads
                            t_ml_loss = t_ml_loss + t_ce.sum()
                            s_out["sample_weights"] = exponential_decay(ce.detach(), self.kd["decay"])
                            if learned or not FUSED_MAKD:
                                t_kdl = compute_kd_losses(t, t_out, s_out, self.heads, t_kdl, role="s2t", temperature=self.kd["temperature"],
                                                          weights=rw_t, learned=te.vln_bert if learned else None)   # s_model = the teacher (:555-556)
                            else:
                                t_kdl = compute_kd_losses_fused(t, t_out, s_out, self.heads, t_kdl, role="s2t", temperature=self.kd["temperature"], weights=rw_t)
                _tk("loss + distillation terms")
                yield t           # the step is launched; nothing below is needed before its actions are (run_interleaved switches here)
                _tk("(outside)")
                a_host = None
                if needs_action:                                                     # the stepper needs it: one [B] copy
                    a_arg = logits.detach().argmax(1)
                    if any(f == "sample" for f in fb):
                        cdf = torch.softmax(logits.detach(), 1).double().cumsum(1)
                        u = od["draws"][t]
                        a_smp = (cdf < (u * cdf[:, -1])[:, None]).sum(1).clamp(max=logits.shape[1] - 1)
                        a_arg = torch.where(od["is_smp"].view(torch.bool), a_smp, a_arg)
                    _tk("action ops")
                    a_host = a_arg.cpu().numpy()
                    _tk("action copy (waits for the GPU)")
                if record:
                    steps.append(dict(logits=logits.detach().float().cpu(), targets=torch.from_numpy(plan["targets"]).clone(),
                                      vpids=plan["gmap_vpids"]))
                done = plan["_done"] if plans is not None else pl.end_step(a_host)
                if record:
                    steps[-1]["actions"] = list(plan["_actions"] if plans is not None else pl.actions)
                _tk("end_step")
                if done:
                    break
        ml = ml_loss / Bn
        kd_sum, total = None, ml
        if te is not None and grad:
            kdl = kd_terms(kdl)
            kd_sum = sum(kdl.values()) / Bn
            total = self.kd["alpha"] * kd_sum + (1 - self.kd["alpha"]) * ml
        traj = pl.finish(torch.stack(stop_probs).cpu().numpy())
        out = dict(loss=total, ml_loss=ml, kdl=kd_sum, kdl_terms=kdl, traj=traj, n_steps=len(stop_probs), decisions=decisions,
                   steps=steps, planner=pl)
        if tt_grad:
            ta = self.kd.get("t_alpha", self.kd["alpha"])
            t_kdl = kd_terms(t_kdl)
            out["t_kdl_terms"] = t_kdl
            out["t_loss"] = ta * (sum(t_kdl.values()) * w_host[0]) + (1 - ta) * (t_ml_loss * w_host[0] / Bn)       # (uniform train_ml)
        return out
