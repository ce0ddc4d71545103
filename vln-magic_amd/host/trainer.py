"""One optimizer step of MAGIC pretraining with MAKD (the loop the reference constructs but never runs,
pretrain_src/train_r2r_magic.py:358-401; reconstructed order in SURVEY §3.1):

    teacher forward (no grad, kdl.train_teacher=false) -> student forward + supervised + MAKD losses ->
    explicit backward -> [DDP: all-reduce of the flat gradient buffer] -> grad-norm clip -> AdamW -> lr schedule

Optimizer arithmetic: pretrain_src/optim/adamw.py:53-112 (two param groups, optim/misc.py:13-22),
schedule pretrain_src/optim/sched.py:17-30, clip 5.0 / betas (0.9,0.98) / wd 0.01 from
pretrain_src/config/r2r_magic_pretrain.json:14-23.

Every per-step scalar (lr, Adam bias correction, MKRW ability weights) lives in device memory, so the whole step
can be captured once per batch-shape into a HIP graph and replayed (`PretrainStep.capture`): ~660 kernel launches
cost one graph launch on the host.  The teacher forward runs on a side stream concurrently with the student
forward (they are independent until the distillation losses).
"""
import math
import os

import torch
import torch.distributed as dist

from .lib import capture as _capture
from . import ops as O
from .plan import build_plan


def warmup_linear(step, warmup, total):
    if step < warmup:
        return step / warmup
    return max(0, (total - step) / (total - warmup))


def get_lr_sched(step, lr, warmup, total):
    v = lr * warmup_linear(step, warmup, total)
    return v if v > 0 else 1e-8


class FusedAdamW:
    """AdamW over the ParamStore's flat buffers: one sum-of-squares launch + one launch per decay group.
    With `schedule=(warmup, total)` the lr / bias-correction scalars are produced on the device by
    `magic_sched_step` (graph-replayable); otherwise they are host floats."""

    def __init__(self, store, lr=5e-5, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, max_grad_norm=5.0, schedule=None):
        self.store, self.lr, self.betas, self.eps, self.wd, self.max_norm = store, lr, betas, eps, weight_decay, max_grad_norm
        self.ss = torch.zeros(1, dtype=torch.float32, device=store.device)
        # steps the AdamW kernel SKIPPED because the gradient norm was not finite (fp16 storage: an overflowed activation gradient); read
        # with `skipped_steps()` at a point that may synchronise -- the reference's fp16 option skips such steps too (amp.GradScaler)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=store.device)
        # dynamic loss scale (fp16 storage): the model's fp32[4] state {S, 1/S, clean steps, pending} (model_pretrain.enable_dynamic_loss_scale);
        # the AdamW launch divides by S and reports updated / skipped in `pending`, the next step's prologue (ops.step_rng) applies GradScaler's
        # rule.  With it the gradient norm is ALWAYS computed -- clipping or not -- so an overflowed gradient is skipped, never applied.
        self.loss_scale = None
        self.t = 0
        self.schedule = schedule
        if schedule is not None:
            # [0] = global_step (the lr schedule's: advances on every step, skipped or not), [1] = the optimizer's state step (Adam's bias correction:
            # taken back by the AdamW launch when it skips a non-finite step) -- csrc/optim.hip sched_advance
            self.step_dev = torch.zeros(2, dtype=torch.int32, device=store.device)
            self.lr_ss = torch.zeros(2, dtype=torch.float32, device=store.device)

    def skipped_steps(self):
        """optimizer steps skipped so far because of a non-finite gradient norm (one device -> host copy)"""
        return int(self.overflow.item())

    def grad_norm_report(self):
        """the LAST optimizer step's global gradient norm and the factor clip_grad_norm_ scaled it by (train_r2r_magic.py:373-375), from the device words that
        step left behind: the sum-of-squares accumulator (zeroed only by the NEXT step's prologue), the pre-scale the update used (1 / world, 1 / static
        gradient scale, 1 / accumulation) and the dynamic loss scale's 1 / S.  Synchronises.  None before the first step or without a norm."""
        if self._last_gscale is None or not (self.max_norm or self.loss_scale is not None):
            return None
        g = self._last_gscale * (float(self.loss_scale[1].item()) if self.loss_scale is not None else 1.0)
        nrm = math.sqrt(max(float(self.ss.item()), 0.0)) * g
        rep = {"grad_norm": float(f"{nrm:.4e}"), "max_grad_norm": self.max_norm}
        if self.max_norm:
            rep["clip_factor"] = float(f"{min(1.0, self.max_norm / (nrm + 1e-6)):.4e}") if math.isfinite(nrm) else 0.0
        return rep

    _last_gscale = None

    def step(self, lr=None, gscale=1.0, ss_zeroed=False, zero_grad=False):
        """ss_zeroed: the gradient-norm accumulator was zeroed earlier in this step (PretrainStep's prologue launch) -- the schedule then
        rides in the sum-of-squares launch; zero_grad: the AdamW kernel zeroes the gradient buffer after consuming it"""
        s = self.store
        lr = self.lr if lr is None else lr
        self.t += 1
        self._last_gscale = float(gscale)
        b1, b2 = self.betas
        lr_ss = None
        use_clip = self.max_norm is not None and self.max_norm > 0
        need_ss = use_clip or self.loss_scale is not None
        fused_sched = self.schedule is not None and need_ss and ss_zeroed
        if fused_sched:                        # schedule + gradient norm: one launch
            O.sumsq_sched(s.grad, self.ss, self.step_dev, self.lr, self.schedule[0], self.schedule[1], b1, b2, self.lr_ss)
            lr_ss, step_size = self.lr_ss, 0.0
        elif self.schedule is not None:        # (the schedule kernel also zeroes the gradient-norm accumulator of this step)
            O.sched_step(self.step_dev, self.lr, self.schedule[0], self.schedule[1], b1, b2, self.lr_ss, zero_me=self.ss if need_ss else None)
            lr_ss, step_size = self.lr_ss, 0.0
        else:
            # (host-side scalars, no schedule: the navigator's / the tests' form.  `t` is a host count, so a step the kernel skips still advances the bias
            # correction here -- knowing would cost a device -> host read per step; fp16 training with loss scaling runs the device-side schedule above)
            step_size = lr * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
            if need_ss:
                self.ss.zero_()
        if need_ss and not fused_sched:
            O.sumsq(s.grad, self.ss)
        shadow = s.shadow if s.half else None
        # both parameter groups (decay | no decay: contiguous in the flat buffer) in ONE launch
        O.adamw(s.total, s.flat, s.grad, s.m, s.v, shadow, lr, b1, b2, self.eps, self.wd, step_size, self.ss if need_ss else None,
                self.max_norm if use_clip else 0.0, gscale, lr_ss=lr_ss, n_decay=s.n_decay, zero_grad=zero_grad, overflow=self.overflow,
                scale_state=self.loss_scale, sched_step=self.step_dev if self.schedule is not None else None)
        s.shadow_clean = True
        if shadow is not None and s.t_spans:           # the AdamW kernel rewrote the bf16 shadow: its transposed copy follows
            s.sync_shadow_t(force=True)
        if shadow is not None and s.f_spans:      # ... and the fragment-order copies the whole-encoder kernels read
            s.sync_shadow_f(force=True)
        return lr


class FlatTorchAdamW:
    """`torch.optim.AdamW(model.parameters(), lr)` + `clip_grad_norm_(model.parameters(), max_norm)` of the navigator's loop
    (map_nav_src/r2r/agent_base.py:122-137, :273) over a ParamStore's flat buffers: one sum-of-squares launch + one update launch for the
    whole model (clip, decay-first AdamW, 16-bit shadow refresh, gradient zeroing) instead of torch's per-tensor foreach chains -- 4 ms of
    kernels and 5 ms of host per MAGIC-L iteration.  Same arithmetic as torch.optim.AdamW: p *= 1 - lr wd; m, v; p -= lr / bc1 * m /
    (sqrt(v / bc2) + eps)  (tests/test_glue_gpu.py compares three steps with torch's)."""

    def __init__(self, store, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.store, self.lr, self.betas, self.eps, self.wd = store, lr, betas, eps, weight_decay
        self.ss = torch.zeros(1, dtype=torch.float32, device=store.device)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=store.device)
        self.t = 0

    def zero_grad(self, set_to_none=True):
        """(the update launch zeroes the gradient buffer it consumed: nothing to do after a step; before the first one the buffer is zero)"""
        if self.t == 0 or not self._clean:
            self.store.zero_grad()
        self._clean = False

    _clean = False

    def step(self, max_norm=None):
        s = self.store
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        clip = max_norm is not None and max_norm > 0
        if clip:
            self.ss.zero_()
            O.sumsq(s.grad, self.ss)
        shadow = s.shadow if s.half else None
        O.adamw(s.total, s.flat, s.grad, s.m, s.v, shadow, self.lr, b1, b2, self.eps * math.sqrt(bc2), self.wd, self.lr * math.sqrt(bc2) / bc1,
                self.ss if clip else None, float(max_norm) if clip else 0.0, 1.0, n_decay=-1, zero_grad=True, overflow=self.overflow, decay_first=True)
        self._clean = True
        s.shadow_clean = True
        if shadow is not None and s.t_spans:
            s.sync_shadow_t(force=True)
        if shadow is not None and s.f_spans:
            s.sync_shadow_f(force=True)


# parameters whose gradients are final once the cross-modal half of the backward pass is over (heads, both co-attention encoders,
# their input embeddings, the distillation projections): everything from `global_encoder` on in storage order (engine.trunk_specs)
LATE_PREFIXES = ("bert.global_encoder.", "vln_bert.global_encoder.")
EMB_TABLE = "bert.embeddings.word_embeddings.weight"


def cu_mask_words(spec, n_cu=256):
    """`N` | `N:low` | `N:spread` | `N:xcd` -> the 32-bit words of a compute-unit mask with N bits set (hipExtStreamCreateWithCUMask: bit i =
    CU i): low = the first N, spread = every (n_cu / N)-th, xcd = the first N / 8 of every block of n_cu / 8"""
    n, _, how = str(spec).partition(":")
    n = max(1, min(int(n), n_cu))
    how = how or "spread"
    if how == "low":
        bits = range(n)
    elif how == "xcd":
        per = max(1, n // 8)
        bits = [x * (n_cu // 8) + i for x in range(8) for i in range(per)]
    elif how == "spread":
        bits = [i * n_cu // n for i in range(n)]
    else:
        raise ValueError(f"compute-unit mask pattern {how!r}")
    words = [0] * ((n_cu + 31) // 32)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    return words


def _side_stream(dev, others=()):
    """the frozen teacher's stream: one that is proven to run BESIDE the student's (and the gradient exchange's) -- lanes.beside: which
    hardware queue a stream lands on is an accident of how many streams the process made before it, and a teacher stream sharing the
    student's queue runs in order with it (its start gate then only ever times out).  MAGIC_TEACHER_CUS=N[:pattern] confines it to N
    compute units instead (a queue property, honoured under graph replay as long as the graph is launched on this stream): the teacher is
    off the critical path, its launches then stop taking CUs from the student's in bursts.  Opt-in: see DESIGN.md section 5 for what it
    measured."""
    spec = os.environ.get("MAGIC_TEACHER_CUS")
    if not spec:
        from . import lanes
        pr = os.environ.get("MAGIC_TEACHER_PRIORITY")
        return lanes.beside([torch.cuda.current_stream(dev)] + [o for o in others if o is not None], device=dev, priority=int(pr) if pr else None)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    words = cu_mask_words(spec, n_cu)
    arr = (ctypes.c_uint32 * len(words))(*words)
    h = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), len(words), arr)
    if rc != 0 or not h.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask({spec}) failed: {rc}")
    return torch.cuda.ExternalStream(h.value, device=dev)


class GradSync:
    """Data-parallel gradient exchange on the flat fp32 gradient buffer, RCCL through torch.distributed, on a side stream; the
    1/world scaling is folded into the AdamW kernel.  Replaces DDP(model, find_unused_parameters=True) of
    pretrain_src/utils/misc.py:62-63: every rank runs the same task each step (data/loader.py:55-59), so task-unused
    parameters simply contribute zeros.

    Buckets follow the ORDER IN WHICH THE EXPLICIT BACKWARD FINISHES GRADIENTS (what DDP's reverse-registration buckets
    approximate): bucket 0 = heads + both cross-modal encoders + distillation projections, final when the backward enters the
    text / panorama encoders; bucket 1 = the top two text blocks + the panorama blocks, final two rounds into that half (their
    weight-gradient GEMMs are flushed there: model_pretrain.MID_CUT); bucket 2 = the lower text blocks + all embeddings, final at
    the end.  `reduce_bucket(0)` and `reduce_bucket(1)` are issued from inside the backward (model.backward(on_bucket=...)) and run on
    the side stream under the rest of the backward pass; only bucket 2 is exposed.
    On steps that only read the word-embedding table through the instruction tokens (sap / cfp / mrc: <= B*L of its 50 265 rows
    carry gradient, 65 % of all gradient bytes otherwise) the last bucket exchanges those ROWS (all-gather of row ids + rows, local
    scatter-add) instead of the dense table."""

    def __init__(self, store, chunk_elems=8 << 20, overlap=None, sparse_rows_cap=None):
        self.store = store
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # MAGIC_DP_STRUCTURE=1 (bench.py --dp-structure): run the data-parallel STRUCTURE -- three backward graphs cut at the bucket boundaries, the
        # bucket collectives issued on the exchange stream between the replays, the optimizer's graph behind the exchange -- in a world-1 `nccl`
        # group, where every collective is the identity: what the cuts and RCCL's launch latency cost is measurable without a second GPU
        self.force = self.world == 1 and bool(os.environ.get("MAGIC_DP_STRUCTURE")) and dist.is_available() and dist.is_initialized()
        self.chunk = chunk_elems
        self.card_shared = self._ranks_share_a_card()
        self.stream = None
        if (self.world > 1 or self.force) and store.device.type == "cuda":
            from . import lanes
            self.stream = lanes.beside([torch.cuda.current_stream(store.device)], device=store.device)     # the exchange runs UNDER the backward
        self.overlap = (not os.environ.get("MAGIC_DDP_NO_OVERLAP")) if overlap is None else overlap
        # round 6: RCCL through its C ABI on the exchange stream (host/rccl.py) -- capturable inside the step's graph, no per-call bookkeeping; None:
        # torch.distributed's calls (gloo rehearsals, MAGIC_RCCL_DIRECT=0, or a communicator that failed its self-test)
        self.rccl = None
        if self.stream is not None:
            from . import rccl as _rccl
            self.rccl = _rccl.make(store.device)
        g0d, g0n = store.first_offset(LATE_PREFIXES, True), store.first_offset(LATE_PREFIXES, False)
        nd, tot = store.n_decay, store.total
        first, rest = [(g0d, nd), (g0n, tot)], [(0, g0d), (nd, g0n)]
        # the middle bucket: what the shared text / panorama backward has finished after its first MID_CUT rounds -- the top MID_CUT text
        # blocks and the top MID_CUT panorama blocks (model_pretrain.backward_phase2's on_cut); the last bucket: the rest (lower text
        # blocks, embeddings, image embeddings).  Stores without such blocks (the navigator's) keep two buckets + an empty middle one.
        mid = store.ranges_of(mid_prefixes(store)) if mid_prefixes(store) else []
        mid = [(max(a, lo), min(b, hi)) for a, b in mid for lo, hi in rest if max(a, lo) < min(b, hi)]
        self.buckets = [first, mid, _subtract(rest, mid)]
        self.table = store.offsets.get(EMB_TABLE)            # (offset, numel, (rows, H)) -- first tensor of the buffer
        self.sparse_cap = sparse_rows_cap
        self._pending = []

    def _ranks_share_a_card(self):
        """do two ranks of this job sit on one physical device (a rehearsal on a one-card box)?  Then the launch layer must not use the
        forms that need a whole launch resident at once (ops.card_is_shared): decided once, the same way on every rank, before anything
        is captured."""
        dev = self.store.device
        if self.world == 1 or dev.type != "cuda":
            return False
        import hashlib
        import socket
        props = torch.cuda.get_device_properties(dev)
        me = repr((socket.gethostname(), str(getattr(props, "uuid", "")), getattr(props, "pci_domain_id", -1), getattr(props, "pci_bus_id", -1),
                   getattr(props, "pci_device_id", -1), os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")), dev.index))
        # a plain int64 all-gather on the device, like every other exchange of this class (no pickled-object collective on the data path)
        mine = torch.tensor([int.from_bytes(hashlib.sha1(me.encode()).digest()[:7], "big")], dtype=torch.int64, device=dev)
        seen = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(seen, mine)
        shared = len({int(t) for t in seen}) < self.world
        if shared:
            O.card_is_shared(True)
        return shared

    # ---- primitives ------------------------------------------------------------------------------------------
    def _ranges(self, ranges):
        g = self.store.grad
        if self.rccl is not None:             # one group launch for all the chunks of the bucket
            with self.rccl.group():
                for lo, hi in ranges:
                    for a in range(lo, hi, self.chunk):
                        self.rccl.all_reduce_(g[a:min(hi, a + self.chunk)])
            return
        for lo, hi in ranges:
            for a in range(lo, hi, self.chunk):
                dist.all_reduce(g[a:min(hi, a + self.chunk)])

    def _on_side(self, fn):
        """run fn on the exchange stream after everything queued on the current stream so far"""
        if self.stream is None:
            fn()
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        if O._EARLY["stream"] is not None:        # the bucket's weight gradients were flushed on the weight-gradient stream (ops.flush_dw_early)
            self.stream.wait_stream(O._EARLY["stream"])
        with torch.cuda.stream(self.stream):
            fn()

    def join(self):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    def _sparse_rows(self, row_ids, cap=None):
        """sum over ranks of the gradient rows `row_ids` (this rank's touched rows: int64 on the device, any order, UNIQUE -- the
        plan's `torch.unique(txt_ids)`) of the word-embedding table: all-gather ids + rows, then every rank rebuilds the touched
        rows as 0 + rank 0's rows + rank 1's rows + ... in rank order.  One rank's ids never collide inside one index_add_, so the
        fp32 sum order is the same on every rank and the replicas stay bitwise identical, as after an all-reduce."""
        off, n, (R, H) = self.table
        tab = self.store.grad[off:off + n].view(R, H)
        cap = self.sparse_cap if cap is None else int(cap)          # THE SAME ON EVERY RANK (gathered buffers are cap-sized)
        ids = torch.zeros(cap, dtype=torch.int64, device=tab.device)
        k = int(row_ids.numel())
        if k > cap:
            raise ValueError(f"sparse embedding exchange: {k} touched rows > cap {cap}")
        ids[:k] = row_ids
        rows = tab.index_select(0, ids)
        rows[k:] = 0                                              # padding slots point at row 0 and carry zeros (x + 0 in any order)
        all_ids = torch.empty(self.world * cap, dtype=torch.int64, device=tab.device)
        all_rows = torch.empty(self.world * cap, H, dtype=tab.dtype, device=tab.device)
        if self.rccl is not None:             # both gathers as one group launch
            with self.rccl.group():
                self.rccl.all_gather(all_ids, ids)
                self.rccl.all_gather(all_rows.view(-1), rows.view(-1))
        else:
            dist.all_gather_into_tensor(all_ids, ids)
            dist.all_gather_into_tensor(all_rows, rows)
        tab.index_fill_(0, row_ids, 0)                            # own contribution comes back through all_rows, in its rank's turn
        for r in range(self.world):
            tab.index_add_(0, all_ids[r * cap:(r + 1) * cap], all_rows[r * cap:(r + 1) * cap])

    def _sparse_rows_padded(self, ids_pad):
        """`_sparse_rows` over a STATIC id buffer (round 6: the form a captured graph can hold when the touched rows change per replay -- streamed
        batches): int64 [cap] on the device, this step's unique row ids first, -1 behind them.  Nothing here depends on how many ids are valid: pads
        gather row 0 with a zero weight and add zeros back into it; `index_fill_` then also clears row 0, whose own contribution (if it was touched
        on this rank) comes back through the gathered rows like every other row's, and whose gradient is zero otherwise (sparse steps reach the table
        only through the instruction lookup).  Same rank-order sums as `_sparse_rows`: bitwise-identical replicas."""
        off, n, (R, H) = self.table
        tab = self.store.grad[off:off + n].view(R, H)
        cap = int(ids_pad.numel())
        valid = ids_pad >= 0
        idx = ids_pad.clamp_min(0)
        rows = tab.index_select(0, idx) * valid.to(tab.dtype)[:, None]
        all_ids = torch.empty(self.world * cap, dtype=torch.int64, device=tab.device)
        all_rows = torch.empty(self.world * cap, H, dtype=tab.dtype, device=tab.device)
        if self.rccl is not None:
            with self.rccl.group():
                self.rccl.all_gather(all_ids, idx)
                self.rccl.all_gather(all_rows.view(-1), rows.view(-1))
        else:
            dist.all_gather_into_tensor(all_ids, idx)
            dist.all_gather_into_tensor(all_rows, rows)
        tab.index_fill_(0, idx, 0)
        for r in range(self.world):
            tab.index_add_(0, all_ids[r * cap:(r + 1) * cap], all_rows[r * cap:(r + 1) * cap])

    # ---- per-bucket API (called from inside the backward) ----------------------------------------------------------
    def reduce_bucket(self, i, touched_rows=None, cap_scale=1, padded=False):
        """launch bucket i's exchange on the side stream.  touched_rows (last bucket only): device int64 ids of the word-embedding
        rows this rank's step wrote, or None for a dense table (mlm: the tied decoder touches every row).  cap_scale: the gathered
        buffers hold cap_scale x sparse_rows_cap rows (gradient accumulation: the rows of cap_scale micro-batches)."""
        if self.world == 1 and not self.force:
            return
        ranges = self.buckets[i]
        # an EMPTY id list is not "no rows": bucket-padded plans carry a zero-length placeholder (host/plan.py) -> dense exchange
        last = len(self.buckets) - 1          # the word-embedding table lives in the last bucket
        # padded: touched_rows is a static -1-padded id buffer (`_sparse_rows_padded`): always the sparse form, whatever the buffer holds this replay
        sparse = i == last and touched_rows is not None and (padded or touched_rows.numel() > 0) and self.table is not None and self.sparse_cap
        if sparse:
            off, n, _ = self.table
            assert off == 0 and ranges[0][0] == 0
            ranges = [(n, ranges[0][1])] + ranges[1:]

        def run():
            self._ranges(ranges)
            if sparse and padded:
                self._sparse_rows_padded(touched_rows)
            elif sparse:
                self._sparse_rows(touched_rows, cap=self.sparse_cap * cap_scale)
        self._on_side(run)

    def all_reduce(self):
        """monolithic form: the whole flat buffer after the backward pass; returns the 1/world factor for the optimizer"""
        if self.world == 1 and not self.force:
            return 1.0
        self._on_side(lambda: self._ranges([(0, self.store.total)]))
        self.join()
        return 1.0 / self.world

    def finish(self):
        """after the last reduce_bucket: make the main stream wait for the exchange; returns the 1/world factor"""
        self.join()
        return 1.0 / self.world if self.world > 1 else 1.0


def mid_prefixes(store):
    """name prefixes of the blocks that are final after MID_CUT rounds of engine.self_stacks_bwd (two stacks advancing from their tops)"""
    from .model_pretrain import MID_CUT
    out = []
    for fmt in ("bert.lang_encoder.layer.{}.", "bert.img_embeddings.pano_encoder.layer.{}."):
        n = 0
        while any(k.startswith(fmt.format(n)) for k in store.offsets):
            n += 1
        out += [fmt.format(i) for i in range(max(0, n - MID_CUT), n)]
    return out


def _subtract(ranges, holes):
    """ranges minus holes (both lists of disjoint [lo, hi)), sorted"""
    out = []
    for lo, hi in sorted(ranges):
        cur = lo
        for a, b in sorted(holes):
            if b <= cur or a >= hi:
                continue
            if a > cur:
                out.append((cur, a))
            cur = max(cur, b)
        if cur < hi:
            out.append((cur, hi))
    return out


def auto_sync(model):
    """Gradient averaging for UNMODIFIED training loops (`loss.backward(); optimizer.step()`): the explicit HIP backward
    writes `.grad` without going through autograd, so `DistributedDataParallel`'s reducer hooks never fire -- a DDP wrapper
    around these models would silently skip the exchange.  Instead, when a process group with world_size > 1 exists, the end
    of the backward pass all-reduces the flat gradient buffer and applies DDP's 1/world averaging in place.
    `model.auto_grad_sync = False` turns it off (PretrainStep does its own exchange and folds the scale into AdamW)."""
    if not getattr(model, "auto_grad_sync", True):
        return
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    sync = getattr(model, "_grad_sync", None)
    if sync is None:
        sync = model._grad_sync = GradSync(model.store)
    scale = sync.all_reduce()
    model.store.grad.mul_(scale)


def broadcast_task(task_id, device):
    """MetaLoader's one non-gradient collective (pretrain_src/data/loader.py:55-59)."""
    t = torch.tensor([task_id], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=0)
    return int(t.item())


def copy_teacher_outputs(src, dst):
    """dst[k].copy_(src[k]) for every tensor a teacher forward returns (incl. the cast inputs it carries)"""
    for k, v in src.items():
        if torch.is_tensor(v):
            dst[k].copy_(v)
        elif k == "cfp":
            for a, b in zip(v, dst[k]):
                b.copy_(a)
        elif k == "inputs":
            for name in ("feats", "loc", "gpos", "vpos", "dist"):
                a, b = getattr(v, name), getattr(dst[k], name)
                if a.data_ptr() != b.data_ptr():
                    b.copy_(a)


class CapturedStep:
    """A HIP graph of one training step bound to one resident batch (+ its plan).  The graph's kernels address the batch and
    plan tensors directly, so the step owns references to them: dropping the caller's copies must not free memory a replay
    still reads (index arrays read from recycled memory are out-of-bounds accesses on the device)."""

    def __init__(self, graph, out, traj_steps, full, keep=None):
        self.graph, self.out, self.traj_steps, self.full = graph, out, traj_steps, full
        self.keep = keep


class PretrainStep:
    def __init__(self, student, teacher=None, lr=5e-5, betas=(0.9, 0.98), weight_decay=0.01, grad_norm=5.0,
                 warmup_steps=10000, num_train_steps=200000, rw_temp=4.0, seed=0, overlap_teacher=True, overlap_dw=True,
                 sparse_embedding_rows=None, accum_steps=1, dynamic_loss_scale=True, loss_scale_init=None, loss_scale_interval=2000):
        """sparse_embedding_rows: an upper bound, THE SAME ON EVERY RANK, on the distinct token ids of one rank's batch (batch size x
        the loader's instruction truncation length, pretrain_src/config/r2r_magic_pretrain.json:7 max_txt_len).  When given, steps
        that touch the word-embedding table only through the instruction lookup exchange rows instead of the dense table."""
        self.student, self.teacher = student, teacher
        self.opt = FusedAdamW(student.store, lr, betas, 1e-6, weight_decay, grad_norm, schedule=(warmup_steps, num_train_steps))
        self.sync = GradSync(student.store, sparse_rows_cap=sparse_embedding_rows)
        # fp16 storage: amp.GradScaler's dynamic loss scale (train_r2r_magic.py:370-371), kept on the device -- halved after a skipped step, doubled
        # after `loss_scale_interval` updates in a row; the first scale is the model's static `grad_scale` unless given
        self.loss_scale_rule = (2.0, 0.5, int(loss_scale_interval))
        if dynamic_loss_scale and student.store.device.type == "cuda" and getattr(student, "grad_scale", 1.0) != 1.0:
            self.opt.loss_scale = student.enable_dynamic_loss_scale(loss_scale_init)
        self.rw_temp = rw_temp
        self.dev = student.store.device
        self.on_gpu = self.dev.type == "cuda"
        self.side = None
        if self.on_gpu and teacher is not None and overlap_teacher and not os.environ.get("MAGIC_NO_TEACHER_SIDE"):
            self.side = _side_stream(self.dev, (self.sync.stream,))
        if self.on_gpu and O.DW_EARLY and O._EARLY["use"] is None:
            # the stream of the mid-backward weight-gradient flush (ops.flush_dw_early): beside the main, the teacher's and the exchange stream
            from . import lanes as _lanes
            O._EARLY["use"] = _lanes.beside([torch.cuda.current_stream(self.dev)] + [x for x in (self.side, self.sync.stream) if x is not None], device=self.dev)
        if self.on_gpu and overlap_dw and O.SIDE["stream"] is None and os.environ.get("MAGIC_DW_SIDE"):   # opt-in: no gain measured on MI355X
            O.SIDE["stream"] = torch.cuda.Stream()       # weight-gradient GEMMs leave the dX critical chain
        self.global_step = 0
        # gradient accumulation (gradient_accumulation_steps, pretrain_src/parser.py:41-45; MetaLoader keeps one task for accum_steps
        # consecutive batches, data/loader.py:50-59): `step()` runs forward + backward on every micro-batch, accumulating into the flat
        # gradient buffer, and exchanges / clips / updates on the last one with the mean gradient (1 / accum_steps folded into the AdamW
        # kernel's pre-scale, next to 1 / world); the data-parallel exchange is skipped on the other micro-batches, as DDP.no_sync() does
        self.accum_steps = int(accum_steps)
        assert self.accum_steps >= 1
        self._micro = 0
        self._window_rows = []      # word-embedding rows touched by the micro-batches of the running accumulation window (None: dense)
        # per-step random scalars (MKRW weights, dropout seed) come from ONE launch (csrc/loss.hip step_rng_kernel) keyed by `seed` and a
        # device-side step counter: inside a captured graph every replay advances the counter and so draws fresh values
        self.seed = int(seed)
        if self.on_gpu:
            self._rng_counter = torch.zeros(1, dtype=torch.int32, device=self.dev)
            self._rw = torch.ones(5, dtype=torch.float32, device=self.dev)
            self._dseed = torch.zeros(2, dtype=torch.int32, device=self.dev)
            student.dropout_seed = self._dseed
            # counters of the teacher stream's start gate (csrc/encoder.hip): it switches itself off after 3 consecutive timeouts
            self.gate = O.gate_stats_new(self.dev)
        self._ss_zeroed = False     # the prologue launch of the running step zeroed the optimizer's gradient-norm accumulator
        self._grad_clean = False    # the previous step's AdamW launch left the gradient buffer zeroed (no fill launch needed)

    def gate_report(self):
        """what the teacher stream's start gate did so far: calls / opened / already_resident / timeouts / disabled / skipped (synchronises)"""
        return O.gate_report(self.gate) if self.on_gpu else None

    def check_health(self):
        """raise if the run so far computed anything it should not trust: a row-split encoder launch whose in-launch hand-off gave up
        (csrc/encoder.hip).  Synchronises: call it where a loss is read anyway (logging, validation, checkpoint)."""
        if self.on_gpu:
            O.check_encoder_health(self.dev)
        h = {"skipped_optimizer_steps": self.opt.skipped_steps() if self.on_gpu else 0, "card_shared_with_other_ranks": self.sync.card_shared,
             "row_split_encoder_launches": bool(O.ENC_ROW_SPLIT)}
        h.update((self.opt.grad_norm_report() if self.on_gpu else None) or {})        # grad_norm / clip_factor of the last optimizer step
        return h

    def gate_reset(self):
        """re-arm a gate that switched itself off (e.g. after a profiler run that serialised the streams)"""
        if self.on_gpu:
            self.gate.zero_()

    def _one_update_per_call(self, what):
        """the teacher-ahead and captured forms run one optimizer update per call / replay: they never accumulate, so the 1 / accum_steps
        of `_opt_step` would silently shrink their gradients"""
        if self.accum_steps != 1:
            raise NotImplementedError(f"{what} runs one optimizer step per call: gradient accumulation (accum_steps={self.accum_steps}) "
                                      "is served by step() only")

    def _graph_ctx(self, g):
        # relaxed: helper threads launch into the capture (lib.lockstep).  Stream priorities were tried and rejected: a
        # high-priority capture stream for the student chain (teacher at normal priority) made replays 1.7x SLOWER.
        return _capture(g, capture_error_mode="relaxed")

    def mkrw(self):
        """MKRW ability weights softmax(randn(5)/rw_temp)*5 (map_nav_src/r2r/agent.py:866-871) AND the step's dropout seed, drawn ON the
        device by one launch, so a replayed graph sees fresh values every step.  Returns the weights (a fixed device buffer)."""
        if not self.on_gpu:
            return torch.softmax(torch.randn(5, dtype=torch.float32) / self.rw_temp, dim=-1) * 5
        g, b, n = self.loss_scale_rule
        O.step_rng(self.seed, self._rng_counter, self.rw_temp, seed_out=self._dseed, rw_out=self._rw, zero_me=self.opt.ss,
                   scale_state=self.opt.loss_scale, growth=g, backoff=b, interval=n)
        self._ss_zeroed = True
        return self._rw

    def _zero_grad(self):
        """gradient accumulators of the step that begins: already zero when the previous step of THIS trainer ended in its AdamW launch
        (which zeroes what it consumed); anything else that wrote gradients in between must call store.zero_grad() itself"""
        if self._grad_clean:
            self._grad_clean = False
            return
        self.student.store.zero_grad()

    # ---- gradient exchange from inside the backward ---------------------------------------------------------------------
    def _touched_rows(self, task, plan):
        """device ids of the word-embedding rows with gradient, or None when the table's gradient is dense (mlm: tied decoder)"""
        if task == "mlm" or self.sync.sparse_cap is None:
            return None
        rows = plan.get("emb_rows")
        return rows if rows is not None and rows.numel() > 0 else None      # bucket-padded plans hold an empty placeholder: dense

    def _window_touched(self):
        """the rows with gradient in the flat buffer at the end of an accumulation window: the UNION over its micro-batches (the buffer
        accumulates every micro-batch's rows; exchanging only the last one's would leave the others unsummed across ranks), or None
        (dense) as soon as one micro-batch touched the table densely"""
        if any(r is None for r in self._window_rows):
            return None
        return self._window_rows[0] if len(self._window_rows) == 1 else torch.unique(torch.cat(self._window_rows))

    def _bucket_hook(self, task, plan):
        """on_bucket callback for model.backward(): launches each bucket's exchange on the side stream as soon as the explicit
        backward has finished that bucket's gradients (bucket 0 runs under the text / panorama backward)"""
        if self.sync.world == 1 or not self.sync.overlap:
            return None
        return lambda i, ctx: self.sync.reduce_bucket(i, self._window_touched() if i == 2 else None, cap_scale=self.accum_steps)

    # ---- the pieces ------------------------------------------------------------------------------------
    def _fwd_bwd(self, batch, task, rw, plan):
        st, te = self.student, self.teacher
        t_out, inputs = None, None
        if te is not None:
            inputs = st._inputs(batch, plan)             # cast once, shared by teacher and student
            if self.side is not None:
                main = torch.cuda.current_stream()
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side), torch.no_grad():
                    t_res = te(batch, task, compute_loss=False, return_outputs=True, plan=plan, inputs=inputs)

                def t_out():                              # joined lazily, right before the distillation losses
                    torch.cuda.current_stream().wait_stream(self.side)
                    return t_res
            else:
                with torch.no_grad():
                    t_out = te(batch, task, compute_loss=False, return_outputs=True, plan=plan, inputs=inputs)
        drawn = self.mkrw()
        if rw is None and te is not None:
            rw = drawn
        if self._micro == 0:
            self._zero_grad()
            self._window_rows = []
        self._window_rows.append(self._touched_rows(task, plan))
        out = st(batch, task, compute_loss=True, teacher_outputs=t_out, rw=rw, plan=plan, inputs=inputs)
        last_micro = self._micro == self.accum_steps - 1
        hook = self._bucket_hook(task, plan) if (last_micro and not torch.cuda.is_current_stream_capturing()) else None
        st.backward(on_bucket=hook)
        self._exchanged = hook is not None
        return out

    # ---- teacher one batch ahead ---------------------------------------------------------------------------
    # The frozen teacher's forward does not depend on the student's update, so the teacher can work on batch i+1 while the
    # student trains on batch i (the reference's PrefetchLoader already holds the next batch, data/loader.py:78-120).  On
    # the side stream its ~75 small launches then fill the gaps of the WHOLE student step (forward + backward, ~3 ms)
    # instead of competing with the student's forward only.  Every step still runs exactly one teacher forward and one
    # student update; only the order of independent work changes.
    def teacher_forward(self, batch, task, plan):
        """teacher outputs for one batch (also carries the cast inputs, reused by the student's step on that batch)"""
        with torch.no_grad():
            inputs = self.student._inputs(batch, plan)
            return self.teacher(batch, task, compute_loss=False, return_outputs=True, plan=plan, inputs=inputs)

    def _fwd_bwd_ahead(self, cur, t_cur, nxt, rw=None):
        """student step on cur = (batch, task, plan) against the ready teacher outputs t_cur, while the teacher runs on nxt"""
        st = self.student
        self._one_update_per_call("step_ahead / capture_ahead")
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            t_next = self.teacher_forward(*nxt)
        batch, task, plan = cur
        drawn = self.mkrw()
        if rw is None:
            rw = drawn
        self._zero_grad()
        self._window_rows = [self._touched_rows(task, plan)]
        out = st(batch, task, compute_loss=True, teacher_outputs=t_cur, rw=rw, plan=plan, inputs=t_cur["inputs"])
        hook = self._bucket_hook(task, plan) if not torch.cuda.is_current_stream_capturing() else None
        st.backward(on_bucket=hook)
        self._exchanged = hook is not None
        main.wait_stream(self.side)
        return out, t_next

    def step_ahead(self, cur, t_cur, nxt, rw=None):
        """eager form: returns (student outputs for cur, teacher outputs for nxt -- pass them as t_cur of the next call)"""
        out, t_next = self._fwd_bwd_ahead(cur, t_cur, nxt, rw)
        self._optimize()
        self.global_step += 1
        return out, t_next

    def capture_ahead(self, cur, t_cur, nxt, rw=None, t_next_into=None):
        """HIP-graph form of step_ahead.  t_cur's tensors must stay alive (they are written by the graph captured for the
        previous batch, or by an eager teacher_forward).  t_next_into: optionally copy the teacher outputs for nxt into this
        existing output dict at the end of the graph (closes the ring when a fixed pool of batches is cycled)."""
        full = self.sync.world == 1 and not self.sync.force and not os.environ.get("MAGIC_FORCE_SPLIT_GRAPH")
        for b in (cur[0], nxt[0]):
            off = [k for k, v in b.items() if torch.is_tensor(v) and v.device.type != self.dev.type and k not in ("traj_vp_row", "traj_view_order")]
            if off:
                raise ValueError(f"capture_ahead() needs both batches resident on {self.dev}; host tensors: {off[:4]}...")
        g = torch.cuda.CUDAGraph()
        with self._graph_ctx(g):
            out, t_next = self._fwd_bwd_ahead(cur, t_cur, nxt, rw)
            if t_next_into is not None:
                copy_teacher_outputs(t_next, t_next_into)
                t_next = t_next_into
            if full:
                self._optimize()
        cs = CapturedStep(g, out, cur[2]["traj_steps"], full, keep=(cur, t_cur, nxt, rw))
        cs.t_next = t_next
        return cs

    def capture_student(self, cur, t_cur, rw=None, keep=None, rccl_in_graph=True, touched_static=None):
        """the student's step on `cur` against the teacher outputs `t_cur` (static buffers a teacher graph fills): one graph holding the
        whole step on one GPU; under data parallelism three graphs cut where the gradient buckets are final + the optimizer's graph,
        replayed by `replay_student` with the RCCL calls between them"""
        self._one_update_per_call("capture_student / capture_split")
        full = self.sync.world == 1 and not self.sync.force and not os.environ.get("MAGIC_FORCE_SPLIT_GRAPH")
        batch, task, plan = cur
        two = (not full) and self.sync.overlap and not os.environ.get("MAGIC_DDP_ONE_GRAPH")
        # round 6: the bucket collectives INSIDE the step's graph (RCCL is graph-capturable): one graph, no cuts, no eager launches between replays -- the
        # exchange stream forks off the capturing stream at every bucket boundary and joins it in front of the optimizer's launches.  For steps whose
        # touched word-embedding rows are fixed at capture time (resident batches, or a dense table); streamed batches, whose row ids change per replay,
        # keep the cut-graph form below (`rccl_in_graph=False`).  MAGIC_DDP_GRAPH_RCCL=0 selects the cut-graph form everywhere.
        if two and rccl_in_graph and os.environ.get("MAGIC_DDP_GRAPH_RCCL", "1") != "0" and self.sync.stream is not None and self.sync.rccl is not None and self.sync.rccl.graph_ok:
            gS = torch.cuda.CUDAGraph()
            # touched_static: a -1-padded id buffer the caller refills before every replay (streamed batches: GradSync._sparse_rows_padded); else the
            # resident batch's own rows, fixed for the life of the graph; None / mlm: the dense table
            padded = touched_static is not None and task != "mlm" and bool(self.sync.sparse_cap)
            touched = touched_static if padded else self._touched_rows(task, plan)
            with self._graph_ctx(gS):
                drawn = self.mkrw()
                rw_ = drawn if rw is None else rw
                self._zero_grad()
                out = self.student(batch, task, compute_loss=True, teacher_outputs=t_cur, rw=rw_, plan=plan, inputs=t_cur["inputs"])
                self.student.backward(on_bucket=lambda i, ctx: self.sync.reduce_bucket(i, touched if i == 2 else None, padded=padded))
                self._opt_step(self.sync.finish())
            cs = CapturedStep(gS, out, plan["traj_steps"], True, keep=keep if keep is not None else (cur, t_cur, rw, touched))
            cs.graph2 = cs.graph3 = cs.graph_opt = None
            cs.touched, cs.rccl_in_graph = touched, True
            return cs
        gS = torch.cuda.CUDAGraph()
        gS2 = None
        with self._graph_ctx(gS):
            drawn = self.mkrw()
            rw_ = drawn if rw is None else rw
            self._zero_grad()
            out = self.student(batch, task, compute_loss=True, teacher_outputs=t_cur, rw=rw_, plan=plan, inputs=t_cur["inputs"])
            if two:
                # data parallel: the student's step is TWO graphs cut where gradient bucket 0 (heads + cross-modal encoders) is
                # final, so its exchange -- issued eagerly between the two replays, on the exchange stream -- runs under the second
                # graph (text / panorama backward).  RCCL calls stay outside the captured graphs.
                self.student.backward_phase1()
                O.flush_dw(keep_active=True)
            else:
                self.student.backward()
            if full:
                self._optimize()
        gS3 = None
        if two:
            # ... and the text / panorama half is cut once more where the middle bucket is final (two rounds in): graph 2 | graph 3
            gS2, gS3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            state = {"ctx": _capture(gS2, pool=gS.pool(), capture_error_mode="relaxed")}
            state["ctx"].__enter__()

            def cut():
                O.flush_dw(keep_active=True)
                state["ctx"].__exit__(None, None, None)
                state["ctx"] = _capture(gS3, pool=gS.pool(), capture_error_mode="relaxed")
                state["ctx"].__enter__()
            try:
                self.student.backward_phase2(on_cut=cut)
            finally:
                state["ctx"].__exit__(None, None, None)
        cs = CapturedStep(gS, out, plan["traj_steps"], full, keep=keep if keep is not None else (cur, t_cur, rw))
        cs.graph2, cs.graph3, cs.touched = gS2, gS3, self._touched_rows(task, plan)
        cs.graph_opt = None
        if two:            # the optimizer's launches as a graph of their own, replayed once the exchange has landed (1 / world is a constant)
            cs.graph_opt = torch.cuda.CUDAGraph()
            with _capture(cs.graph_opt, pool=gS.pool(), capture_error_mode="relaxed"):
                self._opt_step(1.0 / self.sync.world if self.sync.world > 1 else 1.0)
        return cs

    def replay_student(self, cs, touched=None, between=None):
        """replay a `capture_student` step on the current stream.  touched: this batch's word-embedding row ids (streamed batches: they
        change per replay; default: the captured batch's); between(): called after the last backward graph, before the exchange is
        awaited (replay_split launches the next teacher graph there)"""
        O.dw_guard()                                   # the graphs hold deterministic weight-gradient launches (shared workspace: ops.dw_guard)
        if getattr(cs, "rccl_in_graph", False) and touched is not None and touched is not cs.touched:
            raise ValueError("this step was captured with its bucket collectives (and its touched word-embedding rows) inside the graph: capture it with "
                             "rccl_in_graph=False to pass per-replay row ids")
        cs.graph.replay()
        if getattr(cs, "graph2", None) is not None:
            self.sync.reduce_bucket(0)                 # exchange stream: after graph 1, under graphs 2 and 3
            cs.graph2.replay()
            self.sync.reduce_bucket(1)                 # after graph 2, under graph 3
            cs.graph3.replay()
            self.sync.reduce_bucket(2, touched if touched is not None else cs.touched)
            self._exchanged = True
        if between is not None:
            between()
        if not cs.full:
            if getattr(cs, "graph_opt", None) is not None and self._exchanged:
                self.sync.finish()
                self._exchanged = False
                cs.graph_opt.replay()
                self.opt.t += 1
            else:
                self._optimize()
        self.global_step += 1
        return cs.out

    def capture_split(self, cur, t_cur, nxt, rw=None, t_next_into=None):
        """Same work as capture_ahead as TWO graphs: the student step on `cur` (main stream) and the teacher forward on `nxt`
        (side stream), replayed concurrently by `replay_split` without a per-step fork/join inside one graph."""
        cs = self.capture_student(cur, t_cur, rw=rw, keep=(cur, t_cur, nxt, rw))
        gT = torch.cuda.CUDAGraph()
        with _capture(gT, stream=self.side, capture_error_mode="relaxed"):
            if self.student.will_fuse_encoders(cur[2]):
                O.encoder_start_gate(self.gate)   # the student's whole-encoder launch (on `cur`) first: csrc/encoder.hip magic_encoder_start_gate
            t_next = self.teacher_forward(*nxt)
            if t_next_into is not None:
                copy_teacher_outputs(t_next, t_next_into)
                t_next = t_next_into
        cs.t_graph, cs.t_next = gT, t_next
        return cs

    def replay_split(self, cs):
        """teacher graph for the NEXT batch on the side stream (after the previous student step, whose buffers it may recycle),
        student graph on the main stream (after the teacher graph that produced ITS teacher outputs, one replay earlier)"""
        main = torch.cuda.current_stream()
        if getattr(self, "_t_done", None) is not None:
            main.wait_event(self._t_done)
        self.side.wait_stream(main)

        def teacher_next():
            with torch.cuda.stream(self.side):
                cs.t_graph.replay()
                self._t_done = torch.cuda.Event()
                self._t_done.record(self.side)
        return self.replay_student(cs, between=teacher_next)

    def _optimize(self):
        if getattr(self, "_exchanged", False):       # the buckets went out from inside the backward: only wait for them
            gscale = self.sync.finish()
            self._exchanged = False
        else:
            gscale = self.sync.all_reduce()
        self._opt_step(gscale)

    def _opt_step(self, gscale):
        # fp16: the buffer holds grad_scale x the gradient; accumulation: the sum over accum_steps micro-batches
        static = 1.0 if self.opt.loss_scale is not None else float(getattr(self.student, "grad_scale", 1.0))      # (dynamic: the kernel reads 1 / S)
        self.opt.step(gscale=gscale / (static * self.accum_steps),
                      ss_zeroed=self._ss_zeroed, zero_grad=True)
        self._ss_zeroed, self._grad_clean = False, True

    def step(self, batch, task, rw=None, plan=None):
        """one micro-batch: forward + backward; on the last micro-batch of an accumulation window (every call when accum_steps == 1) also
        the gradient exchange, clip and AdamW update.  Returns the model's outputs of this micro-batch."""
        plan = plan if plan is not None else build_plan(batch, task, self.dev)
        out = self._fwd_bwd(batch, task, rw, plan)
        if self._micro == self.accum_steps - 1:
            self._optimize()
            self.global_step += 1
            self._micro = 0
        else:
            self._micro += 1
        return out

    # ---- HIP-graph path --------------------------------------------------------------------------------
    def capture(self, batch, task, plan, rw=None):
        """Capture one step for this (resident) batch.  With world_size 1 the optimizer is inside the graph; with
        data parallelism the graph ends after backward and the all-reduce + optimizer run eagerly after replay."""
        off = [k for k, v in batch.items() if torch.is_tensor(v) and v.device.type != self.dev.type and k not in ("traj_vp_row", "traj_view_order")]
        if off:
            raise ValueError(f"capture() needs the batch resident on {self.dev} (synth.batch_to); host tensors: {off[:4]}...")
        full = self.sync.world == 1 and not self.sync.force and not os.environ.get("MAGIC_FORCE_SPLIT_GRAPH")   # (env: exercise the DP split on 1 GPU)
        if self.accum_steps != 1:
            raise NotImplementedError("captured steps run one optimizer step per replay: use step() with accum_steps > 1")
        g = torch.cuda.CUDAGraph()
        with self._graph_ctx(g):
            out = self._fwd_bwd(batch, task, rw, plan)
            if full:
                self._optimize()
        return CapturedStep(g, out, plan["traj_steps"], full, keep=(batch, plan, rw))

    def replay(self, cs):
        O.dw_guard()
        cs.graph.replay()
        if not cs.full:
            self._optimize()
        self.global_step += 1
        return cs.out
