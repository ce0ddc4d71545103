"""HIP-graph replay for STREAMED batches.

`PretrainStep.capture` replays a graph on the batch it was captured with.  A streamed batch is new every step and ragged; `StreamStep` makes it
replayable: the loader pads every batch to its shape bucket (host/bucket.py) and ships it as ONE packed record whose layout depends on the
bucket only (host/loader.pack); per (task, layout) there is ONE static device buffer, ONE captured graph over the views into it (teacher forward on
the side stream + student forward / MAKD / backward / clip / AdamW), and a handful of device-side scalars (plan["dyn"]) with the batch's TRUE
extents and normalisers.  A step is then: one H2D copy of the record into the static buffer, one small copy of the scalars, one graph launch.

The first batch of a bucket pays the capture (~0.2 s); a run sees a few dozen buckets (bucket.bucket_of: L fixed at the truncation length, K in
steps of 8, sum T in steps of 32, masked tokens in steps of 64).
"""
import pickle

import numpy as np
import torch

from .lib import capture as _capture
from .loader import unpack
from .model_pretrain import DYN_TERMS
from .plan import check_plan


class _Entry:
    pass


class StreamStep:
    def __init__(self, trainer, feature_table=None, rw=None, max_graphs=96):
        """rw: fixed MKRW ability weights (a device tensor) instead of a fresh draw inside every replay (tests); max_graphs: (task, bucket,
        slot) entries kept -- each holds its graphs' private memory pools (~0.5-1 GB at B = 48); the least recently used one is dropped"""
        self.tr, self.dev, self.ftab, self.rw, self.max_graphs = trainer, trainer.dev, feature_table, rw, max_graphs
        self.cache = {}
        self.captures = 0
        self.capture_s = 0.0          # host wall time spent capturing bucket graphs (the GPU idles meanwhile): benches report it apart
        # data parallel (trainer.sync.world > 1): every rank streams its OWN records -- ranks may sit in different buckets in the same step, the
        # exchange only needs the flat gradient buffer, whose layout is the model's -- and the RCCL calls stay outside the graphs: `run`
        # replays trainer.capture_student's three backward-cut graphs with the bucket exchanges between them, `step` one graph up to the end of
        # the backward followed by the monolithic exchange (trainer.capture / replay)

    def _key(self, task, manifest):
        return (task,) + tuple((k, dt, shape, o) for k, dt, shape, o, _ in manifest)

    def _touch(self, key, e):
        """least-recently-used bookkeeping (dicts keep insertion order)"""
        self.cache.pop(key, None)
        self.cache[key] = e
        while len(self.cache) > self.max_graphs:
            old = next(iter(self.cache))
            if old == key:
                break
            torch.cuda.synchronize()              # its graphs may still be in flight
            del self.cache[old]

    def _capture(self, key, task, rec, parsed):
        e = _Entry()
        n = int(rec["buf"].numel())
        e.dbuf = torch.empty(n, dtype=torch.uint8, device=self.dev)
        e.batch, e.plan = unpack(rec, self.dev, dbuf=e.dbuf, parsed=parsed)
        nt = len(DYN_TERMS)
        # pinned staging for the per-batch scalars: a ring, because the host may run several steps ahead of the copies it has queued
        e.ring = [(torch.zeros(2 * nt, dtype=torch.int32).pin_memory(), torch.zeros(nt, dtype=torch.float32).pin_memory(), torch.cuda.Event())
                  for _ in range(4)]
        e.turn, e.keep = 0, [None] * 4
        e.plan["dyn"] = dict(i=torch.zeros(2 * nt, dtype=torch.int32, device=self.dev), f=torch.zeros(nt, dtype=torch.float32, device=self.dev), fill={})
        if self.ftab is not None:
            e.batch["view_table"] = self.ftab
        torch.cuda.synchronize()
        e.cs = self.tr.capture(e.batch, task, e.plan, rw=self.rw)
        e.fill = [(DYN_TERMS.index(t), fn) for t, fn in e.plan["dyn"]["fill"].items()]
        self.captures += 1
        return e

    def step(self, task, rec):
        """one training step on a packed, bucket-padded record (loader.pack_bucketed); returns the step's output dict (device tensors of the
        bucket's static buffers: read them before the next step of the same bucket)"""
        parsed = pickle.loads(rec["blob"])
        manifest, meta = parsed
        if "true" not in meta:
            raise ValueError("StreamStep needs bucket-padded records (PlanCollate(..., bucket={...}) / loader.pack_bucketed)")
        check_plan(dict(limits=meta["limits"], L=meta["L"], V=meta["V"]), self.tr.student.config)
        key = self._key(task, manifest)
        e = self.cache.get(key)
        if e is None:
            e = self._capture(key, task, rec, parsed)          # (copies this record into the new static buffer as well)
        else:
            buf = rec["buf"]
            if not buf.is_pinned():
                buf = buf.pin_memory()
            e.dbuf.copy_(buf, non_blocking=True)
            e.keep[e.turn] = buf                              # the pinned source must outlive the asynchronous copy
        self._touch(key, e)
        true = meta["true"]
        host_i, host_f, ev = e.ring[e.turn]
        ev.synchronize()                                      # (no-op unless the host is four steps of this bucket ahead)
        for i, fn in e.fill:
            vo, vi, norm = fn(true)
            host_i[2 * i], host_i[2 * i + 1], host_f[i] = int(vo), int(vi), float(norm)
        e.plan["dyn"]["i"].copy_(host_i, non_blocking=True)
        e.plan["dyn"]["f"].copy_(host_f, non_blocking=True)
        ev.record()
        e.turn = (e.turn + 1) % len(e.ring)
        out = self.tr.replay(e.cs)
        e.traj_steps = meta["traj_steps"]
        return out, meta

    # ---- teacher one batch ahead (the resident-batch headline's schedule) -------------------------------------------------------------
    # Per (task, bucket, slot) -- slot = step parity, so that two consecutive batches of one bucket never share buffers -- TWO graphs over the
    # slot's static record buffer: T = the frozen teacher's forward (side stream), S = the student's step against T's static outputs (main
    # stream).  While S_i trains on batch i, T_{i+1} already runs on batch i+1, which needs the NEXT record one step early: `run` drives
    # an iterable of (task, record) and yields one (out, meta) per step.
    def _capture_split(self, key, task, rec, parsed):
        import time
        tr = self.tr
        e = _Entry()
        e.dbuf = torch.empty(int(rec["buf"].numel()), dtype=torch.uint8, device=self.dev)
        e.batch, e.plan = unpack(rec, self.dev, dbuf=e.dbuf, parsed=parsed)
        nt = len(DYN_TERMS)
        e.ring = [(torch.zeros(2 * nt, dtype=torch.int32).pin_memory(), torch.zeros(nt, dtype=torch.float32).pin_memory(), torch.cuda.Event())
                  for _ in range(4)]
        e.turn, e.keep = 0, [None] * 4
        e.plan["dyn"] = dict(i=torch.zeros(2 * nt, dtype=torch.int32, device=self.dev), f=torch.zeros(nt, dtype=torch.float32, device=self.dev), fill={})
        if self.ftab is not None:
            e.batch["view_table"] = self.ftab
        torch.cuda.synchronize()
        t_cap = time.perf_counter()          # (after the drain: the steps the host had queued ahead are training time, not capture time)
        e.gT = torch.cuda.CUDAGraph()
        with _capture(e.gT, stream=tr.side, capture_error_mode="relaxed"):
            if tr.student.will_fuse_encoders(e.plan):      # (every bucket of one stream pads to the same L, V: S_i fuses iff this plan does)
                from . import ops as O
                O.encoder_start_gate(tr.gate)   # T_{i+1} starts once S_i's whole-encoder launch has its workgroups resident (trainer.capture_split)
            e.t_out = tr.teacher_forward(e.batch, task, e.plan)
        # data parallel: ONE graph with the bucket collectives inside when the exchange runs on the direct RCCL communicator (trainer.capture_student) --
        # the touched word-embedding rows change per replay, so the graph reads them from a static -1-padded buffer `_stage` refills; else three
        # graphs cut at the bucket boundaries + the optimizer's, the collectives between the replays
        e.ids_buf = None
        if (tr.sync.world > 1 or tr.sync.force) and tr.sync.rccl is not None and tr.sync.rccl.graph_ok and task != "mlm" and tr.sync.sparse_cap:
            e.ids_buf = torch.full((int(tr.sync.sparse_cap),), -1, dtype=torch.int64, device=self.dev)
            e.ids_host = [torch.full((int(tr.sync.sparse_cap),), -1, dtype=torch.int64).pin_memory() for _ in range(4)]
        e.cs = tr.capture_student((e.batch, task, e.plan), e.t_out, rw=self.rw, rccl_in_graph=e.ids_buf is not None or task == "mlm", touched_static=e.ids_buf)
        e.out = e.cs.out
        e.fill = [(DYN_TERMS.index(t), fn) for t, fn in e.plan["dyn"]["fill"].items()]
        e.t_done, e.loaded = torch.cuda.Event(), torch.cuda.Event()
        self.captures += 1
        self.capture_s += time.perf_counter() - t_cap
        return e

    def _stage(self, item, slot):
        """record -> its (bucket, slot) entry: copy + per-batch scalars on the main stream, then the teacher's forward on the side stream"""
        task, rec = item
        parsed = pickle.loads(rec["blob"])
        manifest, meta = parsed
        if "true" not in meta:
            raise ValueError("StreamStep needs bucket-padded records (PlanCollate(..., bucket={...}) / loader.pack_bucketed)")
        check_plan(dict(limits=meta["limits"], L=meta["L"], V=meta["V"]), self.tr.student.config)
        key = self._key(task, manifest) + (("slot", slot),)
        e = self.cache.get(key)
        if e is None:
            e = self._capture_split(key, task, rec, parsed)
        else:
            buf = rec["buf"]
            if not buf.is_pinned():
                buf = buf.pin_memory()
            e.dbuf.copy_(buf, non_blocking=True)
            e.keep[e.turn] = buf
        self._touch(key, e)
        host_i, host_f, ev = e.ring[e.turn]
        ev.synchronize()
        for i, fn in e.fill:
            vo, vi, norm = fn(meta["true"])
            host_i[2 * i], host_i[2 * i + 1], host_f[i] = int(vo), int(vi), float(norm)
        if getattr(e, "ids_buf", None) is not None:
            # the graph reads this batch's rows from the entry's static buffer (pads = -1); no ids in the record: every slot a pad -> nothing exchanged
            # sparsely would be WRONG, so such a record is refused
            ids = meta.get("touched_ids")
            if ids is None:
                raise ValueError("StreamStep (data parallel, collectives inside the graph): records must carry `touched_ids` (loader.pack_bucketed)")
            host = e.ids_host[e.turn]                 # (this ring slot's previous copies have completed: ev.synchronize() above)
            k = len(ids)
            if k > host.numel():
                raise ValueError(f"sparse embedding exchange: {k} touched rows > cap {host.numel()}")
            host[:k] = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64))
            host[k:] = -1
            e.ids_buf.copy_(host, non_blocking=True)
        e.plan["dyn"]["i"].copy_(host_i, non_blocking=True)
        e.plan["dyn"]["f"].copy_(host_f, non_blocking=True)
        ev.record()
        e.turn = (e.turn + 1) % len(e.ring)
        e.touched = None
        if getattr(e, "ids_buf", None) is not None:
            pass                                   # (refilled above, under the ring slot's event)
        elif self.tr.sync.world > 1 and task != "mlm" and self.tr.sync.sparse_cap and meta.get("touched_ids") is not None:
            # the word-embedding rows THIS batch read (the record's own ids, not the captured batch's): trainer.GradSync exchanges those rows
            ids = torch.from_numpy(np.ascontiguousarray(meta["touched_ids"], dtype=np.int64)).pin_memory()
            e.touched, e.keep_ids = ids.to(self.dev, non_blocking=True), ids
        main = torch.cuda.current_stream()
        e.loaded.record(main)                  # after every earlier student step on the main stream (incl. the last user of this slot's buffers)
        side = self.tr.side
        side.wait_event(e.loaded)
        with torch.cuda.stream(side):
            e.gT.replay()
            e.t_done.record(side)
        return e, meta

    def run(self, feed):
        """generator over (task, record) pairs: yields (out, meta) per training step, teacher one batch ahead on the side stream"""
        if self.tr.side is None or self.tr.teacher is None:
            raise RuntimeError("StreamStep.run needs the trainer's teacher side stream (use step() otherwise)")
        it = iter(feed)
        try:
            cur = self._stage(next(it), 0)
        except StopIteration:
            return
        i = 0
        while cur is not None:
            try:
                nxt = self._stage(next(it), (i + 1) % 2)      # T_{i+1} goes out before S_i: it fills the gaps of the whole student step
            except StopIteration:
                nxt = None
            e, meta = cur
            main = torch.cuda.current_stream()
            main.wait_event(e.t_done)
            self.tr.replay_student(e.cs, touched=e.touched)
            yield e.out, meta
            cur = nxt
            i += 1
