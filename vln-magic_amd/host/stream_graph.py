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

from .loader import unpack
from .model_pretrain import DYN_TERMS
from .plan import check_plan


class _Entry:
    pass


class StreamStep:
    def __init__(self, trainer, feature_table=None, rw=None):
        """rw: fixed MKRW ability weights (a device tensor) instead of a fresh draw inside every replay (tests)"""
        self.tr, self.dev, self.ftab, self.rw = trainer, trainer.dev, feature_table, rw
        self.cache = {}
        self.captures = 0
        if trainer.sync.world != 1:
            raise NotImplementedError("StreamStep: single-GPU graphs only (the data-parallel exchange runs between graph halves, trainer.capture_split)")

    def _key(self, task, manifest):
        return (task,) + tuple((k, dt, shape, o) for k, dt, shape, o, _ in manifest)

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
        self.cache[key] = e
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
