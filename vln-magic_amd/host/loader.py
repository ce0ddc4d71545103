"""Streaming batches into the training step (pretrain_src/data/loader.py).

The reference feeds its step through `DataLoader(num_workers=n_workers, pin_memory, collate_fn=<task>_collate)` ->
`MetaLoader` (task draw + rank-0 broadcast, loader.py:18-75) -> `PrefetchLoader` (one batch ahead, non_blocking H2D,
loader.py:90-124).  `MetaLoader` / `PrefetchLoader` are model-agnostic and keep working in front of these models; what this
module adds is the part specific to the HIP engine:

  * `PlanCollate(collate_fn, task)` -- a collate wrapper for the DataLoader WORKERS: next to the padded batch it builds the
    host half of the engine's index plan (`plan.build_plan_host`: CSR aggregation matrices, fusion map, masks; 6-28 ms of pure
    CPU work per B=48 batch, i.e. several GPU steps -- it must not run in the training process);
  * `DevicePrefetcher(loader, device)` -- PrefetchLoader's role plus the plan: one batch ahead on a copy stream it moves the
    pinned batch (30 MB of fp32 view features at B=48) and the plan (ONE packed copy for its ~60 arrays), and hands
    `(task, batch_on_device, plan_on_device)` to `PretrainStep.step` after a stream wait.
"""
import pickle

import numpy as np
import torch

from .plan import build_plan_host

MODEL_KEYS = ("traj_view_img_fts", "traj_vp_row", "traj_view_order", "traj_loc_fts", "gmap_pos_fts", "gmap_pair_dists", "vp_pos_fts",
              "global_act_labels", "local_act_labels")      # the batch entries the model reads; everything else is in the plan


def pack(batch, hp):
    """(batch, host plan) -> ONE contiguous uint8 tensor + a manifest.  A DataLoader worker hands tensors to the training process
    one shared-memory segment (and one file descriptor) per storage: the ~150 small tensors of a batch + plan cost ~35 ms of IPC
    per batch that way, a single packed storage costs well under 1 ms -- and it is also one pinned copy and one H2D copy."""
    arrays = {f"b/{k}": batch[k] for k in MODEL_KEYS if torch.is_tensor(batch.get(k))}
    for k, v in hp["cpu"].items():
        arrays[f"p/{k}"] = v
    for name, (f, t) in hp["csr"].items():
        for j, a in enumerate(f):
            arrays[f"c/{name}/{j}"] = torch.as_tensor(a)
        for j, a in enumerate(t):
            arrays[f"c/{name}_T/{j}"] = torch.as_tensor(a)
    manifest, off = [], 0
    for k, a in arrays.items():
        off = (off + 15) & ~15
        manifest.append((k, str(a.dtype).replace("torch.", ""), tuple(a.shape), off, a.numel() * a.element_size()))
        off += a.numel() * a.element_size()
    buf = torch.empty(max(off, 16), dtype=torch.uint8)
    for (k, _, _, o, n), a in zip(manifest, arrays.values()):
        if n:
            buf[o:o + n] = a.contiguous().reshape(-1).view(torch.uint8)
    meta = dict(hp["meta"])
    meta["last_rows"] = np.asarray(meta["last_rows"]).tolist()
    # manifest + meta travel as ONE bytes object: the DataLoader's pin-memory thread walks every container of a batch in Python
    # (under the GIL, i.e. on the training thread's time) -- ~1500 small objects per batch as lists / tuples, one leaf as bytes
    return dict(buf=buf, blob=pickle.dumps((manifest, meta), protocol=pickle.HIGHEST_PROTOCOL))


def unpack(rec, device, dbuf=None, parsed=None):
    """packed record -> (batch_on_device, plan_on_device): one (pinned) host buffer, one async copy, views.
    dbuf: an existing device buffer of the record's size to copy into (the static buffer a captured graph reads, stream_graph.StreamStep)"""
    device = torch.device(device)
    buf = rec["buf"]
    manifest, meta = parsed if parsed is not None else pickle.loads(rec["blob"])
    if device.type == "cuda" and not buf.is_pinned():
        buf = buf.pin_memory()
    if dbuf is None:
        dbuf = buf.to(device, non_blocking=True)
    else:
        dbuf.copy_(buf, non_blocking=True)
    batch, plan, csr = {}, {}, {}
    typed = {}                       # one reinterpretation of the whole buffer per dtype; every array is then ONE as_strided view of it
    for k, dt, shape, o, n in manifest:
        dtype = getattr(torch, dt)
        if n:
            tv = typed.get(dt)
            if tv is None:
                sz = torch.empty(0, dtype=dtype).element_size()
                tv = typed[dt] = (dbuf[:dbuf.numel() // sz * sz].view(dtype), sz)
            st, acc = [], 1
            for d in reversed(shape):
                st.append(acc)
                acc *= d
            t = tv[0].as_strided(shape, st[::-1], o // tv[1])
        else:
            t = torch.empty(shape, dtype=dtype, device=device)
        kind, rest = k.split("/", 1)
        if kind == "b":
            batch[rest] = t
        elif kind == "p":
            plan[rest] = t
        else:
            name, j = rest.rsplit("/", 1)
            csr.setdefault(name, {})[int(j)] = t
    for name, parts in csr.items():
        plan[name] = tuple(parts[j] for j in range(len(parts)))
    plan.update(meta)
    plan["last_rows"] = np.asarray(plan["last_rows"], np.int64)
    plan["_stage"], plan["_dbuf"] = buf, dbuf        # every tensor above is a view of dbuf's one allocation
    return batch, plan


class PlanCollate:
    """collate_fn for DataLoader workers: `inputs` -> packed record (`pack`).  `collate_fn` is the task's collate
    (`tasks.py:{mlm,mrc,sap,cfp}_collate`, or `synth.collate` bound to the task).  bucket: keyword arguments of bucket.bucket_of -- the batch is
    then padded to its shape bucket (bucket.pad_batch) so that the record has the bucket's one layout (graph replay, stream_graph.StreamStep)."""

    def __init__(self, collate_fn, task, bucket=None):
        self.collate_fn, self.task, self.bucket = collate_fn, task, bucket

    def __call__(self, inputs):
        batch = self.collate_fn(inputs)
        return pack_bucketed(batch, self.task, self.bucket) if self.bucket is not None else pack(batch, build_plan_host(batch, self.task))


def pack_bucketed(batch, task, bucket_kw=None):
    """collated batch -> packed record of its shape bucket"""
    from .bucket import bucket_of, pad_batch
    bk = bucket_of(batch, task, **(bucket_kw or {}))
    padded, true = pad_batch(batch, task, bk)
    return pack(padded, build_plan_host(padded, task, pad=(bk, true)))


def _to_device(x, device):
    if torch.is_tensor(x):
        return x.to(device, non_blocking=True)
    return x


class DevicePrefetcher:
    """iterate (task, packed record) pairs -- e.g. a MetaLoader over PlanCollate loaders -- one batch ahead"""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def _stage(self, item):
        task, rec = item
        if self.stream is None:
            return (task,) + unpack(rec, self.device)
        with torch.cuda.stream(self.stream):
            b, p = unpack(rec, self.device)
        return task, b, p

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            if self.stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
                cur[2]["_dbuf"].record_stream(torch.cuda.current_stream(self.device))    # one storage holds batch + plan
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            yield cur
