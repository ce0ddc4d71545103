"""Gradient lanes: independent parts of ONE autograd pass on separate HIP streams.

The navigator iteration (agent_base.py:243-250) is two rollouts over the same weights whose step chains are latency-bound (a few hundred rows
per launch: ~120 of 256 CUs busy, 5 us of dispatch latency between dependent launches).  Their forward and backward chains are independent
except for ONE thing: both add into the same parameter gradients.  A lane gives a rollout

  * its own stream (lane 0: the ambient stream),
  * its own flat gradient buffer (params.ParamStore: `g()` / the `Lin` / `LN` handles resolve to the current lane's buffer),
  * its own workspace + counters of the deterministic weight-gradient launch (ops.dw_counters),

so the two chains can overlap on the GPU with no read-modify-write on shared memory.  The buffers are summed once, in lane order, when the
backward pass ends (`ParamStore.merge_lanes`, from the end-of-backward callback): the result is the sum the single-stream pass computes, in a
fixed order.

`cur` is the lane the calling code is working for.  Forward code runs under `use(k, stream)`; every autograd Function that writes parameter
gradients records `cur` in its forward and re-enters it in its backward (the autograd engine itself runs a node's backward on the stream its
forward ran on).  One thread at a time: the loop's thread in the forward, the engine's device thread during `backward()`.
"""
import contextlib
import os

import torch

cur = 0
_streams = {}          # (device index, lane) -> torch.cuda.Stream
_used = {}             # device index -> set of lane streams that have work queued since the last join


LANE_PROBE = os.environ.get("MAGIC_LANE_PROBE", "0") == "1"
PROBE_US = 300         # the probe's waiter gives up after this long (a release on another queue arrives within ~10 us)
probe_log = []         # one record per `beside` call (bench.py prints them)


def runs_beside(a, b):
    """do launches on streams a and b run side by side (True) or in order (False)?  The runtime deals streams onto a few hardware queues
    (GPU_MAX_HW_QUEUES, default 4) and two streams on one queue serialise -- csrc/encoder.hip magic_stream_probe.  Drains the device."""
    from . import lib as L
    dev = a.device
    torch.cuda.synchronize(dev)
    w = torch.zeros(2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    with torch.cuda.device(dev):
        L.call("magic_stream_probe", L.P(w), PROBE_US, a.cuda_stream, b.cuda_stream)
    torch.cuda.synchronize(dev)
    return int(w[1].item()) == 1


def beside(others, device=None, priority=None, tries=16):
    """a new stream that is PROVEN to run beside every stream in `others` (the teacher's stream beside the student's, a rollout lane beside
    lane 0, the gradient exchange beside both): torch hands out pool streams round-robin and the runtime deals them onto its hardware
    queues round-robin, so whether two given streams overlap is otherwise an accident of how many streams were made before them.  Takes
    candidates from torch's pool until one passes `runs_beside` against all of `others`; if none does in `tries` (one hardware queue, a
    tool serialising the device) the last candidate is returned and the caller's work runs in order, as it would have.  Outside a capture."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    mk = (lambda: torch.cuda.Stream(device=dev, priority=int(priority))) if priority is not None else (lambda: torch.cuda.Stream(device=dev))
    if torch.cuda.is_current_stream_capturing():
        return mk()
    seen, s = set(), None
    for n in range(1, tries + 1):
        s = mk()
        if s.cuda_stream in seen:            # the pool wrapped around
            break
        seen.add(s.cuda_stream)
        if all(o is not None and s.cuda_stream != o.cuda_stream and runs_beside(s, o) for o in others):
            probe_log.append(dict(tried=n, found=True))
            return s
    probe_log.append(dict(tried=len(seen), found=False))
    return s


def stream(device, k):
    """lane k's stream on `device` (k >= 1; lane 0 is whatever stream is current).  Picked to run beside the stream that is current when the
    lane is first asked for (lane 0's) and beside the lower lanes"""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(k))
    s = _streams.get(key)
    if s is None:
        d = torch.device("cuda", key[0])
        if LANE_PROBE:
            lower = [v for (i, kk), v in _streams.items() if i == key[0] and kk < key[1]]
            s = _streams[key] = beside([torch.cuda.current_stream(d)] + lower, device=d)
        else:
            s = _streams[key] = torch.cuda.Stream(device=d)
    return s


@contextlib.contextmanager
def use(k, s=None):
    """work for lane k (on stream s when given)"""
    global cur
    old = cur
    cur = int(k)
    try:
        if s is not None:
            _used.setdefault(s.device.index, set()).add(s)
            with torch.cuda.stream(s):
                yield
        else:
            yield
    finally:
        cur = old


def fork(device, ks):
    """before the lanes start: their streams wait for everything queued on the current stream"""
    now = torch.cuda.current_stream(device)
    for k in ks:
        if k:
            stream(device, k).wait_stream(now)


def pending(device=None):
    idx = torch.device(device).index if device is not None else None
    return any(v for d, v in _used.items() if idx is None or d == idx)


def join(device=None, forget=True):
    """the current stream waits for every lane stream with queued work (no-op inside a capture / without lanes)"""
    if not _used or torch.cuda.is_current_stream_capturing():
        return
    now = torch.cuda.current_stream(device)
    ss = _used.get(now.device.index)
    if ss:
        for s in ss:
            if s != now:
                now.wait_stream(s)
        if forget:
            ss.clear()


def fence(device=None):
    """two-way: the current stream waits for the lanes, then the lanes wait for the current stream -- around a launch on the current stream
    that reads tensors the lanes produced and frees them afterwards (the eager weight-gradient flush in the middle of a backward pass)"""
    if not _used or torch.cuda.is_current_stream_capturing():
        return None
    now = torch.cuda.current_stream(device)
    ss = [s for s in _used.get(now.device.index, ()) if s != now]
    for s in ss:
        now.wait_stream(s)
    return now, ss


def fence_end(tok):
    if tok:
        now, ss = tok
        for s in ss:
            s.wait_stream(now)


def lane_bwd(fn):
    """backward of an autograd Function that writes parameter gradients: re-enter the gradient lane its forward ran in (the forward records
    `ctx.lane = lanes.cur`), so its `store.g(...)` handles resolve to that lane's buffer whatever lane the autograd engine happens to run the node from"""
    def run(ctx, *grads):
        k = getattr(ctx, "lane", 0)
        if k == cur:
            return fn(ctx, *grads)
        with use(k):
            return fn(ctx, *grads)
    return run
