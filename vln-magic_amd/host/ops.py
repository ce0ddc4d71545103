"""Thin tensor-level wrappers over the C ABI (one Python function per kernel family).

Every function launches hand-written HIP through ctypes on torch's current stream; none of them has a
torch/CPU fallback.  Shapes are checked on the host before launch (a faulting kernel can take the
whole GPU node down).  `FLOPS` accumulates the algorithmic dense-contraction FLOPs (2*m*n*k with the
TRUE, unpadded sizes the caller passes via `flop_dims`) for the roofline report.
"""
import ctypes as C
import os

import torch

from . import lanes
from . import lib as L

# algorithmic FLOPs of the dense contractions, per kernel family ("gemm": magic_gemm / magic_gemm_dw_grouped; "linear_ln": the fused
# dense+LayerNorm kernels; "attn": the fused attention kernels) and in total
FLOPS = {"total": 0.0, "gemm": 0.0, "linear_ln": 0.0, "attn": 0.0, "enc": 0.0, "enabled": False}
# algorithmic bytes of the weight-gradient launch(es): every dY and X element read once + the fp32 dW written once (bench.py prints it next
# to the PMC traffic so the re-read factor can be checked from the JSON line)
BYTES = {"dw": 0.0, "dw_launches": 0}


def _count(m, n, k, batch=1, fam="gemm"):
    if FLOPS["enabled"]:
        f = 2.0 * m * n * k * batch
        FLOPS["total"] += f
        FLOPS[fam] += f


def _chk(cond, msg):
    if not cond:
        raise L.MagicHipError("shape check failed: " + msg)


def gemm(layout, A, B, C, M, N, K, lda, ldb, ldc, *, batch=1, nh=1, sA=(0, 0), sB=(0, 0), sC=(0, 0),
         bias=None, epilogue=0, aux=None, ldaux=0, residual=None, ldr=0, C2=None, ldc2=0, alpha=1.0,
         splitk=1, bias_grad=None, accumulate=False, flop_dims=None):
    dt = L.dt(A.dtype)
    _chk(A.dtype == B.dtype, "A/B dtype")
    c_f32 = 1 if C.dtype == torch.float32 else 0
    fd = flop_dims or (M, N, K)
    _count(fd[0], fd[1], fd[2], batch)
    L.call("magic_gemm", dt, layout, batch, nh, M, N, K,
           L.P(A), lda, sA[0], sA[1], L.P(B), ldb, sB[0], sB[1],
           L.P(C), ldc, sC[0], sC[1], c_f32, 1 if accumulate else 0,
           L.P(bias), epilogue, L.P(aux), ldaux, L.P(residual), ldr, L.P(C2), ldc2,
           float(alpha), splitk, L.P(bias_grad), L.stream())


def linear_fwd(x, W, b, M, *, out=None, epilogue=0, residual=None, pre=None, lda=None, ldc=None, ldb=None, K=None,
               flop_rows=None):
    """y[M,N] = act(x[M,K] @ W[N,K]^T + b) (+ residual).  x may be a strided row view (lda)."""
    N = W.shape[0]
    K = K if K is not None else W.shape[1]
    ldb = ldb if ldb is not None else W.stride(0)
    lda = lda if lda is not None else K
    if out is None:
        out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    ldc = ldc if ldc is not None else N
    gemm(0, x, W, out, M, N, K, lda, ldb, ldc, bias=b, epilogue=epilogue,
         residual=residual, ldr=(ldc if residual is not None else 0), C2=pre, ldc2=N,
         flop_dims=(flop_rows if flop_rows is not None else M, N, K))
    return out


def linear_dx(dy, W, M, *, out=None, epilogue=0, aux=None, residual=None, lda=None, ldc=None, ldb=None, N=None, K=None,
              flop_rows=None):
    """dx[M,K] = dy[M,N] @ W[N,K]  (* act'(aux)) (+ residual)"""
    N = N if N is not None else W.shape[0]
    K = K if K is not None else W.shape[1]
    ldb = ldb if ldb is not None else W.stride(0)
    lda = lda if lda is not None else N
    if out is None:
        out = torch.empty(M, K, dtype=dy.dtype, device=dy.device)
    ldc = ldc if ldc is not None else K
    gemm(1, dy, W, out, M, K, N, lda, ldb, ldc, epilogue=epilogue, aux=aux, ldaux=(K if aux is not None else 0),
         residual=residual, ldr=(ldc if residual is not None else 0),
         flop_dims=(flop_rows if flop_rows is not None else M, K, N))
    return out


# measured on the headline step (profiles/micro/splitk_sweep.sh): (target, min_tiles) = (256, 2) 3.34 ms, (128, 8) 3.26 ms, (128, 32) 3.45 ms --
# fewer, longer splits also write 4x fewer fp32 atomic tiles
SPLITK = {"target": int(os.environ.get("MAGIC_SPLITK_TARGET", "192")), "min_tiles": int(os.environ.get("MAGIC_SPLITK_MIN_TILES", "20")),
          "pow2": {"0": "", "up": "up", "down": "down"}[os.environ.get("MAGIC_SPLITK_POW2", "down")]}


def _splitk(tiles, kred):
    """split-K factor of the weight-gradient GEMM: every split block adds a full 64x64 fp32 tile with atomics (16 KB), so
    the atomic traffic is tiles*splitk*16 KB whatever the true dW size -- keep the grid near one block per CU and give
    each block at least `min_tiles` 64-deep k-tiles."""
    if tiles > 128:
        # wide layers (H >= 768): the output alone gives every CU a tile, but a reduction over thousands of rows still wants splitting;
        # measured (profiles/micro/tn_splitk_scan.py, M = 8192): 768x768 141 -> 43 us at 8 splits, 3072x768 170 -> 115 us at 4-8 (then the
        # fp32 atomic traffic of the extra splits takes over)
        return int(max(1, min(kred // 1024, 8 if tiles <= 256 else 4)))
    ks = max(1, (kred + 64 * SPLITK["min_tiles"] - 1) // (64 * SPLITK["min_tiles"]))
    sk = int(max(1, min(SPLITK["target"] // max(tiles, 1), ks, 64)))
    if SPLITK["pow2"] and sk < 8:
        # 1, 2 or 4 splits: csrc/gemm.hip dw_xcd_groups() then gives each split 8 / sk XCDs of its own ("up": 3 -> 4; "down": 3 -> 2)
        lo = 1 << (sk.bit_length() - 1)
        sk = lo if (sk == lo or SPLITK["pow2"] == "down") else 2 * lo
    return sk


SIDE = {"stream": None, "keep": []}     # optional side stream for the weight-gradient GEMMs (off the dX critical chain)


def join_side():
    if SIDE["stream"] is not None:
        torch.cuda.current_stream().wait_stream(SIDE["stream"])
    SIDE["keep"].clear()


DEFER = {"on": not os.environ.get("MAGIC_NO_GROUPED_DW"), "queue": [], "active": False}


def defer_dw(active):
    """Inside a backward pass: queue the weight-gradient GEMMs and launch them ~8 per kernel at `flush_dw()`."""
    DEFER["active"] = bool(active) and DEFER["on"]


def linear_dw(dy, x, dW, db, M, *, N=None, K=None, lda=None, ldb=None, ldc=None, flop_rows=None):
    """dW[N,K] += dy[M,N]^T @ x[M,K] ; db[N] += colsum(dy)   (fp32 atomics, split-K over M).  Nothing on the backward
    chain depends on dW: when deferral is active the problem is queued (inputs kept alive) for a grouped launch."""
    if not DEFER["active"]:
        side = SIDE["stream"]
        if side is None:
            return _linear_dw(dy, x, dW, db, M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, flop_rows=flop_rows)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            _linear_dw(dy, x, dW, db, M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, flop_rows=flop_rows)
        SIDE["keep"].append((dy, x))
        return
    N = N if N is not None else dW.shape[0]
    K = K if K is not None else dW.shape[1]
    lda = lda if lda is not None else N
    ldb = ldb if ldb is not None else K
    ldc = ldc if ldc is not None else dW.stride(0)
    _count(N, K, flop_rows if flop_rows is not None else M)
    tiles = ((N + 63) // 64) * ((K + 63) // 64)
    DEFER["queue"].append((dy, x, dW, db, M, N, K, lda, ldb, ldc, _splitk(tiles, M)))
    DEFER["bytes"] = DEFER.get("bytes", 0) + dy.numel() * dy.element_size() + x.numel() * x.element_size()
    if DEFER["bytes"] > DW_KEEP_BYTES:          # a long autograd pass (navigator rollouts): bound the operands kept alive by the queue
        flush_dw(keep_active=True)


DW_GROUP = int(os.environ.get("MAGIC_DW_GROUP", "96"))      # problems per grouped weight-gradient launch (csrc DW_MAX = 96)


DW_KEEP_BYTES = int(os.environ.get("MAGIC_DW_KEEP_GB", "8")) << 30


# Deterministic weight gradients (csrc/gemm.hip dw_seam): the K-splits of a dW tile -- and the several problems one Linear queues when it is
# called more than once in a step -- are summed in a FIXED order through a workspace instead of fp32 atomics, so two runs of the same step
# give bitwise-identical weight gradients.  MAGIC_DW_ATOMICS=1 restores the atomic form.  The partial slots and the arrival counters
# (zero when allocated, left zero by every launch) are ONE persistent pair per device, baked into every captured graph: two launches
# that use them must never overlap.  `dw_guard` keeps that true across streams: it remembers the stream of the last user (an eager launch,
# or the replay of a graph that holds such launches) and makes a user on a DIFFERENT stream wait for everything queued on that one.
DW_DETERMINISTIC = not os.environ.get("MAGIC_DW_ATOMICS")
DW_WS_MAX_BYTES = int(os.environ.get("MAGIC_DW_WS_MAX_MB", "256")) << 20
DW_COUNTERS = 1 << 17
DW_WS_PERSIST_BYTES = int(os.environ.get("MAGIC_DW_WS_PERSIST_MB", "64")) << 20
_DW_CNT = {}
_DW_WS = {}


def dw_counters(device=None):
    """the arrival counters of the deterministic weight-gradient launch: ONE persistent block per device (zero when allocated, left zero by
    every launch), shared by the eager steps and every captured graph -- weight-gradient launches of one process are stream-ordered (the
    opt-in side stream MAGIC_DW_SIDE takes per-call counters).  Allocated OUTSIDE graph capture (the models call this when they are built):
    memory taken from a capturing graph's pool would be recycled when that graph dies.  None: first use inside a capture."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, lanes.cur)                             # one pair per gradient lane (host/lanes.py): lanes run on different streams
    c = _DW_CNT.get(key)
    if c is None:
        if torch.cuda.is_current_stream_capturing():
            return None                                # this launch takes per-call counters (a memset node)
        c = _DW_CNT[key] = torch.zeros(DW_COUNTERS, dtype=torch.int32, device=torch.device("cuda", idx))
        # ... and ONE partial-slot workspace shared by every launch that fits it (the headline step needs ~35 MB): a workspace per captured
        # graph cost each capture a 57 MB pool allocation -- the streamed feed's in-window captures went from 9 to 34 ms apiece
        _DW_WS[key] = torch.empty(DW_WS_PERSIST_BYTES // 4, dtype=torch.float32, device=torch.device("cuda", idx))
    return c


def _dw_workspace(nf, device):
    key = (device.index if device.index is not None else torch.cuda.current_device(), lanes.cur)
    w = _DW_WS.get(key)
    if w is not None and nf <= w.numel() and SIDE["stream"] is None:
        return w
    return torch.empty(max(nf, 1), dtype=torch.float32, device=device)


_DW_LAST = {}


def dw_guard(device=None):
    """order this stream behind the previous user of the device's shared weight-gradient workspace / counters when that was another stream.
    Called by every eager deterministic launch and in front of every replay of a graph that may hold one (trainer / step_graphs);
    inside a capture it does nothing -- the replay site guards.  Same stream as last time (the normal case): a dict lookup."""
    if torch.cuda.is_current_stream_capturing():
        return
    cur = torch.cuda.current_stream(device)
    key = (cur.device.index, lanes.cur)
    last = _DW_LAST.get(key)
    if last is not None and last != cur:
        cur.wait_stream(last)
    _DW_LAST[key] = cur


# The weight-gradient stream (round 5): captured step instances (host/step_graphs.py) replay their weight-gradient launches -- grouped dW GEMMs +
# the column sums of partial parameter-gradient rows -- on ONE side stream per device, behind an event of the step's backward graph, so the
# chip-filling dW launch runs under the next step's latency-bound chain.  Every read-modify-write of the flat gradient buffer of those steps
# happens on that stream, in launch order (the deterministic seam's workspace and counters are per device: one stream keeps them safe);
# `join_dw_stream` (called wherever the eager queue is flushed: the end of the backward pass) makes the current stream wait for it.
_DW_STREAM = {}
_DW_PENDING = set()


def dw_stream(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    s = _DW_STREAM.get(key)
    if s is None:
        s = _DW_STREAM[key] = torch.cuda.Stream(device=torch.device("cuda", key))
    return s


def dw_stream_used(device):
    _DW_PENDING.add(torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device())


def join_dw_stream():
    if _DW_PENDING and not torch.cuda.is_current_stream_capturing():
        cur = torch.cuda.current_stream()
        for key in list(_DW_PENDING):
            if key == cur.device.index:
                cur.wait_stream(_DW_STREAM[key])
                _DW_PENDING.discard(key)


# round 6: an EARLY flush of the weight gradients queued so far (the heads' and the cross-modal encoders', at the boundary between the two phases of the
# pretraining backward) on the device's weight-gradient stream: the chip-filling grouped launch then runs UNDER the text / panorama stacks' latency-
# bound row-block chain (240 workgroups on 256 CUs at two per CU: half the chip idles there) instead of in one exposed 133 us launch at the end of the
# backward.  The flushed operands (and the partial-row buffers of the column sums) stay alive until `join_dw_early`, which every later flush and the
# embedding backward (whose table-row atomics meet the tied decoder's dW in the word-embedding gradient) call first.
# MEASURED AND REJECTED as the default (profiles/micro/r06_ab_dw_early.txt, same box, this bench): 1.84-1.86 vs 1.43-1.44 ms/step, and the data-parallel
# structure 1.88 vs 1.54 -- a fork inside the captured step graph runs on the runtime's own branch streams, which share a hardware queue with the teacher's
# stream (its start gate timed out on 84-99 of 198 steps), whichever stream the capture forked to (the first version took the weight-gradient stream as
# it came, the second one picked by lanes.beside: same result).  Opt-in: MAGIC_DW_EARLY=1.
DW_EARLY = os.environ.get("MAGIC_DW_EARLY", "0") == "1"
_EARLY = {"stream": None, "keep": [], "use": None}
# `use`: the stream the early flush runs on -- given by the trainer, which picks one that is MEASURED to run beside the main, the teacher's and the
# exchange stream (lanes.beside: HIP deals a process's streams onto 4 hardware queues; the first version took the device's weight-gradient stream as
# it came, which shared a queue with the teacher's -- 1.85 instead of 1.43 ms/step).  None: no early flush.


def flush_dw_early(device):
    if not (DW_EARLY and DEFER["active"] and DEFER["queue"]) or lanes.cur != 0 or _EARLY["use"] is None:
        return False
    cur = torch.cuda.current_stream(device)
    ds = _EARLY["use"]
    ds.wait_stream(cur)
    _EARLY["keep"].append((list(DEFER["queue"]), list(RBW_JOBS), list(PART_JOBS)))
    with torch.cuda.stream(ds):
        _flush_dw(None, keep_active=True)
    _EARLY["stream"] = ds
    return True


def join_dw_early():
    ds = _EARLY["stream"]
    if ds is not None:
        torch.cuda.current_stream(ds.device).wait_stream(ds)
        _EARLY["stream"] = None
        _EARLY["keep"].clear()


def flush_dw(group=None, keep_active=False):
    join_dw_early()
    join_dw_stream()
    # gradient lanes (host/lanes.py): an EAGER flush launches on the current stream over operands every lane's stream produced, into every
    # lane's gradient buffer -- order it behind the lanes, and the lanes behind it (the operands are freed afterwards)
    tok = lanes.fence() if (DEFER["queue"] or PART_JOBS or RBW_JOBS) else None
    try:
        _flush_dw(group, keep_active)
    finally:
        lanes.fence_end(tok)


# The deferred queue through the TILE-OWNER launch (csrc/gemm.hip gemm_dw_cat_kernel) instead of the split-K launch + deterministic seam: one workgroup
# per 128 x 128 (64 x 64 for narrow problems) tile of a dW walks all rows of all uses of that Linear with the accumulators in registers -- every dY / X
# panel is read once per tile row / column, dW is read-modify-written once, nothing goes through a workspace, and the sum order is fixed by
# construction.  OPT-IN (MAGIC_DW_FLUSH_CAT=1), measured and rejected as the default in round 5: the MAGIC-S step has ~100 Linears of 128 x 128 -- ONE
# tile each, whose workgroup then walks 30-60 k-tiles alone on its CU: 1.689 vs 1.456 ms/step (`bench.py`, same box; the split-K launch exists for exactly
# this); on the MAGIC-L navigator iteration (the instruction encoder's dW at the end of the pass) 126-130 vs 126-131 ms: neutral.
DW_FLUSH_CAT = os.environ.get("MAGIC_DW_FLUSH_CAT", "0") != "0"
_CAT_TABLES = []           # device tables a captured graph reads: alive for the life of the process; eager flushes reuse a ring of (pinned, device, event)
_CAT_RING = {"slots": [None] * 8, "turn": 0, "stream": {}}


def _cat_tables(host_bytes, device):
    """device copy of a flush's operand tables.  Eager: pinned slot of a ring -> device, on the current stream.  Inside a capture: uploaded on a
    stream of its own and waited for on the host (a copy node in front of every replay of the step's weight-gradient launch would sit on the
    step's critical path); the table then belongs to the graph: kept for good."""
    n = int(host_bytes.nbytes)
    dev = torch.device(device)
    if torch.cuda.is_current_stream_capturing():
        up = _CAT_RING["stream"].get(dev.index)
        if up is None:
            up = _CAT_RING["stream"][dev.index] = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(up):
            d = torch.empty(n, dtype=torch.uint8, device=dev)
            d.copy_(torch.from_numpy(host_bytes))
        up.synchronize()
        _CAT_TABLES.append(d)
        return d
    ring = _CAT_RING
    i = ring["turn"] = (ring["turn"] + 1) % len(ring["slots"])
    slot = ring["slots"][i]
    if slot is not None:
        slot[2].synchronize()
    if slot is None or slot[0].numel() < n or slot[1].device != dev:
        cap = max(2 * n, 1 << 14)
        slot = ring["slots"][i] = (torch.empty(cap, dtype=torch.uint8, pin_memory=True), torch.empty(cap, dtype=torch.uint8, device=dev), torch.cuda.Event())
    slot[0].numpy()[:n] = host_bytes
    slot[1][:n].copy_(slot[0][:n], non_blocking=True)
    slot[2].record()
    return slot[1]


def _flush_dw_cat(part, dt):
    """`part`: queue entries of one dtype.  Returns False (nothing launched) when an entry does not fit the tile-owner form."""
    import numpy as np
    probs, segs = {}, []
    for (dy, x, dW, db, M, N, K, lda, ldb, ldc, sk) in part:
        key = dW.data_ptr()
        j = probs.get(key)
        dbp = db.data_ptr() if db is not None else 0
        if j is None:
            j = probs[key] = len(segs)
            segs.append([(key, dbp, int(N), int(K), int(lda), int(ldb), int(ldc)), []])
        elif segs[j][0] != (key, dbp, int(N), int(K), int(lda), int(ldb), int(ldc)):
            return False               # one gradient queued with two shapes: leave it to the split-K launch
        segs[j][1].append((dy.data_ptr(), x.data_ptr(), int(M)))
        if FLOPS["enabled"]:
            BYTES["dw"] += float(M) * (N + K) * dy.element_size() + 4.0 * N * K
    ve = 8 if dt in (torch.bfloat16, torch.float16) else 4
    for meta, _ in segs:
        if meta[4] % ve or meta[5] % ve or meta[6] < meta[3]:
            return False
    wide = [sg for sg in segs if sg[0][2] >= 128 and sg[0][3] >= 128]
    narrow = [sg for sg in segs if not (sg[0][2] >= 128 and sg[0][3] >= 128)]
    launches, total = [], 0
    for group in (wide, narrow):
        group.sort(key=lambda sg: -len(sg[1]))                  # problems with equally many uses share a launch: fewer empty segments
        for a in range(0, len(group), 96):
            sel = group[a:a + 96]
            n_seg = max(len(sg[1]) for sg in sel)
            dy_t, x_t, m_t = np.zeros((len(sel), n_seg), np.int64), np.zeros((len(sel), n_seg), np.int64), np.zeros((len(sel), n_seg), np.int32)
            for i, (_, uses) in enumerate(sel):
                for k, (a_, b_, m_) in enumerate(uses):
                    dy_t[i, k], x_t[i, k], m_t[i, k] = a_, b_, m_
                dy_t[i, len(uses):], x_t[i, len(uses):] = uses[0][0], uses[0][1]          # (empty segments: any valid address)
            launches.append(([sg[0] for sg in sel], n_seg, dy_t, x_t, m_t, total))
            total = (total + len(sel) * n_seg * 20 + 15) & ~15
    host = np.zeros(max(total, 16), np.uint8)
    for probs_, n_seg, dy_t, x_t, m_t, off in launches:
        n = len(probs_) * n_seg
        host[off:off + 8 * n] = dy_t.reshape(-1).view(np.uint8)
        host[off + 8 * n:off + 16 * n] = x_t.reshape(-1).view(np.uint8)
        host[off + 16 * n:off + 20 * n] = m_t.reshape(-1).view(np.uint8)
    base = _cat_tables(host, part[0][0].device).data_ptr()
    for probs_, n_seg, dy_t, x_t, m_t, off in launches:
        n = len(probs_) * n_seg
        dw_cat(dt, probs_, n_seg, base + off, base + off + 8 * n, base + off + 16 * n)
        if FLOPS["enabled"]:
            BYTES["dw_launches"] += 1
    return True


def _flush_dw(group=None, keep_active=False):
    group = group or DW_GROUP
    flush_rbw_parts()
    q = DEFER["queue"]
    for dt in {e[0].dtype for e in q}:                 # one compute dtype per launch
        if DW_FLUSH_CAT and _flush_dw_cat([e for e in q if e[0].dtype == dt], dt):
            continue
        part = [e for e in q if e[0].dtype == dt]
        for i in range(0, len(part), group):
            chunk = part[i:i + group]
            arr = (L.DwDesc * len(chunk))()
            for j, (dy, x, dW, db, M, N, K, lda, ldb, ldc, sk) in enumerate(chunk):
                arr[j] = L.DwDesc(L.P(dy), L.P(x), L.P(dW), L.P(db), M, N, K, lda, ldb, ldc, sk)
                if FLOPS["enabled"]:
                    BYTES["dw"] += float(M) * (N + K) * dy.element_size() + 4.0 * N * K
            if FLOPS["enabled"]:
                BYTES["dw_launches"] += 1
            dw_grouped(dt, arr, len(chunk), chunk[0][0].device)
    q.clear()
    DEFER["bytes"] = 0
    if not keep_active:
        DEFER["active"] = False


def dw_grouped(dt, arr, n, device, deterministic=None):
    """one grouped weight-gradient launch over the descriptor array `arr` (deterministic unless switched off or its workspace would
    exceed DW_WS_MAX_BYTES -- MAGIC-L sized launches with one Linear queued 20 times: those fall back to fp32 atomics)"""
    det = DW_DETERMINISTIC if deterministic is None else deterministic
    ws = cnt = None
    nf = nc = 0
    if det:
        f, c = C.c_longlong(0), C.c_int(0)
        rc = L.load().magic_gemm_dw_ws_need(L.dt(dt), n, C.addressof(arr), C.addressof(f), C.addressof(c))      # host-only: no launch
        if rc != 0:
            raise L.MagicHipError(f"magic_gemm_dw_ws_need failed: {rc}")
        nf, nc = int(f.value), int(c.value)
        if nf * 4 <= DW_WS_MAX_BYTES and nc <= DW_COUNTERS:
            cnt = dw_counters(device) if SIDE["stream"] is None else None
            if cnt is None:
                cnt = torch.zeros(max(nc, 1), dtype=torch.int32, device=device)
            ws = _dw_workspace(nf, torch.device(device))
            if cnt is _DW_CNT.get((cnt.device.index, lanes.cur)) or ws is _DW_WS.get((ws.device.index, lanes.cur)):
                dw_guard(ws.device)
    L.call("magic_gemm_dw_grouped", L.dt(dt), n, arr, L.P(ws), int(ws.numel()) if ws is not None else 0, L.P(cnt), int(cnt.numel()) if cnt is not None else 0, L.stream())
    return ws is not None


def dw_cat(dt, probs, n_seg, dy_tab, x_tab, m_tab):
    """dW_p += sum over n_seg row segments of dY^T X for <= 96 Linears in one launch (csrc/gemm.hip gemm_dw_cat_kernel).  probs: [(dW ptr, db ptr, N, K,
    lda, ldb, ldc)]; dy_tab / x_tab / m_tab: DEVICE addresses of the [len(probs)][n_seg] operand-pointer / row-count tables"""
    arr = (L.DwCatProb * len(probs))()
    for j, (dW, db, N, K, lda, ldb, ldc) in enumerate(probs):
        arr[j] = L.DwCatProb(dW, db or None, N, K, lda, ldb, ldc)
    L.call("magic_gemm_dw_cat", L.dt(dt), len(probs), C.addressof(arr), int(n_seg), int(dy_tab), int(x_tab), int(m_tab), L.stream())


def _linear_dw(dy, x, dW, db, M, *, N=None, K=None, lda=None, ldb=None, ldc=None, flop_rows=None):
    N = N if N is not None else dW.shape[0]
    K = K if K is not None else dW.shape[1]
    lda = lda if lda is not None else N
    ldb = ldb if ldb is not None else K
    ldc = ldc if ldc is not None else dW.stride(0)
    tiles = ((N + 63) // 64) * ((K + 63) // 64)
    gemm(2, dy, x, dW, N, K, M, lda, ldb, ldc, splitk=_splitk(tiles, M), bias_grad=db, accumulate=True,
         flop_dims=(N, K, flop_rows if flop_rows is not None else M))


FUSED_LN = not os.environ.get("MAGIC_NO_FUSED_LN")


_LLN_MAXK = int(os.environ.get("MAGIC_LLN_MAXK", "512"))


def linear_ln_ok(H, K=0):
    """fused dense+add+LayerNorm pays while the per-block serial K loop is short (measured: slower than GEMM + LN
    for the teacher's K=1024 FFN output, faster for K <= 512)"""
    return FUSED_LN and H in (128, 256, 384) and K <= _LLN_MAXK


def _dr(drop):
    """drop = None | (seed uint32[2] device tensor, p, site): counter-based dropout descriptor (csrc/common.hpp DropDesc)"""
    if drop is None or drop[1] <= 0.0:
        return (None, 0.0, 0)
    return (L.P(drop[0]), float(drop[1]), int(drop[2]))


def linear_ln(x, W, b, M, residual, gamma, beta, eps, out, rstd, flop_rows=None, drop=None):
    """out = LayerNorm(dropout(x @ W^T + b) + residual) in one launch (H = W.shape[0] in {128,256,384})."""
    H, K = W.shape
    _count(flop_rows if flop_rows is not None else M, H, K, fam="linear_ln")
    L.call("magic_linear_ln", L.dt(x.dtype), M, H, K, L.P(x), x.stride(0), L.P(W), W.stride(0), L.P(b), L.P(residual),
           residual.stride(0) if residual is not None else 0, L.P(gamma), L.P(beta), float(eps), L.P(out), L.P(rstd), *_dr(drop), L.stream())
    return out


def linear_act_ln(x, W, b, M, act, pre_out, gamma, beta, eps, out, rstd, flop_rows=None):
    """out = LayerNorm(act(x @ W^T + b)) in one launch, the pre-activation kept in pre_out (a prediction head's transform; H in {128,256,384});
    act: 1 gelu, 2 relu"""
    H, K = W.shape
    _count(flop_rows if flop_rows is not None else M, H, K, fam="linear_ln")
    L.call("magic_linear_act_ln", L.dt(x.dtype), M, H, K, L.P(x), x.stride(0), L.P(W), W.stride(0), L.P(b), int(act), L.P(pre_out),
           L.P(gamma), L.P(beta), float(eps), L.P(out), L.P(rstd), L.stream())
    return out


FUSED_LNB = not os.environ.get("MAGIC_NO_FUSED_LNB")


def linear_lnbwd_ok(H, K):
    return FUSED_LNB and H in (128, 256) and K <= 512


def linear_lnbwd(x, W, M, residual, y, gamma, beta, rstd, dx, dgamma, dbeta, drop=None, dxm=None, flop_rows=None):
    """dx = LayerNorm-backward(x @ W + residual; y, gamma, beta, rstd) in one launch; W [K, H] = the forward weight of the dense
    whose input gradient this is; dxm = dx * dropout mask (when the LayerNorm's dense branch was dropped)."""
    K, H = W.shape
    _count(flop_rows if flop_rows is not None else M, H, K, fam="linear_ln")
    L.call("magic_linear_lnbwd", L.dt(x.dtype), M, H, K, L.P(x), x.stride(0), L.P(W), W.stride(0), L.P(residual),
           residual.stride(0) if residual is not None else 0, L.P(y), L.P(gamma), L.P(beta), L.P(rstd), L.P(dx), L.P(dxm),
           L.P(dgamma), L.P(dbeta), *_dr(drop), L.stream())
    return dx


def dropout(x, out, rows, cols, ld, drop):
    """out = x * keep / (1-p) with the mask of `drop` over the logical [rows, cols] tensor (row pitch ld); x may be out."""
    L.call("magic_dropout", L.dt(x.dtype), rows, cols, ld, L.P(x), L.P(out), *_dr(drop), L.stream())
    return out


def view_gather(table, vp_row, order, out):
    """out[p, j, :] = table[vp_row[p], order[p, j], :] (order < 0 -> zeros); table [n, 36, D], out [Np, V, D] same dtype."""
    Np, V = order.shape
    _chk(table.dtype == out.dtype and table.is_contiguous() and out.is_contiguous() and table.shape[1] == 36, "view_gather layout")
    _chk(vp_row.dtype == torch.int32 and order.dtype == torch.int32 and order.is_contiguous(), "view_gather indices int32")
    L.call("magic_view_gather", L.dt(table.dtype), Np, V, table.shape[2], L.P(table), table.shape[0], L.P(vp_row), L.P(order), L.P(out), L.stream())
    return out


def _tab(t):
    """t = None | (table, idx|None, mod, off)"""
    if t is None:
        return (None, None, 0, 0)
    return (L.P(t[0]), L.P(t[1]), int(t[2]), int(t[3]))


def ln_fwd(M, H, out, *, in0=None, in1=None, tabs=(None, None, None), gamma=None, beta=None, eps=1e-12, rstd=None, do_ln=True,
           drop_in0=None, drop_out=None, out_drop=None):
    """drop_in0 / drop_out = (seed, p, site): dropout on in0 before the sum / on the output (written to out_drop)."""
    t0, t1, t2 = [_tab(t) for t in tabs]
    d = drop_in0 if (drop_in0 is not None and drop_in0[1] > 0) else drop_out
    seed, p, _ = _dr(d)
    s_in = int(drop_in0[2]) if (p > 0 and drop_in0 is not None) else 0
    s_out = int(drop_out[2]) if (p > 0 and drop_out is not None) else 0
    L.call("magic_ln_fwd", L.dt(out.dtype), M, H, L.P(in0), L.P(in1), *t0, *t1, *t2,
           L.P(gamma), L.P(beta), float(eps), L.P(out), L.P(rstd), 1 if do_ln else 0, seed, p, s_in, s_out, L.P(out_drop), L.stream())
    return out


def _dtab(t):
    """t = None | (idx|None, mod, off, dtab_fp32, small)"""
    if t is None:
        return (None, 0, 0, None, 0)
    return (L.P(t[0]), int(t[1]), int(t[2]), L.P(t[3]), int(t[4]))


SPLIT_PGRAD = bool(os.environ.get("MAGIC_SPLIT_PGRAD"))     # opt-in: measured neutral-to-slower (the extra launch costs what the atomics did)


def ln_bwd(M, H, dy, *, y=None, gamma=None, beta=None, rstd=None, dx=None, dgamma=None, dbeta=None,
           dtabs=(None, None, None), do_ln=True, drop_dy=None, drop_dx=None, dxm=None, hot0=-1):
    """drop_dy: the forward dropped its output (dy masked on load); drop_dx: the forward dropped in0 (dxm = dx * mask); hot0: a row of
    indexed table 0 hit by many input rows (the padding token), reduced per workgroup before the atomics."""
    d0, d1, d2 = [_dtab(t) for t in dtabs]
    dd = drop_dy if (drop_dy is not None and drop_dy[1] > 0) else drop_dx
    seed, p, _ = _dr(dd)
    s_dy = int(drop_dy[2]) if (p > 0 and drop_dy is not None) else 0
    s_dx = int(drop_dx[2]) if (p > 0 and drop_dx is not None) else 0
    partial = 0
    if do_ln and dgamma is not None and part_ok(H):
        # the gamma / beta sums of every workgroup go to its own row of a partial buffer; one column-sum launch per flush adds them up
        nblk = _ln_blocks(M, H, any(t[3] is not None for t in (d0, d1, d2)))
        pt = torch.empty(2, nblk, H, dtype=torch.float32, device=dy.device)
        PART_JOBS.append((pt[0], dgamma, nblk, H, H))
        PART_JOBS.append((pt[1], dbeta, nblk, H, H))
        dgamma, dbeta, partial = pt[0], pt[1], 1
    elif do_ln and dgamma is not None and SPLIT_PGRAD and M >= 512:
        # row kernel fully parallel (no same-address atomics) + a separate low-contention column reduction
        L.call("magic_ln_pgrad", L.dt(dy.dtype), M, H, L.P(dy), L.P(y), L.P(gamma), L.P(beta), L.P(dgamma), L.P(dbeta), L.stream())
        dgamma = dbeta = None
    L.call("magic_ln_bwd", L.dt(dy.dtype), M, H, L.P(dy), L.P(y), L.P(gamma), L.P(beta), L.P(rstd), L.P(dx),
           L.P(dgamma), L.P(dbeta), *d0, *d1, *d2, 1 if do_ln else 0, seed, p, s_dy, s_dx, L.P(dxm), int(hot0), partial, L.stream())
    return dx


def ln_bwd_tail(M, H, dy32, y, gamma, beta, rstd, act_pre, act, dx, dgamma, dbeta, nslab=1):
    """LayerNorm backward of a head's transform with fp32 dy in and act'(act_pre) folded into dx (csrc/rowops.hip magic_ln_bwd_tail):
    cast + ln_bwd + dact as one launch.  nslab > 1: dy32 is [nslab, M, H], the slabs of a deterministic split-K product (gemm(..., splitk=-nslab)),
    added in slab order on load"""
    _chk(dy32.dtype == torch.float32 and dy32.is_contiguous() and y.dtype == act_pre.dtype == dx.dtype, "ln_bwd_tail operands")
    _chk(1 <= int(nslab) <= 256 and dy32.numel() >= int(nslab) * M * H, "ln_bwd_tail slabs")
    act = int(act) | ((int(nslab) << 8) if int(nslab) > 1 else 0)
    if dgamma is not None and part_ok(H) and H <= 384:          # gamma / beta sums of every workgroup to its own row; the flush's column-sum launch adds them up in order (H = 768: the block count of the lean shape differs)
        nblk = _ln_blocks(M, H, False)
        pt = torch.empty(2, nblk, H, dtype=torch.float32, device=dy32.device)
        PART_JOBS.append((pt[0], dgamma, nblk, H, H))
        PART_JOBS.append((pt[1], dbeta, nblk, H, H))
        dgamma, dbeta, act = pt[0], pt[1], act | 0x40
    L.call("magic_ln_bwd_tail", L.dt(y.dtype), M, H, L.P(dy32), L.P(y), L.P(gamma), L.P(beta), L.P(rstd), L.P(act_pre), int(act), L.P(dx),
           L.P(dgamma), L.P(dbeta), L.stream())
    return dx


# Parameter gradients of the row kernels through PARTIAL rows (round 5): at the wide model sizes (H >= 384: MAGIC-B / MAGIC-L) on the few hundred
# rows of a navigator step, a LayerNorm / position-embedding backward launch was nothing but its same-address fp32 atomics (ln_bwd 17.6 us for
# 608 x 768 rows, smallk_ln_bwd 58 us).  Inside a backward pass (the weight-gradient queue is active, so a flush is coming) every workgroup stores
# its sums in its own row and `flush_part_jobs` adds the rows up: one launch per <= 96 vectors, block order, reproducible.
PART_PG = os.environ.get("MAGIC_LN_PARTIAL", "1") != "0"
# Round 6: every reduction that HAS an ordered form takes it by default -- the embedding stage's panorama half through partial rows (embed_in_bwd), the MLM head's
# vocabulary input gradient as split-K slabs (host/model_pretrain.py), and the partial-row parameter gradients at EVERY width (before: from H = 384 up, where they
# are also the faster form).  4 / 3 / 4 of 354 parameter tensors still differ run to run on sap / mlm / cfp (round 5: 37 / 20 / 26 -- and 233 on mlm whenever launch
# timing shifted); + 5 us per step of the headline cycle, all of it the MLM slabs (profiles/micro/r06_ab_determinism_cost*.txt).  MAGIC_DETERMINISTIC=0: partial rows
# from H = 384 only (MAGIC_MLM_DX_ATOMICS=1 / MAGIC_EMBED_BWD_PARTIAL=0 switch the other two back).
DETERMINISTIC = os.environ.get("MAGIC_DETERMINISTIC", "1") != "0"
PART_MIN_H = int(os.environ.get("MAGIC_LN_PARTIAL_MIN_H", "128" if DETERMINISTIC else "384"))
PART_JOBS = []         # (partial rows [nblk, stride] view, destination vector, nblk, len, stride)
_LNB = {}


def part_ok(H):
    return PART_PG and H >= PART_MIN_H and DEFER["active"]


def _ln_blocks(M, H, has_tables=False):
    k = (M, H, bool(has_tables))
    v = _LNB.get(k)
    if v is None:
        v = _LNB[k] = int(L.load().magic_ln_bwd_blocks(M, H, 1 if has_tables else 0))
        _chk(v > 0, "magic_ln_bwd_blocks")
    return v


_SPREL = {}            # destination -> [partial buffer [rows, 2], rows used, index of its two jobs in PART_JOBS] of the current flush window (attn_bwd)


def flush_part_jobs():
    _SPREL.clear()
    while PART_JOBS:
        chunk, rest, seen = [], [], set()
        for job in PART_JOBS:             # one launch holds a destination at most once (its read-modify-write is not atomic)
            key = job[1].data_ptr()
            if key in seen or len(chunk) == 96:
                rest.append(job)
            else:
                seen.add(key)
                chunk.append(job)
        PART_JOBS[:] = rest
        n = len(chunk)
        parts, dsts = (C.c_void_p * n)(), (C.c_void_p * n)()
        nb, ln, st = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
        for i, (pt, dst, k, length, stride) in enumerate(chunk):
            parts[i], dsts[i], nb[i], ln[i], st[i] = pt.data_ptr(), dst.data_ptr(), k, length, stride
        L.call("magic_colsum_add_v", n, C.addressof(parts), C.addressof(dsts), C.addressof(nb), C.addressof(ln), C.addressof(st), L.stream())
        PART_KEEP[:] = [chunk]            # the partial buffers stay alive until the next flush (the launch reads them asynchronously)


PART_KEEP = []


def smallk_ln_fwd(M, H, Kin, x, W, b, gamma, beta, eps, out, rstd):
    _chk(x.dtype == torch.float32 and x.is_contiguous(), "smallk x fp32 contiguous")
    L.call("magic_smallk_ln_fwd", L.dt(out.dtype), M, H, Kin, L.P(x), L.P(W), L.P(b), L.P(gamma), L.P(beta), float(eps),
           L.P(out), L.P(rstd), L.stream())
    return out


def node_in_fwd(H, problems):
    """input stage of 1 or 2 cross-modal encoders in one launch (csrc/rowops.hip node_in_fwd_kernel).  problems: dicts with M, Kin, x, W, b,
    gamma, beta, eps, A, rstd, out and optionally add0 | (src1, csr1[, src2, csr2]) and (tab, tab_idx); csr = (ptr, idx, w)"""
    _chk(1 <= len(problems) <= 2, "node_in_fwd problems")
    arr = (L.NodeIn * len(problems))()
    for j, q in enumerate(problems):
        _chk(q["x"].dtype == torch.float32 and q["x"].is_contiguous(), "node_in x fp32 contiguous")
        d = arr[j]
        d.M, d.Kin, d.eps = int(q["M"]), int(q["Kin"]), float(q["eps"])
        for k in ("x", "W", "b", "gamma", "beta", "A", "rstd", "out", "add0", "src1", "src2", "tab", "tab_idx"):
            setattr(d, k, L.P(q.get(k)))
        for i in (1, 2):
            csr = q.get(f"csr{i}")
            if csr is not None:
                _chk(csr[0].dtype == torch.int32 and csr[0].numel() == d.M + 1, "node_in csr ptr")
                setattr(d, f"ptr{i}", L.P(csr[0])); setattr(d, f"idx{i}", L.P(csr[1])); setattr(d, f"w{i}", L.P(csr[2]))
    L.call("magic_node_in_fwd", L.dt(problems[0]["out"].dtype), H, len(problems), __import__("ctypes").addressof(arr), L.stream())


EMBED_IN = not os.environ.get("MAGIC_NO_EMBED_IN")


def embed_in_fwd(H, pano, text=None):
    """the panorama encoder's input stage (image LayerNorm, location linear + LayerNorm, sum LayerNorm + dropout) and, optionally, the text
    embedding's gathers + LayerNorm + dropout as ONE launch (csrc/rowops.hip embed_in_fwd_kernel): bit-identical to ln_fwd -> smallk_ln_fwd ->
    ln_fwd (+ the text ln_fwd).  pano: dict(M, Kin, eps, P0, g1, b1, A1, rstd1, loc, W, b, g2, b2, A2, rstd2, nav_tab, nav_idx, tok_tab, g3, b3,
    X0, rstd3[, X0d, drop=(seed, p, site)]); text: the keyword arguments of ln_fwd + M + out."""
    a = L.PanoIn()
    a.M, a.Kin, a.eps = int(pano["M"]), int(pano["Kin"]), float(pano["eps"])
    _chk(pano["loc"].dtype == torch.float32 and pano["loc"].is_contiguous() and pano["nav_idx"].dtype == torch.int32, "embed_in_fwd: loc fp32, nav_idx int32")
    for k in ("P0", "g1", "b1", "A1", "rstd1", "loc", "W", "b", "g2", "b2", "A2", "rstd2", "nav_tab", "nav_idx", "tok_tab", "g3", "b3", "X0", "rstd3", "X0d"):
        setattr(a, k, L.P(pano.get(k)))
    seed, p_, site = _dr(pano.get("drop"))
    a.dout.seed, a.dout.site, a.dout.p = seed, int(site), float(p_)
    tp = None
    if text is not None:
        t = L.LnIn()
        t.M, t.do_ln, t.in0, t.in1 = int(text["M"]), 1 if text.get("do_ln", True) else 0, L.P(text.get("in0")), L.P(text.get("in1"))
        for i, tb in enumerate(text.get("tabs", (None, None, None))):
            tab, idx, mod, off = _tab(tb)
            t.tab[i], t.idx[i], t.mod[i], t.off[i] = tab, idx, mod, off
        t.gamma, t.beta, t.eps, t.out, t.rstd = L.P(text.get("gamma")), L.P(text.get("beta")), float(text.get("eps", 1e-12)), L.P(text["out"]), L.P(text.get("rstd"))
        d_in, d_out = text.get("drop_in0"), text.get("drop_out")
        d = d_in if (d_in is not None and d_in[1] > 0) else d_out
        seed, p_, _ = _dr(d)
        t.drop_seed, t.drop_p = seed, float(p_)
        t.site_in0 = int(d_in[2]) if (p_ > 0 and d_in is not None) else 0
        t.site_out = int(d_out[2]) if (p_ > 0 and d_out is not None) else 0
        t.out_drop = L.P(text.get("out_drop"))
        tp = C.addressof(t)
    L.call("magic_embed_in_fwd", L.dt(pano["X0"].dtype), H, C.addressof(a), tp, L.stream())


_EIB_OK = {}
EIB_KEEP = []


def embed_in_bwd_ok(H, Kin):
    if not EMBED_IN:
        return False
    key = (H, Kin)
    if key not in _EIB_OK:
        _EIB_OK[key] = bool(L.load().magic_embed_in_bwd_supported(H, Kin))
    return _EIB_OK[key]


# round 6: the panorama half's (11 + Kin) parameter-gradient vectors through partial rows + the flush's column-sum launch instead of one round of same-address atomics
# per workgroup -- those atomics were the launch's time (csrc/rowops.hip pib_blocks), and the sums become reproducible.  MAGIC_EMBED_BWD_PARTIAL=0: the atomic form.
EMBED_BWD_PARTIAL = os.environ.get("MAGIC_EMBED_BWD_PARTIAL", "1") != "0"


def embed_in_bwd(H, pano, text=None):
    """backward of the panorama encoder's input stage (sum / image / location LayerNorm backwards, nav-type / token-type / loc_linear gradients)
    and, optionally, the text embedding's LayerNorm backward + table scatters as ONE launch (csrc/rowops.hip embed_in_bwd_kernel).
    pano: dict(M, Kin, dy, X0, rstd3, g3, b3, dg3, db3, nav_idx, d_nav, d_tok, A1, rstd1, g1, b1, dg1, db1, dP0, A2, rstd2, g2, b2, dg2, db2,
    loc, dW, dbl[, drop_dy=(seed, p, site)]); text: the keyword arguments of ln_bwd + M + dy."""
    a = L.PanoInBwd()
    a.M, a.Kin = int(pano["M"]), int(pano["Kin"])
    for k in ("dy", "X0", "rstd3", "g3", "b3", "dg3", "db3", "nav_idx", "d_nav", "d_tok", "A1", "rstd1", "g1", "b1", "dg1", "db1", "dP0",
              "A2", "rstd2", "g2", "b2", "dg2", "db2", "loc", "dW", "dbl"):
        setattr(a, k, L.P(pano[k]))
    seed, p_, site = _dr(pano.get("drop_dy"))
    a.ddy.seed, a.ddy.site, a.ddy.p = seed, int(site), float(p_)
    part = None
    if EMBED_BWD_PARTIAL and PART_PG and DEFER["active"]:
        Kin, M = int(pano["Kin"]), int(pano["M"])
        nbt = _ln_blocks(int(text["M"]), H, any(tb is not None and tb[3] is not None for tb in text.get("dtabs", ()))) if text is not None else 0
        nblk = int(L.load().magic_embed_in_bwd_blocks(M, H, nbt, 1))
        _chk(nblk > 0, "magic_embed_in_bwd_blocks")
        stride = (11 + Kin) * H
        part = torch.empty(nblk, stride, dtype=torch.float32, device=pano["dy"].device)
        a.part, a.pad0_, a.pad1_ = L.P(part), nblk, stride
        flat = part.view(-1)
        for off, length, dst in ((0, H, pano["dg3"]), (H, H, pano["db3"]), (2 * H, min(3 * H, pano["d_nav"].numel()), pano["d_nav"]), (5 * H, H, pano["d_tok"]),
                                 (6 * H, H, pano["dg1"]), (7 * H, H, pano["db1"]), (8 * H, H, pano["dg2"]), (9 * H, H, pano["db2"]), (10 * H, H, pano["dbl"]),
                                 (11 * H, H * Kin, pano["dW"])):
            PART_JOBS.append((flat[off:], dst, nblk, length, stride))
    tp = None
    if text is not None:
        t = L.LnBwdIn()
        t.M, t.do_ln = int(text["M"]), 1 if text.get("do_ln", True) else 0
        for k in ("dy", "y", "gamma", "beta", "rstd", "dx", "dgamma", "dbeta"):
            setattr(t, k, L.P(text.get(k)))
        for i, tb in enumerate(text.get("dtabs", (None, None, None))):
            idx, mod, off, dt_, small = _dtab(tb)
            t.idx[i], t.mod[i], t.off[i], t.d[i], t.small[i] = idx, mod, off, dt_, small
        d_dy, d_dx = text.get("drop_dy"), text.get("drop_dx")
        dd = d_dy if (d_dy is not None and d_dy[1] > 0) else d_dx
        seed, p_, _ = _dr(dd)
        t.drop_seed, t.drop_p = seed, float(p_)
        t.site_dy = int(d_dy[2]) if (p_ > 0 and d_dy is not None) else 0
        t.site_dx = int(d_dx[2]) if (p_ > 0 and d_dx is not None) else 0
        t.hot0, t.dxm = int(text.get("hot0", -1)), L.P(text.get("dxm"))
        if text.get("dgamma") is not None and text.get("do_ln", True) and part_ok(H):
            nbt_ = _ln_blocks(int(text["M"]), H, any(tb is not None and tb[3] is not None for tb in text.get("dtabs", ())))
            ptt = torch.empty(2, nbt_, H, dtype=torch.float32, device=text["dy"].device)
            PART_JOBS.append((ptt[0], text["dgamma"], nbt_, H, H))
            PART_JOBS.append((ptt[1], text["dbeta"], nbt_, H, H))
            t.dgamma, t.dbeta, t.partial = L.P(ptt[0]), L.P(ptt[1]), 1
        tp = C.addressof(t)
    # the partial LayerNorm gradients the row-block launches of this backward queued (their vectors are H wide, each destination once) ride along
    take, keep, seen = [], [], set()
    for job in RBW_JOBS:
        key = job[1].data_ptr()
        if len(take) < 96 and key not in seen and job[1].numel() == H:
            seen.add(key)
            take.append(job)
        else:
            keep.append(job)
    if keep:                       # (a destination queued twice: everything goes through flush_rbw_parts, in queue order)
        take = []
    else:
        RBW_JOBS[:] = []
    n = len(take)
    parts, dsts, nb = (C.c_void_p * max(n, 1))(), (C.c_void_p * max(n, 1))(), (C.c_int * max(n, 1))()
    for i, (pt, dst, k) in enumerate(take):
        parts[i], dsts[i], nb[i] = pt.data_ptr(), dst.data_ptr(), k
    L.call("magic_embed_in_bwd", L.dt(pano["X0"].dtype), H, C.addressof(a), tp, n, C.addressof(parts), C.addressof(dsts), C.addressof(nb), L.stream())
    EIB_KEEP[:] = [take]           # the partial buffers stay alive until the next call (the launch reads them asynchronously)


def _skb_part(M, Mmax, H, Kin, dW, db, dgamma, dbeta, device):
    """partial rows [blocks, H (Kin + 3)] of one position-embedding backward + its four column-sum jobs"""
    nblk = int(L.load().magic_smallk_ln_bwd_blocks(M, H, Mmax))       # (pair launch: the tile shape follows the larger problem)
    _chk(nblk > 0, "magic_smallk_ln_bwd_blocks")
    stride = H * (Kin + 3)
    pt = torch.empty(nblk, stride, dtype=torch.float32, device=device)
    flat = pt.view(-1)
    for off, length, dst in ((0, H * Kin, dW), (H * Kin, H, db), (H * (Kin + 1), H, dgamma), (H * (Kin + 2), H, dbeta)):
        PART_JOBS.append((flat[off:], dst, nblk, length, stride))
    return pt


def smallk_ln_bwd(M, H, Kin, x, dy, y, gamma, beta, rstd, dW, db, dgamma, dbeta):
    pt = _skb_part(M, M, H, Kin, dW, db, dgamma, dbeta, dy.device) if part_ok(H) else None
    L.call("magic_smallk_ln_bwd", L.dt(dy.dtype), M, H, Kin, L.P(x), L.P(dy), L.P(y), L.P(gamma), L.P(beta), L.P(rstd),
           L.P(dW), L.P(db), L.P(dgamma), L.P(dbeta), L.P(pt), L.stream())


def softmax_fwd(S, Pout, B, nh, Nq, Nk, ldp, scale, kmask=None, dist=None, sprel_w=None, sprel_b=None):
    _chk(S.dtype == torch.float32 and S.numel() >= B * nh * Nq * ldp and Pout.numel() >= B * nh * Nq * ldp, "softmax buffers")
    L.call("magic_softmax_fwd", L.dt(Pout.dtype), B, nh, Nq, Nk, ldp, L.P(S), L.P(Pout), float(scale), L.P(kmask), L.P(dist),
           L.P(sprel_w), L.P(sprel_b), L.stream())


def softmax_bwd(Pm, dP, dS, B, nh, Nq, Nk, ldp, scale, dist=None, dsprel_w=None, dsprel_b=None):
    L.call("magic_softmax_bwd", L.dt(Pm.dtype), B, nh, Nq, Nk, ldp, L.P(Pm), L.P(dP), L.P(dS), float(scale), L.P(dist),
           L.P(dsprel_w), L.P(dsprel_b), L.stream())


_ATTN_OK = {}


def attn_supported(dtype, Nq, Nk, backward):
    key = (dtype, Nq, Nk, backward)
    if key not in _ATTN_OK:
        _ATTN_OK[key] = bool(L.load().magic_attn_supported(L.dt(dtype), Nq, Nk, 1 if backward else 0))
    return _ATTN_OK[key]


def attn_fwd(q, ldq, k, v, ldkv, Pm, ldp, ctx, B, nh, Nq, Nk, H, scale, kmask=None, dist=None, sprel_w=None, sprel_b=None, flops=0.0,
             drop=None, Pd=None):
    if FLOPS["enabled"]:
        FLOPS["total"] += 4.0 * flops
        FLOPS["attn"] += 4.0 * flops
    L.call("magic_attn_fwd", L.dt(q.dtype), B, nh, Nq, Nk, L.P(q), ldq, L.P(k), L.P(v), ldkv, L.P(Pm), ldp, L.P(ctx), H, float(scale),
           L.P(kmask), L.P(dist), L.P(sprel_w), L.P(sprel_b), *_dr(drop), L.P(Pd), L.stream())


def attn_bwd(q, ldq, k, v, ldkv, Pm, ldp, dctx, B, nh, Nq, Nk, H, scale, dP_init, dq, lddq, dk, dv, lddkv, dist=None, dsprel_w=None,
             dsprel_b=None, flops=0.0, drop=None):
    if dsprel_w is not None and dsprel_b is not None and part_ok(H):
        # (round 6) the graph-distance bias gradients: every (sample, head) workgroup stores its pair, the flush's column-sum launch adds them up in order.  The
        # blocks of one encoder share the two scalars: ONE partial buffer per destination and flush takes the pairs of every block (a column-sum launch
        # holds a destination once: a buffer per block would mean a launch per block)
        key = dsprel_w.data_ptr()
        ent = _SPREL.get(key)
        rows = B * nh
        if ent is not None and not (ent[2] + 1 < len(PART_JOBS) and PART_JOBS[ent[2]][0].data_ptr() == ent[0].data_ptr()):
            ent = None                                   # (the queue was emptied without a flush: host/step_graphs.py)
        if ent is None or ent[1] + rows > ent[0].shape[0]:
            pt = torch.empty(max(16 * rows, 1024), 2, dtype=torch.float32, device=dsprel_w.device)
            ent = _SPREL[key] = [pt, 0, len(PART_JOBS)]
            flat = pt.view(-1)
            PART_JOBS.append((flat, dsprel_w.reshape(-1), 0, 1, 2))
            PART_JOBS.append((flat[1:], dsprel_b.reshape(-1), 0, 1, 2))
        pt, used, j = ent
        ent[1] = used + rows
        for jj in (j, j + 1):
            a_, d_, _, ln_, st_ = PART_JOBS[jj]
            PART_JOBS[jj] = (a_, d_, ent[1], ln_, st_)
        dsprel_w, dsprel_b = pt[used:], None
    if FLOPS["enabled"]:
        FLOPS["total"] += 8.0 * flops
        FLOPS["attn"] += 8.0 * flops
    L.call("magic_attn_bwd", L.dt(q.dtype), B, nh, Nq, Nk, L.P(q), ldq, L.P(k), L.P(v), ldkv, L.P(Pm), ldp, L.P(dctx), H, float(scale),
           L.P(dP_init), L.P(dq), lddq, L.P(dk), L.P(dv), lddkv, L.P(dist), L.P(dsprel_w), L.P(dsprel_b), *_dr(drop), L.stream())


def attn_bwd_ks_ok(dtype, Nq, Nk):
    """fused backward for long keys (csrc/attention.hip attn_bwd_ks_kernel): 16-bit storage, 128 < Nk <= 512"""
    key = (dtype, Nq, Nk, 2)
    if key not in _ATTN_OK:
        _ATTN_OK[key] = bool(L.load().magic_attn_supported(L.dt(dtype), Nq, Nk, 2))
    return _ATTN_OK[key]


def attn_bwd_ks(q, ldq, k, v, ldkv, Pm, ldp, o, dctx, B, nh, Nq, Nk, H, scale, dq, lddq, dk, dv, lddkv, accumulate_kv=False, flops=0.0, drop=None):
    """o: the forward's output [B*Nq, H]; accumulate_kv: dk / dv += (the per-episode K/V cache gradient collected over the steps)"""
    if FLOPS["enabled"]:
        FLOPS["total"] += 8.0 * flops
        FLOPS["attn"] += 8.0 * flops
    L.call("magic_attn_bwd_ks", L.dt(q.dtype), B, nh, Nq, Nk, L.P(q), ldq, L.P(k), L.P(v), ldkv, L.P(Pm), ldp, L.P(o), L.P(dctx), H, float(scale),
           L.P(dq), lddq, L.P(dk), L.P(dv), lddkv, 1 if accumulate_kv else 0, *_dr(drop), L.stream())


FUSED_ENC = not os.environ.get("MAGIC_NO_FUSED_ENC")
ENC_ROW_SPLIT = os.environ.get("MAGIC_ENC_RS", "1") != "0"      # row-split form of the whole-encoder forward (csrc/encoder.hip, encoder_rs_kernel)
XENC_ROW_SPLIT = os.environ.get("MAGIC_XENC_RS", "1") != "0"   # the same for the cross-modal encoders (xencoder_rs_kernel)
CARD_SHARED = [False]


def card_is_shared(shared=True):
    """Tell the launch layer that other processes run kernels on this card (several ranks mapped onto one device: a rehearsal of the
    data-parallel path on a one-card box).  The row-split encoder launches hand activations between workgroups of ONE launch and need
    all of them resident at once; a neighbour's kernels can take the slots the late workgroups need, the bounded waits then give up
    and `check_encoder_health` raises.  On a shared card the one-workgroup-per-sample forms (no waits between workgroups) run instead."""
    global ENC_ROW_SPLIT, XENC_ROW_SPLIT
    CARD_SHARED[0] = bool(shared)
    if shared:
        ENC_ROW_SPLIT = XENC_ROW_SPLIT = False


ENC_SYNC_LAST = [None]                                          # the last launch's sync words (word 0 = 1: a bounded wait gave up) -- tests read it
_ENC_OK = {}


def encoder_ok(dtype, H, I, nh, N, nlayers):
    """whole-encoder launch available for this shape? (bf16, H = 128, 2 heads, FFN 512, <= 80 tokens, <= 6 layers)"""
    if not FUSED_ENC or dtype not in L.HALF:
        return False
    key = (H, I, nh, N, nlayers)
    if key not in _ENC_OK:
        _ENC_OK[key] = bool(L.load().magic_encoder_supported(L.dt(dtype), H, I, nh, N, nlayers))
    return _ENC_OK[key]


def encoder_fwd(segs, seed, p_attn, p_hidden, eps, scale):
    """segs: 1 or 2 dicts(x, kmask, nsamp, N, ldp, layers=[dict of tensors / site ids per layer], flops) -- csrc/encoder.hip.  The
    text encoder goes first (its workgroups run 6 layers: the long pole)."""
    import ctypes as C
    _chk(1 <= len(segs) <= 2, "encoder segments")
    P = L.EncParams()
    P.nseg = len(segs)
    P.p_attn, P.p_hidden, P.eps, P.scale = float(p_attn), float(p_hidden), float(eps), float(scale)
    P.seed = L.P(seed)
    for i, sg in enumerate(segs):
        S = P.seg[i]
        x = sg["x"]
        _chk(x.dtype in L.HALF and x.is_contiguous() and x.shape[0] == sg["nsamp"] * sg["N"], "encoder input [nsamp*N, H] bf16 / fp16")
        dtype = x.dtype
        S.x, S.kmask, S.nsamp, S.N, S.ldp, S.nlayers = L.P(x), L.P(sg["kmask"]), sg["nsamp"], sg["N"], sg["ldp"], len(sg["layers"])
        for j, ly in enumerate(sg["layers"]):
            D = S.L[j]
            for k in ("Wqkv", "bqkv", "Wo", "bo", "g1", "be1", "W1", "bi", "W2", "bo2", "g2", "be2", "qkv", "P", "Pd", "ctx", "a", "z", "g", "out",
                      "rstd_a", "rstd_o"):
                setattr(D, k, L.P(ly.get(k)))
            D.site_attn, D.site_ao, D.site_out = int(ly.get("site_attn", 0)), int(ly.get("site_ao", 0)), int(ly.get("site_out", 0))
        if FLOPS["enabled"]:
            FLOPS["total"] += sg["flops"]
            FLOPS["enc"] += sg["flops"]
    if ENC_ROW_SPLIT:         # one workgroup per (sample, 16-row tile); the arrival counters live in a buffer of this call
        words = 4 + 6 * sum(sg["nsamp"] for sg in segs)
        sync = torch.empty((words + 3) // 4 * 4, dtype=torch.int32, device=segs[0]["x"].device)
        P.sync, P.sync_words = L.P(sync), sync.numel()
        ENC_SYNC_LAST[0] = sync
    L.call("magic_encoder_fwd", L.dt(dtype), C.addressof(P), C.sizeof(P), L.stream())


TEACHER_GATE_US = int(os.environ.get("MAGIC_TEACHER_GATE_US", "400"))       # 0: off
TEACHER_GATE_RECENT_US = int(os.environ.get("MAGIC_TEACHER_GATE_RECENT_US", "150"))
GATE_FIELDS = ("calls", "opened", "already_resident", "timeouts", "consecutive_timeouts", "disabled", "skipped")


def gate_stats_new(device):
    """the 8 counters a start gate keeps (csrc/encoder.hip): zeroed device memory owned by whoever inserts the gate"""
    return torch.zeros(8, dtype=torch.int32, device=device)


def gate_report(stats):
    """counters of a start gate as a dict (one small device -> host copy: call it at a point that may synchronise)"""
    v = stats.cpu().tolist()
    return {k: int(v[i]) for i, k in enumerate(GATE_FIELDS)}


def encoder_start_gate(stats, timeout_us=None, recent_us=None):
    """park the current stream until the next whole-encoder launch (another stream's) has its workgroups resident, a launch of the last
    `recent_us` counting as that launch; bounded by `timeout_us`; switches itself off after 3 consecutive timeouts (csrc/encoder.hip)"""
    t = TEACHER_GATE_US if timeout_us is None else int(timeout_us)
    r = TEACHER_GATE_RECENT_US if recent_us is None else int(recent_us)
    if t > 0:
        L.call("magic_encoder_start_gate", t, r, L.P(stats), L.stream())


def encoder_health(device=None):
    """(gave_up, launches) of the row-split whole-encoder kernels since the process started; synchronises.  gave_up != 0: a bounded in-launch
    hand-off wait ran out (the launch was not fully resident) and activations were computed from rows that never arrived -- raise."""
    out = torch.zeros(2, dtype=torch.int32, device=device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    L.call("magic_encoder_health", L.P(out), L.stream())
    v = out.cpu().tolist()
    return int(v[0]), int(v[1])


def check_encoder_health(device=None):
    gave_up, _ = encoder_health(device)
    if gave_up:
        raise L.MagicHipError(f"{gave_up} in-launch hand-off wait(s) of the row-split encoder kernels gave up: the launch was not resident at once "
                              "(a second process on the card, MAGIC_ENC_RS_ALL with a grid larger than the chip); the activations of those steps are wrong")


FUSED_CHAIN = not os.environ.get("MAGIC_NO_CHAIN")
_CHAIN_OK = {}


def chain_ok(dtype, H, I):
    """forward-only row chain (csrc/chain.hip) available? (16-bit storage, H = 256, FFN 1024: the frozen teacher's width)"""
    if not FUSED_CHAIN or dtype not in L.HALF:
        return False
    key = (dtype, H, I)
    if key not in _CHAIN_OK:
        _CHAIN_OK[key] = bool(L.load().magic_chain_supported(L.dt(dtype), H, I))
    return _CHAIN_OK[key]


def chain_tile_rows(rows=0):
    """rows per workgroup of the teacher's row chain (csrc/chain.hip): 0 = query; 1 = by launch size (default: 64-row tiles for launches that
    would need two rounds of 32-row tiles), 32 | 64 = that form always.  Set before any graph capture.  Returns the setting in force."""
    r = L.load().magic_chain_tile_rows(int(rows))
    _chk(r in (1, 32, 64), "chain_tile_rows: 0, 1, 32 or 64")
    return r


def pack_frag(W):
    """[N, K] 16-bit weight -> the same elements in MFMA-fragment order (flat), as csrc/chain.hip reads them (magic_pack_frag_spans)"""
    import ctypes as C
    _chk(W.dtype in L.HALF and W.is_contiguous() and W.dim() == 2, "pack_frag: contiguous 16-bit matrix")
    out = torch.empty(W.numel(), dtype=W.dtype, device=W.device)
    off, rows, cols = (C.c_longlong * 1)(0), (C.c_int * 1)(W.shape[0]), (C.c_int * 1)(W.shape[1])
    L.call("magic_pack_frag_spans", L.P(W), L.P(out), 1, C.addressof(off), C.addressof(rows), C.addressof(cols), L.stream())
    return out


def chain_fwd(x, res, M, Wa, ba, g1, b1, eps, *, y1=None, ffn=None, y2=None, proj=None, proj_out=None, flop_rows=None):
    """y1 = LN(x Wa^T + ba + res); ffn = (W1, bi, W2, bo2, g2, b2, I): y2 = LN(gelu(y1 W1^T + bi) W2^T + bo2 + y1); proj = (Wp, bp, Np):
    proj_out = y_last Wp^T + bp [M, Np].  Every W in fragment order (flat: pack_frag / ParamStore.f_span).  Forward only, no dropout
    (csrc/chain.hip); pairable."""
    import ctypes as C
    H = res.shape[1]
    _chk(x.dtype in L.HALF and x.stride(-1) == 1 and res.is_contiguous() and res.dtype == x.dtype, "chain inputs 16-bit, rows contiguous")
    _chk(x.shape[0] >= M and res.shape[0] >= M and Wa.is_contiguous() and Wa.numel() == H * H and Wa.dtype == x.dtype, "chain stage 1 shapes")
    P = L.ChainParams()
    P.M, P.ld_in, P.eps = int(M), int(x.stride(0)), float(eps)
    P.inp, P.res, P.Wa, P.ba, P.g1, P.b1, P.y1 = L.P(x), L.P(res), L.P(Wa), L.P(ba), L.P(g1), L.P(b1), L.P(y1)
    fl = H * H
    if ffn is not None:
        W1, bi, W2, bo2, g2, b2, I = ffn
        _chk(W1.is_contiguous() and W2.is_contiguous() and W1.numel() == I * H and W2.numel() == I * H and y2 is not None, "chain FFN shapes")
        P.W1, P.bi, P.W2, P.bo2, P.g2, P.b2, P.y2 = L.P(W1), L.P(bi), L.P(W2), L.P(bo2), L.P(g2), L.P(b2), L.P(y2)
        fl += 2 * H * I
    if proj is not None:
        Wp, bp, Np = proj
        _chk(Wp.is_contiguous() and Wp.numel() == Np * H and proj_out is not None and proj_out.is_contiguous() and proj_out.shape[1] == Np,
             "chain projection shapes")
        P.Wp, P.bp, P.proj, P.Np = L.P(Wp), L.P(bp), L.P(proj_out), int(Np)
        fl += Np * H
    if FLOPS["enabled"]:
        f = 2.0 * (M if flop_rows is None else flop_rows) * fl
        FLOPS["total"] += f
        FLOPS["chain"] = FLOPS.get("chain", 0.0) + f
    L.call("magic_chain_fwd", L.dt(x.dtype), C.addressof(P), C.sizeof(P), L.stream())


FUSED_RBW = not os.environ.get("MAGIC_NO_FUSED_RBW")
_RBW_OK = {}


def rowbwd_ok(dtype, H, I):
    if not FUSED_RBW or dtype not in L.HALF:
        return False
    key = (H, I)
    if key not in _RBW_OK:
        lib = L.load()
        _RBW_OK[key] = bool(lib.magic_rowbwd_supported(L.dt(dtype), H, I)) and lib.magic_rowbwd_params_bytes() == __import__("ctypes").sizeof(L.RbwParams)
    return _RBW_OK[key]


# LayerNorm gradients of the row-block backward through PARTIAL buffers (csrc/encbwd.hip ln_bwd_rows, magic_colsum_add): every workgroup stores
# its gamma / beta sums in its own row, one column-sum launch per flush adds them up in block order -- instead of 512 atomics per workgroup
# (measured: the atomics of the 12 row-block launches were 45 us of the 1.52 ms step).  MAGIC_RBW_ATOMICS=1: the atomic form.
RBW_PARTIAL = not os.environ.get("MAGIC_RBW_ATOMICS")
RBW_JOBS = []          # (partial buffer, destination gradient vector, blocks): queued by rowbwd(), launched by flush_rbw_parts()


def flush_rbw_parts():
    """add the queued partial LayerNorm gradients into the parameter gradients (<= 96 vectors per launch); called wherever the weight-gradient
    queue is flushed, i.e. before anything reads those gradients (bucket exchanges, the gradient norm)"""
    flush_part_jobs()
    while RBW_JOBS:
        # one launch holds each destination at most once (its read-modify-write is not atomic): a parameter used by several queued launches
        # -- the navigator's per-step backwards share their LayerNorms -- goes out over as many launches, in queue order
        chunk, rest, seen = [], [], set()
        for job in RBW_JOBS:
            key = job[1].data_ptr()
            if key in seen or len(chunk) == 96:
                rest.append(job)
            else:
                seen.add(key)
                chunk.append(job)
        RBW_JOBS[:] = rest
        n = len(chunk)
        parts, dsts, nb = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_int * n)()
        for i, (pt, dst, k) in enumerate(chunk):
            parts[i], dsts[i], nb[i] = pt.data_ptr(), dst.data_ptr(), k
        L.call("magic_colsum_add", int(chunk[0][1].numel()), n, C.addressof(parts), C.addressof(dsts), C.addressof(nb), L.stream())


def rowbwd_attn_ok(dtype, H, I, nh, N):
    """may a rowbwd segment carry the attention backward of the block above (mode 1 / 2)?"""
    return bool(RBW_ATTN_MODE != 0 and rowbwd_ok(dtype, H, I) and L.load().magic_rowbwd_attn_supported(L.dt(dtype), H, I, nh, N))


# round 6: the attention backward of block j+1 inside the row-block launch of block j (csrc/encbwd.hip attn_tile_stage).  MAGIC_RBW_ATTN=0: the
# round 2-5 structure (magic_rowbwd and magic_attn_bwd alternating, two launches per block).
RBW_ATTN_MODE = int(os.environ.get("MAGIC_RBW_ATTN", "1"))      # 0: alternating launches; 1: inside once one long stack is left (default); 2: inside for every stack from its top


def rowbwd(segs, seed, p_hidden, p_attn=0.0, scale=0.125):
    """segs: 1 or 2 dicts with the fields of magic_rowbwd_seg (tensors) + M + flops: the per-token backward chain of one block per
    segment (csrc/encbwd.hip); segments with `mode` 1 / 2 also run the attention backward of the block above (their N / ldp / qkv_a / P_a / o_a /
    dctx_a / dP_init / dqkv_out / site_attn fields)"""
    _chk(1 <= len(segs) <= 2, "rowbwd segments")
    P = L.RbwParams()
    P.nseg, P.p_hidden, P.seed = len(segs), float(p_hidden), L.P(seed)
    P.p_attn, P.scale = float(p_attn), float(scale)
    att = any(int(sg.get("mode", 0)) for sg in segs)
    subst = {}
    if RBW_PARTIAL:
        rows = 16 if att else int(L.load().magic_rowbwd_rows(sum(int(sg["M"]) for sg in segs)))
        P.pad1 = 1
        for i, sg in enumerate(segs):
            nblk = (int(sg["M"]) + rows - 1) // rows
            if int(sg.get("mode", 0)):
                nblk = (int(sg["M"]) // int(sg["N"])) * ((int(sg["N"]) + 15) // 16)
            for k in ("dg2", "db2", "dg1", "db1"):
                dst = sg.get(k)
                if dst is not None:
                    pt = torch.empty(nblk * dst.numel(), dtype=torch.float32, device=dst.device)
                    subst[(i, k)] = pt
                    RBW_JOBS.append((pt, dst, nblk))
    for i, sg in enumerate(segs):
        S = P.seg[i]
        S.M, S.kt = int(sg["M"]), int(sg.get("kt", 12))
        for k in L.RBW_PTRS:
            setattr(S, k, L.P(subst.get((i, k), sg.get(k))))
        S.site_out, S.site_ao = int(sg.get("site_out", 0)), int(sg.get("site_ao", 0))
        S.mode = int(sg.get("mode", 0))
        if S.mode:
            S.N, S.ldp, S.ntile, S.site_attn = int(sg["N"]), int(sg["ldp"]), (int(sg["N"]) + 15) // 16, int(sg.get("site_attn", 0))
            for k in L.RBW_ATT_PTRS + L.RBW_DIST_PTRS:
                setattr(S, k, L.P(sg.get(k)))
        if FLOPS["enabled"]:
            FLOPS["total"] += sg["flops"]
            FLOPS["enc"] += sg["flops"]
    any_t = next(segs[0][k] for k in ("y2", "qkv_a", "dao_n") if segs[0].get(k) is not None)      # (a mode-2 segment has no LayerNorm operands)
    L.call("magic_rowbwd", L.dt(any_t.dtype), C.addressof(P), C.sizeof(P), L.stream())


_XENC_OK = {}
FUSED_XENC = not os.environ.get("MAGIC_NO_FUSED_XENC")


def xencoder_ok(dtype, H, I, nh, Nq, Nk, nlayers):
    if not FUSED_ENC or not FUSED_XENC or dtype not in L.HALF:
        return False
    key = (H, I, nh, Nq, Nk, nlayers)
    if key not in _XENC_OK:
        lib = L.load()
        _XENC_OK[key] = bool(lib.magic_xencoder_supported(L.dt(dtype), H, I, nh, Nq, Nk, nlayers)) and \
            lib.magic_xencoder_params_bytes() == __import__("ctypes").sizeof(L.XParams)
    return _XENC_OK[key]


def xencoder_fwd(segs, seed, p_attn, p_hidden, eps, scale):
    """segs: 1 or 2 dicts(x, cx, qmask, cmask, dist, sprel_w, sprel_b, nsamp, Nq, Nk, ldps, ldpc, layers=[...], flops): the global and the
    local co-attention encoder as one launch (csrc/encoder.hip, xencoder_fwd_kernel)"""
    import ctypes as C
    _chk(1 <= len(segs) <= 2, "cross-encoder segments")
    P = L.XParams()
    P.nseg = len(segs)
    P.p_attn, P.p_hidden, P.eps, P.scale = float(p_attn), float(p_hidden), float(eps), float(scale)
    P.seed = L.P(seed)
    for i, sg in enumerate(segs):
        S = P.seg[i]
        x, cx = sg["x"], sg["cx"]
        _chk(x.dtype in L.HALF and x.is_contiguous() and x.shape[0] == sg["nsamp"] * sg["Nq"], "cross-encoder queries [nsamp*Nq, H] bf16 / fp16")
        _chk(cx.dtype == x.dtype and cx.is_contiguous() and cx.shape[0] == sg["nsamp"] * sg["Nk"], "cross-encoder context [nsamp*Nk, H], the queries' dtype")
        dtype = x.dtype
        S.x, S.cx, S.qmask, S.cmask = L.P(x), L.P(cx), L.P(sg["qmask"]), L.P(sg["cmask"])
        S.dist, S.sprel_w, S.sprel_b = L.P(sg.get("dist")), L.P(sg.get("sprel_w")), L.P(sg.get("sprel_b"))
        S.nsamp, S.Nq, S.Nk, S.ldps, S.ldpc, S.nlayers = sg["nsamp"], sg["Nq"], sg["Nk"], sg["ldps"], sg["ldpc"], len(sg["layers"])
        for j, ly in enumerate(sg["layers"]):
            D = S.L[j]
            for k in L.XL_PTRS:
                setattr(D, k, L.P(ly.get(k)))
            for k in ("site_attn", "site_ao", "site_cattn", "site_co", "site_out"):
                setattr(D, k, int(ly.get(k, 0)))
        if FLOPS["enabled"]:
            FLOPS["total"] += sg["flops"]
            FLOPS["enc"] += sg["flops"]
    if ENC_ROW_SPLIT and XENC_ROW_SPLIT:         # one workgroup per (sample, 16-row query tile) when the launch's tiles fit the chip (the entry point decides)
        words = 4 + 6 * sum(sg["nsamp"] for sg in segs)
        sync = torch.empty((words + 3) // 4 * 4, dtype=torch.int32, device=segs[0]["x"].device)
        P.sync, P.sync_words = L.P(sync), sync.numel()
        ENC_SYNC_LAST[0] = sync
    L.call("magic_xencoder_fwd", L.dt(dtype), C.addressof(P), C.sizeof(P), L.stream())


def head_mean_fwd(Pm, out, B, nh, inner):
    L.call("magic_head_mean_fwd", L.dt(Pm.dtype), B, nh, inner, L.P(Pm), L.P(out), L.stream())


def head_mean_bwd(g, dP, B, nh, inner, accumulate=False):
    L.call("magic_head_mean_bwd", B, nh, inner, L.P(g), L.P(dP), 1 if accumulate else 0, L.stream())


def lndot_fwd(Y, M, H, gamma, beta, eps, w2, b2, logit):
    L.call("magic_lndot_fwd", L.dt(Y.dtype), M, H, L.P(Y), L.P(gamma), L.P(beta), float(eps), L.P(w2), L.P(b2), L.P(logit), L.stream())


def lndot_bwd(Y, M, H, gamma, beta, eps, w2, dlogit, dZ, dgamma, dbeta, dw2, db2):
    part = None
    if part_ok(H):             # (round 6) every workgroup's four sums to its own row, added up in block order by the flush's column-sum launch
        nblk = int(L.load().magic_lndot_bwd_blocks(int(M)))
        stride = 3 * H + 1
        part = torch.empty(nblk, stride, dtype=torch.float32, device=Y.device)
        flat = part.view(-1)
        for off, length, dst in ((0, H, dgamma), (H, H, dbeta), (2 * H, H, dw2), (3 * H, 1, db2)):
            PART_JOBS.append((flat[off:], dst.reshape(-1), nblk, length, stride))
    L.call("magic_lndot_bwd", L.dt(Y.dtype), M, H, L.P(Y), L.P(gamma), L.P(beta), float(eps), L.P(w2), L.P(dlogit), L.P(dZ),
           L.P(dgamma), L.P(dbeta), L.P(dw2), L.P(db2), L.P(part), L.stream())


def ce_rows(logits, M, N, ld, labels, *, ignore_index=-100, coef=0.0, row_w=None, loss_row=None, dlogits=None, ldd=0,
            accumulate=False, w_out=None, w_rate=0.0):
    _chk(labels.dtype == torch.int32, "labels int32")
    L.call("magic_ce_rows", L.dt(logits.dtype), M, N, L.P(logits), ld, L.P(labels), ignore_index, float(coef), L.P(row_w),
           L.P(loss_row), L.P(dlogits), ldd, 1 if accumulate else 0, L.P(w_out), float(w_rate), L.stream())


def softkl_rows(logits, M, N, ld, targets, *, coef=0.0, row_w=None, loss_row=None, dlogits=None, ldd=0):
    """per-row KL(targets || softmax(logits)) and its coef-scaled gradient (MRC head)"""
    _chk(targets.dtype == torch.float32 and targets.stride(-1) == 1, "softkl targets fp32")
    _chk(row_w is None or (row_w.dtype == torch.float32 and row_w.numel() >= M), "softkl row_w fp32 [M]")
    L.call("magic_softkl_rows", L.dt(logits.dtype), M, N, L.P(logits), ld, L.P(targets), targets.stride(0), float(coef),
           L.P(row_w), L.P(loss_row), L.P(dlogits), ldd, L.stream())


def kd_rows(s, t, M, N, ld, temperature, *, w=None, norm=1.0, coef=0.0, coef_dev=None, loss_row=None, ds=None, accumulate=False):
    _chk(s.dtype == torch.float32 and t.dtype == torch.float32, "kd logits fp32")
    L.call("magic_kd_rows", M, N, L.P(s), L.P(t), ld, float(temperature), L.P(w), float(norm), float(coef), L.P(coef_dev), L.P(loss_row),
           L.P(ds), 1 if accumulate else 0, L.stream())


def mse(s, t, outer, inner, s_stride, t_stride, *, w=None, rows_per_w=1, norm=1.0, coef=0.0, coef_dev=None, loss=None, ds=None, g_stride=0,
        accumulate=False):
    _chk(s.dtype == t.dtype, "mse dtypes")
    g_f32 = 1 if (ds is not None and ds.dtype == torch.float32) else 0
    if s.dtype == torch.float32:
        g_f32 = 1
    L.call("magic_mse", L.dt(s.dtype), g_f32, outer, inner, L.P(s), s_stride, L.P(t), t_stride, L.P(w), rows_per_w, float(norm),
           float(coef), L.P(coef_dev), L.P(loss), L.P(ds), g_stride, 1 if accumulate else 0, L.stream())


def mse_multi(problems):
    """problems: list of dicts with the keyword arguments of mse() (s, t, outer, inner, s_stride, t_stride, w, rows_per_w, norm,
    coef, coef_dev, loss, ds, g_stride, accumulate) -- all of one compute dtype; one launch for up to 10 of them."""
    # one launch for all terms: fp32 inputs (the head-mean panorama attention maps) ride inside the 16-bit launch (descriptor flag bit 1)
    half = [q["s"].dtype for q in problems if q["s"].dtype != torch.float32]
    dt0 = half[0] if half else torch.float32
    i = 0
    while i < len(problems):
        chunk = problems[i:i + 10]
        arr = (L.MseDesc * len(chunk))()
        for j, q in enumerate(chunk):
            s_, t_, ds = q["s"], q["t"], q.get("ds")
            in32 = s_.dtype == torch.float32
            _chk(s_.dtype == t_.dtype and (in32 or s_.dtype == dt0), "mse_multi dtypes")
            _chk(not in32 or ds is None or ds.dtype == torch.float32, "mse_multi: fp32 inputs take an fp32 gradient")
            g_f32 = (3 if dt0 != torch.float32 else 1) if in32 else (1 if (ds is not None and ds.dtype == torch.float32) else 0)
            arr[j] = L.MseDesc(g_f32, q["outer"], q["inner"], L.P(s_), q["s_stride"], L.P(t_), q["t_stride"], L.P(q.get("w")),
                               q.get("rows_per_w", 1), float(q.get("norm", 1.0)), float(q.get("coef", 0.0)), L.P(q.get("coef_dev")),
                               L.P(q.get("loss")), L.P(ds), q.get("g_stride", 0), 1 if q.get("accumulate") else 0,
                               L.P(q.get("valid_dev")), L.P(q.get("norm_dev")), int(q.get("valid_mod", 0)))
        L.call("magic_mse_multi", L.dt(dt0), len(chunk), arr, L.stream())
        i += len(chunk)


def csr_gather(src, ptr, idx, w, out, n_out, H, accumulate=False):
    _chk(ptr.dtype == torch.int32 and idx.dtype == torch.int32 and ptr.numel() == n_out + 1, "csr arrays")
    L.call("magic_csr_gather", L.dt(src.dtype), n_out, H, L.P(src), L.P(ptr), L.P(idx), L.P(w), L.P(out), 1 if accumulate else 0,
           L.stream())


def cfp_loss_ok(B, H):
    return B <= 64 and H <= 256 and H % 8 == 0 and not os.environ.get("MAGIC_NO_FUSED_CFP")


_CFP_COUNTER = {}


def cfp_loss(B, H, a, txt, temperature, coef, rows, d_a=None, d_txt=None):
    """a: the three [B,H] head outputs (map, viewpoint, fused); rows: fp32 [6,B] loss rows; d_a (3 tensors) / d_txt: gradients or None"""
    _chk(len(a) == 3 and all(x.is_contiguous() and x.dtype == txt.dtype for x in a) and txt.is_contiguous(), "cfp_loss operands")
    d = d_a if d_a is not None else (None, None, None)
    part = cnt = None
    if d_txt is not None:
        part = torch.empty(3, B, H, dtype=torch.float32, device=txt.device)
        cnt = _CFP_COUNTER.get(txt.device)       # one persistent int32 per device: zero between launches (the kernel resets it)
        if cnt is None:
            cnt = _CFP_COUNTER[txt.device] = torch.zeros(1, dtype=torch.int32, device=txt.device)
    L.call("magic_cfp_loss", L.dt(txt.dtype), B, H, L.P(a[0]), L.P(a[1]), L.P(a[2]), L.P(txt), float(temperature), float(coef), L.P(rows),
           L.P(d[0]), L.P(d[1]), L.P(d[2]), L.P(d_txt), L.P(part), L.P(cnt), L.stream())


def csr_gather_multi(H, problems):
    """<= 4 independent gathers in one launch.  problems: dicts(out, n_out, src1, csr1=(ptr, idx, w)[, src2, csr2][, accumulate])"""
    import ctypes as C
    _chk(1 <= len(problems) <= 4, "csr_gather_multi problems")
    arr = (L.CsrProb * len(problems))()
    for j, q in enumerate(problems):
        d = arr[j]
        d.n_out, d.accumulate = int(q["n_out"]), 1 if q.get("accumulate") else 0
        d.out = L.P(q["out"])
        for i in (1, 2):
            csr = q.get(f"csr{i}")
            if csr is None:
                continue
            _chk(csr[0].dtype == torch.int32 and csr[1].dtype == torch.int32 and csr[0].numel() == d.n_out + 1, "csr arrays")
            setattr(d, f"src{i}", L.P(q[f"src{i}"])); setattr(d, f"ptr{i}", L.P(csr[0])); setattr(d, f"idx{i}", L.P(csr[1])); setattr(d, f"w{i}", L.P(csr[2]))
    L.call("magic_csr_gather_multi", L.dt(problems[0]["out"].dtype), H, len(problems), C.addressof(arr), L.stream())


def smallk_ln_bwd_pair(H, problems):
    """two smallk_ln_bwd problems in one launch; dicts with the arguments of smallk_ln_bwd"""
    import ctypes as C
    _chk(len(problems) == 2, "smallk_ln_bwd_pair takes two problems")
    arr = (L.SkbProb * 2)()
    Mmax = max(int(q["M"]) for q in problems)
    for j, q in enumerate(problems):
        d = arr[j]
        d.M, d.Kin = int(q["M"]), int(q["Kin"])
        for k in ("x", "dy", "y", "gamma", "beta", "rstd", "dW", "db", "dgamma", "dbeta"):
            setattr(d, k, L.P(q[k]))
        if part_ok(H):
            d.part = L.P(_skb_part(d.M, Mmax, H, d.Kin, q["dW"], q["db"], q["dgamma"], q["dbeta"], q["dy"].device))
    L.call("magic_smallk_ln_bwd_pair", L.dt(problems[0]["dy"].dtype), H, C.addressof(arr), L.stream())


def pano_fuse_fwd(x, lens, wf, bf, fused, probs, N, V, H, P=None, nh=0, inner=0, pmean=None):
    """P [N, nh, inner] (+ pmean fp32 [N, inner]): the head-mean of the panorama attention map rides in the same launch"""
    _chk(P is None or (P.dtype == x.dtype and P.is_contiguous() and pmean is not None and pmean.dtype == torch.float32), "pano_fuse head-mean operands")
    L.call("magic_pano_fuse_fwd", L.dt(x.dtype), N, V, H, L.P(x), L.P(lens), L.P(wf), L.P(bf), L.P(fused), L.P(probs),
           L.P(P), int(nh), int(inner), L.P(pmean), L.stream())


def pano_fuse_bwd(x, probs, wf, dfused, dx, dwf, dbf, N, V, H):
    if dwf is not None and dbf is not None and part_ok(H):          # (round 6) the fusion Linear's gradients through partial rows, added up in order by the flush
        nblk = int(L.load().magic_pano_fuse_bwd_blocks(int(N)))
        pt = torch.empty(nblk, H + 1, dtype=torch.float32, device=x.device)
        flat = pt.view(-1)
        PART_JOBS.append((flat, dwf.reshape(-1), nblk, H, H + 1))
        PART_JOBS.append((flat[H:], dbf.reshape(-1), nblk, 1, H + 1))
        dwf, dbf = pt, None
    L.call("magic_pano_fuse_bwd", L.dt(x.dtype), N, V, H, L.P(x), L.P(probs), L.P(wf), L.P(dfused), L.P(dx), L.P(dwf), L.P(dbf),
           L.stream())


def sap_fuse_fwd(B, K, Vp, g_raw, l_raw, fuse_raw, gmask, lmask, fsrc, bwmask, use_gate, gl, ll, fl):
    L.call("magic_sap_fuse_fwd", B, K, Vp, L.P(g_raw), L.P(l_raw), L.P(fuse_raw), L.P(gmask), L.P(lmask), L.P(fsrc), L.P(bwmask),
           1 if use_gate else 0, L.P(gl), L.P(ll), L.P(fl), L.stream())


SAP_LOSS_FUSED = not os.environ.get("MAGIC_NO_SAP_LOSS_FUSED")


def sap_fuse_loss(B, K, Vp, g_raw, l_raw, fuse_raw, gmask, lmask, fsrc, bwmask, use_gate, gl, ll, fl, glab, llab, coef, rows, *, dgl=None, dll=None,
                  dfl=None, ignore_index=-100, t_fused=None, w_rate=0.0, w_out=None, T=1.0, kd_norm=0.0, kd_coef=0.0, kd_coef_dev=None, kd_rows=None):
    """sap_fuse_fwd + the three CE rows + teacher-sample weights + action-distillation rows, one launch (csrc/graphops.hip sap_fuse_loss_kernel)"""
    import ctypes as C
    _chk(glab.dtype == torch.int32 and llab.dtype == torch.int32 and rows.dtype == torch.float32 and rows.numel() >= 3 * B, "sap_fuse_loss labels int32, rows fp32 [3, B]")
    _chk(t_fused is None or (t_fused.dtype == torch.float32 and t_fused.is_contiguous() and tuple(t_fused.shape) == (B, K)), "sap_fuse_loss teacher logits fp32 [B, K]")
    P = L.SapLossParams()
    P.B, P.K, P.Vp, P.use_gate = B, K, Vp, 1 if use_gate else 0
    for k, v in (("g_raw", g_raw), ("l_raw", l_raw), ("fuse_raw", fuse_raw), ("gmask", gmask), ("lmask", lmask), ("fsrc", fsrc), ("bwmask", bwmask),
                 ("gl", gl), ("ll", ll), ("fl", fl), ("glab", glab), ("llab", llab), ("rows", rows), ("dgl", dgl), ("dll", dll), ("dfl", dfl),
                 ("t_fused", t_fused), ("w_out", w_out), ("kd_coef_dev", kd_coef_dev), ("kd_rows", kd_rows)):
        setattr(P, k, L.P(v))
    P.ignore_index, P.coef, P.w_rate, P.T, P.kd_norm, P.kd_coef = int(ignore_index), float(coef), float(w_rate), float(T), float(kd_norm), float(kd_coef)
    L.call("magic_sap_fuse_loss", C.addressof(P), C.sizeof(P), L.stream())


def sap_fuse_bwd(B, K, Vp, g_raw, l_raw, fuse_raw, gmask, lmask, fsrc, bwmask, use_gate, dgl, dll, dfl, dg_raw, dl_raw, dfuse_raw):
    L.call("magic_sap_fuse_bwd", B, K, Vp, L.P(g_raw), L.P(l_raw), L.P(fuse_raw), L.P(gmask), L.P(lmask), L.P(fsrc), L.P(bwmask),
           1 if use_gate else 0, L.P(dgl), L.P(dll), L.P(dfl), L.P(dg_raw), L.P(dl_raw), L.P(dfuse_raw), L.stream())


def step_rng(base_seed, counter, rw_temp, seed_out=None, rw_out=None, zero_me=None, scale_state=None, growth=2.0, backoff=0.5, interval=2000):
    """per-step random scalars in one launch: dropout seed (int32[2]) and / or MKRW weights (fp32[5]); `counter`: int32[1] device word, advanced;
    zero_me: one fp32 word set to 0 (the optimizer's gradient-norm accumulator of the step that begins); scale_state: fp32[4] dynamic loss
    scale {S, 1/S, clean steps, pending} updated by amp.GradScaler's rule from what the previous step's AdamW launch left in `pending`"""
    L.call("magic_step_rng", int(base_seed) & 0xFFFFFFFFFFFFFFFF, L.P(counter), float(rw_temp), L.P(seed_out), L.P(rw_out), L.P(zero_me),
           L.P(scale_state), float(growth), float(backoff), int(interval), L.stream())


def rowgate_fwd(mode, x, M, H, wx, *, e=None, we=None, b0=None, b1=None, out_s=None, out=None, gsave=None):
    """mode 0: out_s[m] = x[m] . wx + b0; mode 1: out[m] = e[m] * sigmoid(x[m] . wx + e[m] . we + b0 + b1), gsave[m] = the gate (csrc/rowops.hip)"""
    L.call("magic_rowgate_fwd", L.dt(x.dtype), M, H, mode, L.P(x), L.P(e), L.P(wx), L.P(we), L.P(b0), L.P(b1), L.P(out_s), L.P(out), L.P(gsave), L.stream())


def rowgate_bwd(mode, x, M, H, wx, *, e=None, we=None, gsave=None, dy=None, dout=None, dx=None, de=None, dwx=None, dwe=None, db0=None, db1=None):
    L.call("magic_rowgate_bwd", L.dt(x.dtype), M, H, mode, L.P(x), L.P(e), L.P(wx), L.P(we), L.P(gsave), L.P(dy), L.P(dout), L.P(dx), L.P(de),
           L.P(dwx), L.P(dwe), L.P(db0), L.P(db1), L.stream())


def seed_scale(scale_state):
    """register (None: clear) the device word the gradient-seeding loss launches recorded from now on multiply their gradient coefficients by
    (csrc/loss.hip magic_seed_scale): scale_state[0] of the dynamic loss scale"""
    L.call("magic_seed_scale", L.P(scale_state))


def loss_assemble(rows, row_w, row_scale, kd_rows, slots, rw, alpha, has_kd, out):
    """out[13] = {supervised, 10 weighted MAKD terms, their sum, total loss} (csrc/loss.hip loss_assemble_kernel)"""
    _chk(rows.dtype == torch.float32 and rows.is_contiguous() and out.dtype == torch.float32 and out.numel() >= 13, "loss_assemble layout")
    L.call("magic_loss_assemble", L.P(rows), rows.numel(), L.P(row_w), float(row_scale), L.P(kd_rows), kd_rows.numel() if kd_rows is not None else 0,
           L.P(slots), L.P(rw), float(alpha), 1 if has_kd else 0, L.P(out), L.stream())
    return out


def sumsq(g, out):
    L.call("magic_sumsq", g.numel(), L.P(g), L.P(out), L.stream())


def sumsq_sched(g, out, step, lr0, warmup, total, b1, b2, lr_ss):
    """sumsq + the device-side schedule step in one launch (out must be zero already); step: int32[2] = {global_step, optimizer state step}"""
    _chk(step.numel() >= 2 and step.dtype == torch.int32, "schedule step words: int32[2] {global_step, optimizer state step}")
    L.call("magic_sumsq_sched", g.numel(), L.P(g), L.P(out), L.P(step), float(lr0), int(warmup), int(total), float(b1), float(b2), L.P(lr_ss), L.stream())


def adamw(n, p, g, m, v, shadow, lr, b1, b2, eps, wd, step_size, sumsq_buf, max_norm, gscale, lr_ss=None, n_decay=-1, zero_grad=False, overflow=None,
          scale_state=None, sched_step=None, decay_first=False):
    """n_decay: the first n_decay elements take the weight decay, the rest none (-1: all); zero_grad: g := 0 after use; overflow: int32[1]
    device counter of steps skipped because the gradient norm was not finite (fp16: GradScaler's skip, never NaN weights); scale_state: the
    dynamic loss scale's fp32[4] (the gradient buffer holds S x the gradient; `pending` is left for the next step's prologue); sched_step: the
    device-side schedule's step words int32[2] {global_step, optimizer state step}: a skipped update takes the state step back by one"""
    _chk(sched_step is None or (sched_step.numel() >= 2 and sched_step.dtype == torch.int32), "schedule step words: int32[2] {global_step, optimizer state step}")
    L.call("magic_adamw", n, L.P(p), L.P(g), L.P(m), L.P(v), L.P(shadow), L.dt(shadow.dtype) if shadow is not None else 1, float(lr), float(b1), float(b2), float(eps), float(wd),
           float(step_size), L.P(sumsq_buf), float(max_norm), float(gscale), L.P(lr_ss), int(n_decay), 1 if zero_grad else 0, L.P(overflow),
           L.P(scale_state), L.P(sched_step), 1 if decay_first else 0, L.stream())


def sched_step(step, lr0, warmup, total, b1, b2, lr_ss, zero_me=None):
    _chk(step.numel() >= 2 and step.dtype == torch.int32, "schedule step words: int32[2] {global_step, optimizer state step}")
    L.call("magic_sched_step", L.P(step), float(lr0), int(warmup), int(total), float(b1), float(b2), L.P(lr_ss), L.P(zero_me), L.stream())


def cast_to(x, dtype, out=None):
    """fp32 <-> bf16 / fp16 conversion kernel (no-op if dtypes agree)."""
    if x.dtype == dtype:
        return x
    _chk((x.dtype == torch.float32) != (dtype == torch.float32) and (x.dtype in L.HALF or dtype in L.HALF), "cast_to: fp32 <-> 16-bit only")
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    to16 = dtype in L.HALF
    L.call("magic_cast", L.dt(dtype if to16 else x.dtype), 1 if to16 else 0, x.numel(), L.P(x), L.P(out), L.stream())
    return out


def add_(y, x):
    L.call("magic_add", L.dt(y.dtype), y.numel(), L.P(x), L.P(y), L.stream())
    return y


def add_n(y, xs):
    """y += sum(xs) (<= 8 tensors of y's shape and dtype), fp32 sum with one rounding"""
    import ctypes as C
    _chk(1 <= len(xs) <= 8 and all(x.dtype == y.dtype and x.numel() == y.numel() and x.is_contiguous() for x in xs), "add_n operands")
    arr = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
    L.call("magic_add_n", L.dt(y.dtype), y.numel(), len(xs), C.addressof(arr), L.P(y), L.stream())
    return y


def dact(dy, z, kind, out=None):
    if out is None:
        out = torch.empty_like(dy)
    L.call("magic_dact", L.dt(dy.dtype), kind, dy.numel(), L.P(dy), L.P(z), L.P(out), L.stream())
    return out
