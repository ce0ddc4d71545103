"""RCCL through its C ABI (ctypes), for the gradient exchange of host/trainer.GradSync: `ncclAllReduce` / `ncclAllGather` issued straight onto the
exchange stream -- plain kernel launches on OUR stream, so they can sit inside a captured HIP graph (torch's ProcessGroupNCCL cannot: its watchdog
thread queries events recorded in the capturing stream and aborts with hipErrorCapturedEvent, measured in round 6) and cost no per-call bookkeeping
(`dist.all_reduce`: 20-40 us of host time and two event round trips per call; a step issues 5-8 of them).

The communicator is OURS (ncclCommInitRank with an id rank 0 makes and every rank receives through the already-initialised torch.distributed group --
the reference's `init_distributed`, pretrain_src/utils/misc.py:57-71, has built that group), on the librccl.so torch itself loaded, so one RCCL runtime
serves both.  Same semantics as the torch calls it replaces: in-place sum all-reduce of fp32 ranges, all-gather of equal-sized int64 / fp32 blocks.
One process per GPU; RCCL moves the bytes over xGMI.  `selftest()` all-reduces a vector of ones and checks the world size came back: a communicator
that does not pass it is dropped and GradSync keeps torch.distributed (the path the 2-rank `gloo` rehearsals exercise); `selftest_captured()` does the
same inside a captured, twice-replayed graph under a deadline -- only a communicator that passes it ON EVERY RANK (`graph_ok`) gets its collectives
captured into the step graph, the others issue them between the cut graphs.  N > 1 has never run on hardware in this repository: these two tests are
what stands between an 8-GPU launch and a hang."""
import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_FLOAT32, NCCL_INT64, NCCL_SUM = 7, 4, 0          # rccl.h: ncclDataType_t / ncclRedOp_t


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * 128)]         # rccl.h NCCL_UNIQUE_ID_BYTES (c_ubyte: a c_char array field reads back truncated at the first NUL)


class RcclError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            raise RcclError(f"{path}: torch's RCCL library not found")
        l = C.CDLL(path, mode=C.RTLD_GLOBAL)
        vp, sz = C.c_void_p, C.c_size_t
        l.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        l.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, _UniqueId, C.c_int]
        l.ncclCommDestroy.argtypes = [vp]
        l.ncclCommAbort.argtypes, l.ncclCommAbort.restype = [vp], C.c_int
        l.ncclAllReduce.argtypes = [vp, vp, sz, C.c_int, C.c_int, vp, vp]
        l.ncclAllGather.argtypes = [vp, vp, sz, C.c_int, vp, vp]
        l.ncclGroupStart.argtypes, l.ncclGroupEnd.argtypes = [], []
        l.ncclGetErrorString.argtypes, l.ncclGetErrorString.restype = [C.c_int], C.c_char_p
        for f in (l.ncclGetUniqueId, l.ncclCommInitRank, l.ncclCommDestroy, l.ncclAllReduce, l.ncclAllGather, l.ncclGroupStart, l.ncclGroupEnd):
            f.restype = C.c_int
        _lib = l
    return _lib


def _chk(rc, what):
    if rc != 0:
        raise RcclError(f"{what}: {lib().ncclGetErrorString(rc).decode()} ({rc})")


def _raw_stream():
    return torch.cuda.current_stream().cuda_stream


class RcclComm:
    """one communicator over the ranks of torch.distributed's default group, this rank on `device`"""

    def __init__(self, device):
        self.dev = torch.device(device)
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        l = lib()
        uid = _UniqueId()
        if self.rank == 0:
            _chk(l.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        # the 128 id bytes travel through the group the launcher's init built (device tensor under `nccl`, host tensor under `gloo`)
        on_dev = dist.get_backend() == "nccl"
        t = torch.frombuffer(bytearray(C.string_at(C.byref(uid), 128) if self.rank == 0 else bytes(128)), dtype=torch.uint8).clone()
        t = t.to(self.dev) if on_dev else t
        if self.world > 1:
            dist.broadcast(t, src=0)
        raw = bytes(t.cpu().tolist())
        C.memmove(C.byref(uid), raw, 128)
        self.comm = C.c_void_p()
        self.graph_ok = False                     # set by make(): the captured self-test passed on EVERY rank
        with torch.cuda.device(self.dev):
            _chk(l.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")

    def all_reduce_(self, t):
        """in-place fp32 sum over the ranks, on torch's current stream"""
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RcclError("all_reduce_: contiguous fp32 tensors only")
        _chk(lib().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), NCCL_FLOAT32, NCCL_SUM, self.comm, _raw_stream()), "ncclAllReduce")

    def all_gather(self, out, inp):
        """out[r * n : (r + 1) * n] = rank r's inp (n = inp.numel(); fp32 or int64), on torch's current stream"""
        dt = {torch.float32: NCCL_FLOAT32, torch.int64: NCCL_INT64}.get(inp.dtype)
        if dt is None or out.dtype != inp.dtype or not (inp.is_contiguous() and out.is_contiguous()) or out.numel() != inp.numel() * self.world:
            raise RcclError("all_gather: contiguous fp32 / int64 blocks, out = world x inp")
        _chk(lib().ncclAllGather(inp.data_ptr(), out.data_ptr(), inp.numel(), dt, self.comm, _raw_stream()), "ncclAllGather")

    class _Group:
        def __enter__(self):
            _chk(lib().ncclGroupStart(), "ncclGroupStart")

        def __exit__(self, *exc):
            _chk(lib().ncclGroupEnd(), "ncclGroupEnd")
            return False

    def group(self):
        """calls inside become ONE launch (a bucket's chunks; the two gathers of the sparse row exchange)"""
        return RcclComm._Group()

    def selftest(self):
        x = torch.ones(64, dtype=torch.float32, device=self.dev)
        self.all_reduce_(x)
        ids = torch.full((4,), self.rank, dtype=torch.int64, device=self.dev)
        got = torch.empty(4 * self.world, dtype=torch.int64, device=self.dev)
        self.all_gather(got, ids)
        torch.cuda.synchronize(self.dev)
        return bool((x == float(self.world)).all()) and got.view(self.world, 4)[:, 0].tolist() == list(range(self.world))

    def selftest_captured(self, deadline_s=20.0):
        """the same all-reduce INSIDE a captured HIP graph, forked to a side stream as the step's bucket collectives are (trainer.capture_student), replayed
        twice: the values must come back world and world^2 within `deadline_s` -- a node whose RCCL cannot be captured (or whose replay hangs) is found out
        here, at start-up, on a 256 KB vector, not inside the first training step.  False: the caller keeps the cut-graph form.  On a timeout the
        communicator is ABORTED (ncclCommAbort: its kernels leave their spin loops) and must not be used again."""
        import time
        from .lib import capture
        x = torch.ones(1 << 16, dtype=torch.float32, device=self.dev)
        side = torch.cuda.Stream(device=self.dev)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(self.dev)
        with torch.cuda.device(self.dev):
            try:
                with capture(g, capture_error_mode="relaxed"):      # (the mode the step captures use; other threads -- torch's NCCL watchdog -- keep making HIP calls meanwhile)
                    cur = torch.cuda.current_stream()
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        with self.group():
                            self.all_reduce_(x[:1 << 15])
                            self.all_reduce_(x[1 << 15:])
                    cur.wait_stream(side)
            except Exception:              # noqa: BLE001 -- a refused capture is an answer, not an error
                return False
            done = torch.cuda.Event()
            g.replay()
            g.replay()
            done.record()
            t0 = time.monotonic()
            while not done.query():
                if time.monotonic() - t0 > deadline_s:
                    lib().ncclCommAbort(self.comm)
                    self.comm = None
                    return False
                time.sleep(0.002)
        return bool((x == float(self.world) ** 2).all())

    def destroy(self):
        if getattr(self, "comm", None):
            lib().ncclCommDestroy(self.comm)
            self.comm = None


def make(device):
    """a self-tested communicator, or None (no `nccl` group / MAGIC_RCCL_DIRECT=0 / anything failed: the caller keeps torch.distributed)"""
    if os.environ.get("MAGIC_RCCL_DIRECT", "1") == "0" or not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl":
        return None
    c, ok, why = None, False, ""
    try:
        c = RcclComm(device)
        ok = c.selftest()
        if ok:
            c.graph_ok = c.selftest_captured()
            ok = c.comm is not None                    # (a timed-out captured test aborted the communicator)
    except Exception as e:              # noqa: BLE001 -- never take the run down: torch.distributed does the same exchange
        ok, why = False, repr(e)
    # every rank must take the SAME path (a rank on torch.distributed and its peer on the direct communicator would wait for each other for ever):
    # the verdicts are min-reduced over the group the launcher built
    flags = torch.tensor([1 if ok else 0, 1 if (ok and c.graph_ok) else 0], dtype=torch.int32, device=torch.device(device))
    if dist.get_world_size() > 1:
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    all_ok, all_graph = (int(v) for v in flags.tolist())
    if c is not None and ok and all_ok:
        c.graph_ok = bool(all_graph)
        return c
    if c is not None and ok:
        c.destroy()
    import warnings
    warnings.warn(f"direct RCCL communicator not available on every rank ({why or 'self-test failed'}): gradient exchange through torch.distributed")
    return None
