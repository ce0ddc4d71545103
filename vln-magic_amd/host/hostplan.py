"""ctypes binding of `_magic_hostplan.so` (csrc/hostplan.c): native forms of the navigator planner's two hot loops -- the hop counts of
`FloydGraph.path` (speaker_utils.py:527-546) and the nDTW expert's table rows (agent.py:356-363, eval_utils.py cal_dtw).  Host-side
bookkeeping, no GPU: when the library has not been built the callers keep their Python forms (same arithmetic, same order, same results:
tests/test_navplan_cpu.py runs both)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.environ.get("MAGIC_HOSTPLAN_PATH") or os.path.join(os.path.dirname(_HERE), "_magic_hostplan.so")      # (env: the sanitizer build, csrc/Makefile `asan`)
SRC = os.path.join(os.path.dirname(_HERE), "csrc", "hostplan.c")
ABI_VERSION = 2              # csrc/hostplan.c MP_ABI_VERSION
_lib = None
_tried = False


class HostPlanError(RuntimeError):
    pass


def _check(l):
    """refuse a stale object: ctypes would call an older mp_plan_nav signature with today's argument list and corrupt memory silently"""
    try:
        l.mp_abi_version.restype, l.mp_src_id.restype = C.c_int, C.c_char_p
        abi, sid = int(l.mp_abi_version()), l.mp_src_id().decode()
    except AttributeError:
        raise HostPlanError(f"{PATH} predates mp_abi_version(): rebuild (make -C vln-magic_amd/csrc)") from None
    if abi != ABI_VERSION:
        raise HostPlanError(f"{PATH} reports ABI {abi}, this tree binds ABI {ABI_VERSION}: rebuild (make -C vln-magic_amd/csrc)")
    if os.path.exists(SRC):
        import hashlib
        with open(SRC, "rb") as f:
            want = hashlib.sha256(f.read()).hexdigest()[:16]
        if sid != want:
            raise HostPlanError(f"{PATH} was built from another csrc/hostplan.c (id {sid}, source {want}): rebuild (make -C vln-magic_amd/csrc)")


def lib():
    global _lib, _tried
    if not _tried:
        _tried = True
        if os.path.exists(PATH) and not os.environ.get("MAGIC_NO_HOSTPLAN"):
            l = C.CDLL(PATH)
            _check(l)
            vp, i32 = C.c_void_p, C.c_int
            l.mp_hops_row.argtypes = [vp, i32, i32, i32, vp, vp]
            l.mp_dtw_cands.argtypes = [vp, i32, vp, i32, vp, vp, vp, i32, vp, vp]
            l.mp_dtw_extend.argtypes = [vp, i32, vp, i32, vp, vp, i32, vp, vp]
            i64 = C.c_longlong
            l.mp_batch_new.argtypes, l.mp_batch_new.restype = [i32], vp
            l.mp_batch_free.argtypes, l.mp_batch_free.restype = [vp], None
            l.mp_batch_set.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32]
            l.mp_update_graph.argtypes = [vp, i32, i32, i32, vp, i32]
            l.mp_plan_nav.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i64, i64, i64] + [vp] * 11 + [vp, vp, vp, i32, vp]
            for f in (l.mp_hops_row, l.mp_dtw_cands, l.mp_dtw_extend, l.mp_batch_set, l.mp_update_graph, l.mp_plan_nav):
                f.restype = i32
            _lib = l
    return _lib


def hops_row(via, n, i):
    """via: int32 [cap, cap] C-contiguous; returns int64 [n] or None (library not built)"""
    l = lib()
    if l is None or n > 512:
        return None
    out = np.empty(n, np.int64)
    memo = np.empty(n * n, np.int16)
    if l.mp_hops_row(via.ctypes.data, via.shape[1], n, int(i), out.ctypes.data, memo.ctypes.data) != 0:
        return None
    return out


class DenseDist:
    """dense float64 copy of a scan's `shortest_distances[scan]` (dict of dicts, r2r/env.py:115-119) + viewpoint -> index: built once per scan"""

    def __init__(self, dist):
        self.names = list(dist.keys())
        self.index = {v: i for i, v in enumerate(self.names)}
        n = len(self.names)
        self.D = np.empty((n, n), np.float64)
        for a, row in dist.items():
            ia = self.index[a]
            for b, v in row.items():
                self.D[ia, self.index[b]] = v
        self.n = n
        self.paths = {}           # (from, to) -> int32 dense indices of shortest_paths[from][to][1:] (filled by the expert as it asks)


def dtw_extend(dd, row, nodes, ref_idx):
    """row (list / array of G + 1 floats) advanced by the path nodes `nodes` (viewpoint names); returns a float64 array"""
    l = lib()
    G = len(ref_idx)
    r = np.ascontiguousarray(row, np.float64)
    nd = np.array([dd.index[v] for v in nodes], np.int32)
    out = np.empty(G + 1, np.float64)
    tmp = np.empty(2 * (G + 1), np.float64)
    l.mp_dtw_extend(dd.D.ctypes.data, dd.n, r.ctypes.data, G, ref_idx.ctypes.data, nd.ctypes.data, len(nd), out.ctypes.data, tmp.ctypes.data)
    return out


def dtw_cands(dd, row, paths, ref_idx):
    """last DTW entry of `row` extended by each of `paths` (lists of viewpoint names): float64 [len(paths)]"""
    l = lib()
    G = len(ref_idx)
    r = np.ascontiguousarray(row, np.float64)
    ix = dd.index
    flat, offs = [], [0]
    for p in paths:
        flat += [ix[v] for v in p]
        offs.append(len(flat))
    nodes = np.array(flat if flat else [0], np.int32)
    offs = np.array(offs, np.int32)
    out = np.empty(max(len(paths), 1), np.float64)
    tmp = np.empty(2 * (G + 1), np.float64)
    l.mp_dtw_cands(dd.D.ctypes.data, dd.n, r.ctypes.data, G, ref_idx.ctypes.data, nodes.ctypes.data, offs.ctypes.data, len(paths), out.ctypes.data,
                   tmp.ctypes.data)
    return out[:len(paths)]


def dtw_cands_idx(dd, row, idx_paths, ref_idx):
    """dtw_cands over paths already given as dense node-index arrays (int32; DenseDist.paths caches them per (from, to): a scan's shortest paths
    never change, the per-node dict lookups of `dtw_cands` were most of the expert's host time)"""
    l = lib()
    G = len(ref_idx)
    r = np.ascontiguousarray(row, np.float64)
    n = len(idx_paths)
    offs = np.zeros(n + 1, np.int32)
    if n:
        offs[1:] = np.cumsum([len(p) for p in idx_paths])
    nodes = np.concatenate(idx_paths) if (n and offs[-1]) else np.zeros(1, np.int32)
    out = np.empty(max(n, 1), np.float64)
    tmp = np.empty(2 * (G + 1), np.float64)
    l.mp_dtw_cands(dd.D.ctypes.data, dd.n, r.ctypes.data, G, ref_idx.ctypes.data, nodes.ctypes.data, offs.ctypes.data, n, out.ctypes.data, tmp.ctypes.data)
    return out[:n]


CAP, VMAX = 128, 64          # nodes per episode the native planner state STARTS with (it doubles on demand up to MAXN) / views recorded per unvisited node
MAXN = 512                   # csrc/hostplan.c MP_MAXN: the most map nodes one episode may hold in the native planner


class NativeBatch:
    """the per-episode planner state of one rollout registered with csrc/hostplan.c (mp_batch_*): FloydGraph / GraphMap arrays by pointer +
    the planner's own by-dense-id arrays (step ids, fused-embedding rows, rows of the views that showed an unvisited node)"""

    def __init__(self, gmaps):
        l = lib()
        self.B = len(gmaps)
        self.h = l.mp_batch_new(self.B)
        if not self.h:
            raise MemoryError("mp_batch_new")
        self.step = np.zeros((self.B, CAP), np.int64)
        self.fused = np.full((self.B, CAP), -1, np.int64)
        self.vcount = np.zeros((self.B, CAP), np.int32)
        self.vrows = np.zeros((self.B, CAP, VMAX), np.int64)
        self.keys = [None] * self.B
        self.gmaps = gmaps
        for i in range(self.B):
            self.register(i)

    def grow(self, n):
        """make room for `n` nodes per episode: the by-dense-id arrays double (contents kept) and every episode re-registers its new rows.  The reference's
        map has no bound (map_nav_src/r2r/agent.py: GraphMap grows with every observation); RxR rollouts (max_action_len 28, sampled actions) can pass the
        128 nodes the state starts with."""
        cap = self.step.shape[1]
        if n <= cap:
            return
        if n > MAXN:
            raise ValueError(f"native planner: an episode's map grew to {n} nodes, past the {MAXN} the C core is built for (MAGIC_NO_HOSTPLAN=1 selects the Python planner)")
        new = min(MAXN, max(2 * cap, n))

        def wider(a, fill):
            b = np.full((a.shape[0], new) + a.shape[2:], fill, a.dtype)
            b[:, :cap] = a
            return b
        self.step, self.fused, self.vcount, self.vrows = wider(self.step, 0), wider(self.fused, -1), wider(self.vcount, 0), wider(self.vrows, 0)
        self.keys = [None] * self.B
        for i in range(self.B):
            self.register(i)

    def register(self, i):
        g = self.gmaps[i]
        gr = g.graph
        objs = self.keys[i]
        if objs is not None and objs[0] is gr._d and objs[1] is gr._via and objs[2] is gr._seen and objs[3] is g.pos_by_id:
            return                      # (the arrays registered last time are still the graph's: FloydGraph reallocates only when it outgrows its capacity)
        key = (gr._d.ctypes.data, gr._via.ctypes.data, gr._seen.ctypes.data, g.pos_by_id.ctypes.data)
        if True:
            if gr._d.shape[0] > 512 or not (gr._d.flags.c_contiguous and gr._via.flags.c_contiguous and g.pos_by_id.flags.c_contiguous):
                raise ValueError("native planner: graph arrays must be C-contiguous with at most 512 nodes")
            lib().mp_batch_set(self.h, i, key[0], key[1], key[2], gr._d.shape[1], key[3], self.step[i].ctypes.data, self.fused[i].ctypes.data,
                               self.vcount[i].ctypes.data, self.vrows[i].ctypes.data, VMAX)
            self.keys[i] = (gr._d, gr._via, gr._seen, g.pos_by_id)

    def __del__(self):
        l, h = _lib, getattr(self, "h", None)
        if l is not None and h:
            l.mp_batch_free(h)
            self.h = None
