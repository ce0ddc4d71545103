"""ctypes binding of `_magic_hostplan.so` (csrc/hostplan.c): native forms of the navigator planner's two hot loops -- the hop counts of
`FloydGraph.path` (speaker_utils.py:527-546) and the nDTW expert's table rows (agent.py:356-363, eval_utils.py cal_dtw).  Host-side
bookkeeping, no GPU: when the library has not been built the callers keep their Python forms (same arithmetic, same order, same results:
tests/test_navplan_cpu.py runs both)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(os.path.dirname(_HERE), "_magic_hostplan.so")
_lib = None
_tried = False


def lib():
    global _lib, _tried
    if not _tried:
        _tried = True
        if os.path.exists(PATH) and not os.environ.get("MAGIC_NO_HOSTPLAN"):
            l = C.CDLL(PATH)
            vp, i32 = C.c_void_p, C.c_int
            l.mp_hops_row.argtypes = [vp, i32, i32, i32, vp, vp]
            l.mp_dtw_cands.argtypes = [vp, i32, vp, i32, vp, vp, vp, i32, vp, vp]
            l.mp_dtw_extend.argtypes = [vp, i32, vp, i32, vp, vp, i32, vp, vp]
            for f in (l.mp_hops_row, l.mp_dtw_cands, l.mp_dtw_extend):
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
