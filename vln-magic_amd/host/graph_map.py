"""Topological map of one episode: `GraphMap`, `FloydGraph`, `pad_tensors_wgrad`.

The reference imports `GraphMap` from its withheld `models.graph_utils` and `pad_tensors_wgrad` from the withheld
`models.ops` (map_nav_src/r2r/agent.py:29-31); the API below is the one the agent uses:
  GraphMap(start_vp) :755, .update_graph(ob) :757,:1102, .node_positions :185, .graph.visited / .distance / .path :192,:220,:384,
  .node_step_ids :205,:875, .node_stop_scores :990,:1083, .update_node_embed(vp, embed, rewrite=, teacher=) :910-924,
  .get_node_embed(vp, teacher) :206, .get_pos_fts(cur_vp, vpids, heading, elevation) :212,:263,:301, .start_vp :268.
`FloydGraph` exists in the reference as map_nav_src/r2r/speaker_utils.py:501-546 (pinned: tests/golden/nav_loop.pt); the rest is
[LINEAGE] DUET `models/graph_utils.py`, constrained by those call sites and by env.get_gmap_pos_fts (r2r/env.py:213-235).

Storage is index based: viewpoints get dense ids in arrival order, the all-pairs table is a growing float64 matrix relaxed
with one vectorised min-plus step per `update(k)`, and the planner (host/nav_plan.py) reads these arrays directly instead of
calling `.distance()` per pair.  Embeddings handed to `update_node_embed` stay whatever tensors the caller passes (device
tensors in the compat path); the index-plan path (host/nav_rollout.py) never stores embeddings here at all.
"""
import math

import numpy as np
import torch

MAX_DIST = 30.0          # r2r/env.py:22
MAX_STEP = 10.0          # r2r/env.py:23
UNREACHED = 95959595     # speaker_utils.py:503 (its "infinite" distance; kept so .distance() answers identically)


def rel_pos(a, b, base_heading=0.0, base_elevation=0.0):
    """heading / elevation / distance of b seen from a (utils/data.py:157-174); a: (3,), b: (..., 3) float64"""
    a = np.asarray(a, np.float64)                                # (3,) or (..., 3) matching b; base_* scalar or (...)
    b = np.asarray(b, np.float64)
    d = b - a
    xy = np.maximum(np.sqrt(d[..., 0] ** 2 + d[..., 1] ** 2), 1e-8)
    xyz = np.maximum(np.sqrt(d[..., 0] ** 2 + d[..., 1] ** 2 + d[..., 2] ** 2), 1e-8)
    heading = np.arcsin(d[..., 0] / xy)                          # the simulator's x/y axes are swapped
    heading = np.where(b[..., 1] < a[..., 1], np.pi - heading, heading) - base_heading
    elevation = np.arcsin(d[..., 2] / xyz) - base_elevation
    return heading, elevation, xyz


def angle_fts(headings, elevations, size=4):
    """[sin h, cos h, sin e, cos e] * (size // 4), float32 (utils/data.py:176-182)"""
    f = np.stack([np.sin(headings), np.cos(headings), np.sin(elevations), np.cos(elevations)], -1).astype(np.float32)
    return np.tile(f, (1, size // 4)) if size // 4 > 1 else f


class FloydGraph:
    """Incremental all-pairs shortest paths over the viewpoints seen so far (speaker_utils.py:501-546)."""

    def __init__(self, cap=32):
        self.index = {}                                          # viewpoint id -> dense id
        self.names = []
        self._d = np.full((cap, cap), float(UNREACHED))
        self._via = np.full((cap, cap), -1, np.int32)            # -1: direct edge (the reference's "")
        self._seen = np.zeros(cap, bool)

    def _id(self, x):
        i = self.index.get(x)
        if i is None:
            i = self.index[x] = len(self.names)
            self.names.append(x)
            if i >= self._d.shape[0]:
                n = 2 * self._d.shape[0]
                d = np.full((n, n), float(UNREACHED))
                d[:i, :i] = self._d[:i, :i]
                v = np.full((n, n), -1, np.int32)
                v[:i, :i] = self._via[:i, :i]
                s = np.zeros(n, bool)
                s[:i] = self._seen[:i]
                self._d, self._via, self._seen = d, v, s
        return i

    def __len__(self):
        return len(self.names)

    def dist_row(self, i):
        """distances from dense id i to every known viewpoint (0 to itself)"""
        r = self._d[i, :len(self.names)].copy()
        r[i] = 0.0
        return r

    def matrix(self):
        """distances between all known viewpoints, dense-id order (diagonal 0)"""
        n = len(self.names)
        m = self._d[:n, :n].copy()
        np.fill_diagonal(m, 0.0)
        return m

    def distance(self, x, y):
        if x == y:
            return 0
        i, j = self.index.get(x), self.index.get(y)
        if i is None or j is None:
            return UNREACHED
        v = self._d[i, j]
        return UNREACHED if v == UNREACHED else float(v)

    def add_edge(self, x, y, dis):
        i, j = self._id(x), self._id(y)
        if dis < self._d[i, j]:
            self._d[i, j] = self._d[j, i] = dis
            self._via[i, j] = self._via[j, i] = -1

    def update(self, k):
        """relax every pair through k (row/column k cannot change during the pass, so one vectorised step equals the
        reference's double loop), then mark k visited"""
        kk = self._id(k)
        n = len(self.names)
        d = self._d[:n, :n]
        cand = d[:, kk][:, None] + d[kk, :][None, :]
        better = cand < d
        np.fill_diagonal(better, False)
        d[better] = cand[better]
        self._via[:n, :n][better] = kk
        self._seen[kk] = True

    def visited(self, k):
        i = self.index.get(k)
        return bool(i is not None and self._seen[i])

    def _path(self, i, j, out):
        if i == j:
            return
        k = self._via[i, j]
        if k < 0:
            out.append(self.names[j])
        else:
            self._path(i, k, out)
            self._path(k, j, out)

    def path(self, x, y):
        """viewpoints after x up to and including y"""
        out = []
        if x != y:
            self._path(self._id(x), self._id(y), out)
        return out

    def hops(self, i):
        """len(path(names[i], names[j])) for every known j, as an int array"""
        n = len(self.names)
        from . import hostplan
        out = hostplan.hops_row(self._via, n, i)        # native form of the recursion below (csrc/hostplan.c); None: library not built
        if out is not None:
            return out
        memo = {}

        def h(a, b):
            if a == b:
                return 0
            key = (a, b)
            if key not in memo:
                k = self._via[a, b]
                memo[key] = 1 if k < 0 else h(a, k) + h(k, b)
            return memo[key]
        return np.array([h(i, j) for j in range(n)], np.int64)


class GraphMap:
    def __init__(self, start_vp):
        self.start_vp = start_vp
        self.node_positions = {}            # insertion order = the order the agent lists map nodes in (agent.py:185)
        self.graph = FloydGraph()
        self.pos_by_id = np.zeros((32, 3), np.float64)      # positions indexed by the graph's dense ids (planner gathers)
        self.node_embeds = {}
        self.teacher_node_embeds = {}
        self.node_stop_scores = {}
        self.node_nav_scores = {}
        self.node_step_ids = {}

    def update_graph(self, ob):
        p = ob["position"]
        self.node_positions[ob["viewpoint"]] = p
        for cc in ob["candidate"]:
            q = cc["position"]
            self.node_positions[cc["viewpointId"]] = q
            self.graph.add_edge(ob["viewpoint"], cc["viewpointId"],
                                math.sqrt((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2 + (p[2] - q[2]) ** 2))
        self.graph.update(ob["viewpoint"])
        n = len(self.graph)
        if n > self.pos_by_id.shape[0]:
            grown = np.zeros((max(2 * self.pos_by_id.shape[0], n), 3), np.float64)
            grown[:self.pos_by_id.shape[0]] = self.pos_by_id
            self.pos_by_id = grown
        ix = self.graph.index
        self.pos_by_id[ix[ob["viewpoint"]]] = p
        for cc in ob["candidate"]:
            self.pos_by_id[ix[cc["viewpointId"]]] = cc["position"]

    def update_node_embed(self, vp, embed, rewrite=False, teacher=False):
        """visited node: `rewrite=True` pins its embedding; unvisited node: running sum + count of every view of it"""
        store = self.teacher_node_embeds if teacher else self.node_embeds
        if rewrite or vp not in store:
            store[vp] = [embed, 1]
        else:
            cur = store[vp]
            store[vp] = [cur[0] + embed, cur[1] + 1]

    def get_node_embed(self, vp, teacher=False):
        e, n = (self.teacher_node_embeds if teacher else self.node_embeds)[vp]
        return e / n

    def get_pos_fts(self, cur_vp, gmap_vpids, cur_heading, cur_elevation, angle_feat_size=4):
        """[K, 7]: sin/cos heading, sin/cos elevation, line distance / 30, graph distance / 30, graph hops / 10; zeros for None
        (elevation is taken against the horizon, not the camera pitch: [LINEAGE] DUET passes base_elevation=0)"""
        n = len(gmap_vpids)
        ang = np.zeros((n, 2), np.float32)
        dist = np.zeros((n, 3), np.float32)
        for j, vp in enumerate(gmap_vpids):
            if vp is None:
                continue
            h, e, d = rel_pos(self.node_positions[cur_vp], self.node_positions[vp], base_heading=cur_heading, base_elevation=0)
            ang[j] = (h, e)
            dist[j] = (d / MAX_DIST, self.graph.distance(cur_vp, vp) / MAX_DIST, len(self.graph.path(cur_vp, vp)) / MAX_STEP)
        return np.concatenate([angle_fts(ang[:, 0], ang[:, 1], angle_feat_size), dist], 1)

    def save_to_json(self):
        nodes = {vp: {"location": pos, "visited": self.graph.visited(vp)} for vp, pos in self.node_positions.items()}
        edges = [(a, b, self.graph.distance(a, b)) for i, a in enumerate(self.graph.names) for b in self.graph.names[i + 1:]
                 if self.graph._via[self.graph.index[a], self.graph.index[b]] < 0 and self.graph.distance(a, b) != UNREACHED]
        return {"nodes": nodes, "edges": edges}


def pad_tensors_wgrad(tensors, lens=None):
    """pad [n_i, ...] tensors to [B, max n, ...] keeping autograd history (agent.py:234; [LINEAGE] DUET models/ops.py)"""
    lens = [t.size(0) for t in tensors] if lens is None else lens
    mx = max(lens)
    out = []
    for t, n in zip(tensors, lens):
        if n < mx:
            t = torch.cat([t, t.new_zeros((mx - n,) + tuple(t.shape[1:]))], 0)
        out.append(t)
    return torch.stack(out, 0)
