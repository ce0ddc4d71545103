"""Index plans for the navigator step loop.

`GMapNavAgent.rollout` (map_nav_src/r2r/agent.py:722-1160) interleaves three model calls per step with per-sample Python
that touches device tensors: `gmap.update_node_embed(vp, pano_embeds[i, j])` per candidate (:905-924), `torch.stack` of
`get_node_embed` per map node and `pad_tensors_wgrad` per sample (:206-210,:234), `nav_probs[i, 0].item()` per sample
(:986-996), `.cpu()` masks for the fusion loops.  None of that bookkeeping depends on tensor VALUES -- only on the
observations the simulator hands to the host -- so it is done here once per step in numpy and reaches the device as index
arrays; the embeddings themselves never leave HBM:

  * every step's panorama outputs live in an append-only device log (rows: B x V views, then B fused, then B [cls]);
  * a map node's embedding is a weighted sum of log rows -- visited: the fused row of the step that visited it
    (`rewrite=True`, :910), unvisited: the mean of every candidate view of it seen so far (:916-924) -- so
    `gmap_img_embeds` and `vp_img_embeds` ([stop], [mem] = previous [cls], views; :206-210,:294-297) are ONE CSR gather over
    the log, and their backward is the transposed CSR (the gradient reaches every earlier step's panorama encoder);
  * positions / step ids / pair distances / masks / fusion map / expert actions are the arrays of
    `_nav_gmap_variable` :175-251, `_nav_vp_variable_mem` :290-328, `_teacher_action` :330-373, built from
    `graph_map.GraphMap`'s dense tables instead of per-pair method calls.

Pure numpy: tested on CPU against the reference-style loops (tests/test_navplan_cpu.py).
"""
import math
import os

import numpy as np

from .graph_map import MAX_DIST, MAX_STEP, GraphMap, angle_fts, rel_pos

IGNORE = -100
DEFER_GRAPH = os.environ.get("MAGIC_NAV_DEFER_GRAPH", "1") != "0"      # end_step's graph updates wait until begin_nav (under the panorama launch)


def _csr_from_coo(out_rows, src_rows, w, n_out):
    order = np.argsort(out_rows, kind="stable")
    ptr = np.zeros(n_out + 1, np.int64)
    np.add.at(ptr, out_rows + 1, 1)
    ptr = np.cumsum(ptr).astype(np.int32)
    if len(order) == 0:
        return ptr, np.zeros(1, np.int32), np.zeros(1, np.float32)
    return ptr, src_rows[order].astype(np.int32), w[order].astype(np.float32)


def dtw_rows(dist, row, nodes, ref):
    """advance the DTW table of r2r/eval_utils.py cal_dtw by the path nodes `nodes`: `row` = table row of the path so far
    (row 0 = [0, inf, ...]); the recurrence only looks one row back, so a rollout keeps ONE row per episode for the walked
    prefix and each expert query extends it by the 1-3 nodes of the candidate's connecting path instead of redoing the
    whole (prediction x reference) table per candidate as the reference does (agent.py:356-363)."""
    for p in nodes:
        dp = dist[p]
        new = [math.inf] * len(row)
        left = math.inf
        for j in range(1, len(row)):
            up, diag = row[j], row[j - 1]
            best = up if up < diag else diag
            if left < best:
                best = left
            left = new[j] = dp[ref[j - 1]] + best
        row = new
    return row


def ndtw(dist, pred, ref, threshold=3.0):
    """exp(-DTW / (threshold * |ref|)) (r2r/eval_utils.py cal_dtw), dist[a][b] = shortest distance"""
    row = dtw_rows(dist, [0.0] + [math.inf] * len(ref), pred, ref)
    return math.exp(-row[-1] / (threshold * len(ref)))


class NavPlanner:
    """Host mirror of one rollout: owns the per-episode GraphMaps, emits one plan (dict of numpy arrays) per step."""

    def __init__(self, env, obs, feedback="teacher", max_action_len=15, expert_policy="spl", angle_table=None, train=True,
                 pad_V=0, pad_K=0, k_bucket=1):
        """pad_V / pad_K: minimum padded view / map-token counts (static shapes for a HIP-graph step: host/nav_graph.py); k_bucket: the
        map-token count of a step is rounded up to a multiple of it (shape buckets of the captured training steps: host/step_graphs.py);
        padded tokens are masked everywhere, so the valid outputs do not change."""
        self.pad_V, self.pad_K, self.k_bucket = pad_V, pad_K, max(1, int(k_bucket))
        self.env, self.obs = env, obs
        self.B = len(obs)
        # feedback: one mode for the batch, or one per episode -- an iteration's teacher-forced and DAgger rollouts
        # (agent_base.py:243-250) can then share every model call as ONE batch of 2B episodes (host/nav_rollout.py)
        self.fb = [feedback] * len(obs) if isinstance(feedback, str) else list(feedback)
        self.feedback, self.T, self.expert = feedback, max_action_len, expert_policy
        self.train = train
        self.angle_table = env.angle_table if angle_table is None else angle_table
        self.gmaps = [GraphMap(ob["viewpoint"]) for ob in obs]
        from . import hostplan
        self.native = None
        if hostplan.lib() is not None:
            # native per-sample loops (csrc/hostplan.c): the graph arrays sized once (no reallocation under the registered pointers in the common case)
            from .graph_map import FloydGraph
            for g in self.gmaps:
                g.graph = FloydGraph(cap=hostplan.CAP)
                g.pos_by_id = np.zeros((hostplan.CAP, 3), np.float64)
            self.native = hostplan.NativeBatch(self.gmaps)
        for i, (g, ob) in enumerate(zip(self.gmaps, obs)):
            self._update_graph(i, g, ob)
        self.traj = [dict(instr_id=ob["instr_id"], path=[[ob["viewpoint"]]]) for ob in obs]
        self.ended = np.zeros(self.B, bool)
        self.just_ended = np.zeros(self.B, bool)
        self.end_obs_vp = [None] * self.B          # viewpoint at the moment the episode ended (stop-node fix-up)
        self.stop_refs = [[] for _ in range(self.B)]     # (step, viewpoint) pairs whose stop probability was recorded
        self.fused_row = [dict() for _ in range(self.B)]   # visited viewpoint -> log row of its fused embedding
        self.view_rows = [dict() for _ in range(self.B)]   # unvisited viewpoint -> log rows of the views that showed it
        self.log_base, self.log_V = [], []         # per step: first log row, padded view count
        self.log_rows = 0
        self._pano_cache = getattr(env, "_pano_plan_cache", None)
        if self._pano_cache is None:
            self._pano_cache = {}
            try:
                env._pano_plan_cache = self._pano_cache
            except AttributeError:
                pass
        self._flat = {}                            # episode -> (flattened walked path, [segments consumed])
        self._dtw = {}                             # episode -> (nodes consumed, DTW row) of the walked path
        self._ref_idx = {}                         # episode -> its ground-truth path as indices of the scan's dense distance table
        self._pending = []                         # (episode, observation) pairs not yet entered into the graphs (end_step -> begin_nav)
        self.t = 0

    def _update_graph(self, i, g, ob):
        """GraphMap.update_graph(ob); with the native core the edge / relaxation loops run in C on the same arrays (csrc/hostplan.c mp_update_graph)"""
        nb = self.native
        if nb is None:
            return g.update_graph(ob)
        gr = g.graph
        vp = ob["viewpoint"]
        g.node_positions[vp] = ob["position"]
        cur = gr._id(vp)
        cid = []
        for cc in ob["candidate"]:
            g.node_positions[cc["viewpointId"]] = cc["position"]
            cid.append(gr._id(cc["viewpointId"]))
        n = len(gr.names)
        if n > g.pos_by_id.shape[0]:
            grown = np.zeros((max(2 * g.pos_by_id.shape[0], n), 3), np.float64)
            grown[:g.pos_by_id.shape[0]] = g.pos_by_id
            g.pos_by_id = grown
        g.pos_by_id[cur] = ob["position"]
        if cid:
            g.pos_by_id[cid] = [cc["position"] for cc in ob["candidate"]]
        if n > nb.step.shape[1]:
            nb.grow(n)              # (doubles up to csrc/hostplan.c's MP_MAXN = 512 nodes per episode; the Python planner has no bound)
        nb.register(i)
        import ctypes as C
        ca = (C.c_int32 * max(len(cid), 1))(*cid)
        from . import hostplan
        if hostplan.lib().mp_update_graph(nb.h, i, n, cur, ca, len(cid)) != 0:
            raise RuntimeError("mp_update_graph failed")

    def _dense_dist(self, scan, dist):
        """dense copy of env.shortest_distances[scan] for the native expert (built once per scan, kept on the env)"""
        cache = getattr(self.env, "_dense_dist_cache", None)
        if cache is None:
            cache = {}
            try:
                self.env._dense_dist_cache = cache
            except AttributeError:
                pass
        dd = cache.get(scan)
        if dd is None:
            from .hostplan import DenseDist
            dd = cache[scan] = DenseDist(dist)
        return dd


    # ---- language (agent.py:63-90) -----------------------------------------------------------------------------
    def language(self):
        lens = np.array([len(ob["instr_encoding"]) for ob in self.obs], np.int64)
        ids = np.zeros((self.B, int(lens.max())), np.int64)
        for i, ob in enumerate(self.obs):
            ids[i, :lens[i]] = ob["instr_encoding"]
        return dict(txt_ids=ids, txt_lens=lens)

    # ---- one step ----------------------------------------------------------------------------------------------
    def begin_step(self):
        """everything the three model calls of step t need, from the current observations"""
        p = self.begin_pano()
        p.update(self.begin_nav())
        return p

    def begin_pano(self):
        """first half of a step's plan: the panorama inputs (cheap, cached per viewpoint) -- the rollout launches the panorama
        encoder on them and builds the second half (`begin_nav`) while the GPU is busy"""
        t, B, obs, gmaps = self.t, self.B, self.obs, self.gmaps
        for i, g in enumerate(gmaps):
            if not self.ended[i]:
                g.node_step_ids[obs[i]["viewpoint"]] = t + 1          # agent.py:873-875
                if self.native is not None:
                    self.native.step[i, g.graph.index[obs[i]["viewpoint"]]] = t + 1
        # -- panorama inputs (agent.py:111-173): candidate views first, then the remaining views in index order
        # (static per (viewpoint, facing view): cached across steps and rollouts)
        cand_vpids, view_lens, per = [], np.zeros(B, np.int64), []
        for i, ob in enumerate(obs):
            key = (ob.get("row", (ob["scan"], ob["viewpoint"])), ob["viewIndex"])
            ent = self._pano_cache.get(key)
            if ent is None:
                cpts = [c["pointId"] for c in ob["candidate"]]
                used = set(cpts)
                order = np.array(cpts + [k for k in range(36) if k not in used], np.int32)
                nc, n = len(cpts), len(order)
                lf = np.ones((n, 7), np.float32)
                if nc:
                    ch = np.array([c["heading"] for c in ob["candidate"]], np.float64)
                    ce = np.array([c["elevation"] for c in ob["candidate"]], np.float64)
                    lf[:nc, :4] = np.stack([np.sin(ch), np.cos(ch), np.sin(ce), np.cos(ce)], 1).astype(np.float32)
                lf[nc:, :4] = self.angle_table[ob["viewIndex"]][order[nc:]]
                ent = self._pano_cache[key] = (order, nc, lf, [c["viewpointId"] for c in ob["candidate"]])
            per.append(ent)
            cand_vpids.append(ent[3])
            view_lens[i] = len(ent[0])
        V = max(int(view_lens.max()), self.pad_V)
        view_order = np.full((B, V), -1, np.int32)            # -1: padded slot, the gather writes zeros (pad_tensors, agent.py:155)
        nav_types = np.zeros((B, V), np.int64)
        loc = np.zeros((B, V, 7), np.float32)
        for i, (order, nc, lf, _) in enumerate(per):
            n = len(order)
            view_order[i, :n] = order
            nav_types[i, :nc] = 1
            loc[i, :n] = lf
        vp_rows = np.array([ob["row"] for ob in obs], np.int32)
        self._half = (V, cand_vpids, view_lens, nav_types)
        return dict(t=t, B=B, V=V, V_valid=int(view_lens.max()), vp_rows=vp_rows, view_order=view_order, loc_fts=loc, nav_types=nav_types, view_lens=view_lens,
                    cand_vpids=cand_vpids)

    def _begin_nav_native(self):
        """begin_nav with the per-sample loops in C (csrc/hostplan.c mp_plan_nav): same arrays, same entry order"""
        from . import hostplan
        t, B, obs, gmaps, nb = self.t, self.B, self.obs, self.gmaps, self.native
        V, cand_vpids, view_lens, nav_types = self._half
        base = self.log_rows
        self.log_base.append(base)
        self.log_V.append(V)
        self.log_rows += B * (V + 2)
        fused0, cls0 = base + B * V, base + B * V + B
        prev_cls0 = (self.log_base[t - 1] + B * self.log_V[t - 1] + B) if t > 0 else -1
        n_nodes = np.array([len(g.graph.names) for g in gmaps], np.int32)
        lens = n_nodes.astype(np.int64) + 2
        K = max((int(lens.max()) + self.k_bucket - 1) // self.k_bucket * self.k_bucket, self.pad_K)
        Vp = V + 2
        ci = np.array([g.graph.index[ob["viewpoint"]] for g, ob in zip(gmaps, obs)], np.int32)
        start = np.array([g.graph.index[g.start_vp] for g in gmaps], np.int32)
        flat, coff = [], [0]
        for g, cv in zip(gmaps, cand_vpids):
            gix = g.graph.index
            flat += [gix[c] for c in cv]
            coff.append(len(flat))
        cid = np.array(flat if flat else [0], np.int32)
        coff = np.array(coff, np.int32)
        ended = self.ended.astype(np.uint8)
        step_ids = np.zeros((B, K), np.int64)
        pair = np.zeros((B, K, K), np.float32)
        visited = np.zeros((B, K), np.uint8)
        fsrc = np.full((B, K), -1, np.int32)
        bw = np.zeros((B, Vp), np.uint8)
        order = np.zeros((B, K), np.int32)
        nv = np.zeros(B, np.int32)
        tot_cap = int(n_nodes.sum()) + B + len(flat)
        tp, gd, hp = np.empty((tot_cap, 3), np.float64), np.empty(tot_cap, np.float64), np.empty(tot_cap + 512, np.int64)
        seg_off = np.zeros(B + 1, np.int32)
        coo_cap = 2 * B + int(n_nodes.sum()) + int(nb.vcount.sum()) + len(flat) + 64      # (every recorded view row + this step's new ones)
        coo_o, coo_s, coo_w = np.empty(coo_cap, np.int64), np.empty(coo_cap, np.int64), np.empty(coo_cap, np.float32)
        ncoo = np.zeros(1, np.int32)
        for i in range(B):
            nb.register(i)
        if True:
            rc = hostplan.lib().mp_plan_nav(nb.h, t, K, V, n_nodes.ctypes.data, ended.ctypes.data, ci.ctypes.data, start.ctypes.data, cid.ctypes.data,
                                            coff.ctypes.data, base, fused0, prev_cls0, step_ids.ctypes.data, pair.ctypes.data, visited.ctypes.data,
                                            fsrc.ctypes.data, bw.ctypes.data, order.ctypes.data, nv.ctypes.data, tp.ctypes.data, gd.ctypes.data,
                                            hp.ctypes.data, seg_off.ctypes.data, coo_o.ctypes.data, coo_s.ctypes.data, coo_w.ctypes.data, coo_cap,
                                            ncoo.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"mp_plan_nav failed ({rc}): a map with more nodes than the step's token count, or more than {hostplan.VMAX} views of one node")
        visited = visited.view(np.bool_)
        gmask = np.arange(K)[None] < lens[:, None]
        gmask[:, 1] = False                                            # [mem] can never be chosen (agent.py:233)
        vpid_lists, no_left = [], np.zeros(B, bool)
        for i, g in enumerate(gmaps):
            gn, n = g.graph.names, int(n_nodes[i])
            vpid_lists.append([None, None] + [gn[k] for k in order[i, :n].tolist()])
            no_left[i] = int(nv[i]) == n
        tot = int(seg_off[B])
        cur_pos = np.array([g.node_positions[ob["viewpoint"]] for g, ob in zip(gmaps, obs)], np.float64).reshape(B, 3)
        cur_head = np.array([ob["heading"] for ob in obs], np.float64)
        rep = np.repeat(np.arange(B), np.diff(seg_off))
        h, e, d = rel_pos(cur_pos[rep], tp[:tot], base_heading=cur_head[rep], base_elevation=0)
        feats = np.concatenate([angle_fts(h.astype(np.float32), e.astype(np.float32)),
                                np.stack([d / MAX_DIST, gd[:tot] / MAX_DIST, hp[:tot] / MAX_STEP], 1).astype(np.float32)], 1)
        pos = np.zeros((B, K, 7), np.float32)
        vp_pos = np.zeros((B, Vp, 14), np.float32)
        pos[:, :2, 1] = pos[:, :2, 3] = 1.0                           # None tokens: angle (0, 0) -> cos = 1, distances 0
        for i in range(B):                                            # (agent.py:290-328 for the local tokens: start | candidate)
            o, ng = int(seg_off[i]), int(n_nodes[i])
            nc = int(coff[i + 1] - coff[i])
            pos[i, 2:2 + ng] = feats[o:o + ng]
            vp_pos[i, :, :7] = feats[o + ng]
            vp_pos[i, 2:2 + nc, 7:] = feats[o + ng + 1:o + ng + 1 + nc]
        m = int(ncoo[0])
        vo = (B * K + np.arange(B)[:, None] * Vp + 2 + np.arange(V)[None]).reshape(-1)
        vs = (base + np.arange(B)[:, None] * V + np.arange(V)[None]).reshape(-1)
        out_rows = np.concatenate([coo_o[:m], vo])
        src_rows = np.concatenate([coo_s[:m], vs])
        w = np.concatenate([coo_w[:m], np.ones(len(vo), np.float32)])
        n_out = B * K + B * Vp
        csr = _csr_from_coo(out_rows, src_rows, w, n_out)
        csr_t = _csr_from_coo(src_rows, out_rows, w, cls0) if self.train else None     # sources are rows < cls0
        vp_nav = np.concatenate([np.ones((B, 1), bool), np.zeros((B, 1), bool), nav_types == 1], 1)
        vp_masks = np.arange(Vp)[None] < (view_lens + 2)[:, None]
        vp_cand = [[None, None] + c for c in cand_vpids]
        targets = self._teacher_action(vpid_lists, visited)
        self._cur = dict(vpids=vpid_lists, no_left=no_left, targets=targets)
        return dict(K=K, Vp=Vp, K_valid=int(lens.max()), log_base=base, log_fused=fused0, log_cls=cls0, log_rows=self.log_rows,
                    gmap_vpids=vpid_lists, gmap_lens=lens, gmap_step_ids=step_ids, gmap_pos_fts=pos, gmap_pair_dists=pair,
                    gmap_visited_masks=visited, gmap_masks=gmask, no_vp_left=no_left,
                    vp_pos_fts=vp_pos, vp_nav_masks=vp_nav, vp_masks=vp_masks, vp_cand_vpids=vp_cand,
                    csr=csr, csr_t=csr_t, n_out=n_out, fsrc=fsrc, bw=bw, targets=targets)

    def _flush_graph_updates(self):
        pend, self._pending = self._pending, []
        for i, ob in pend:
            self._update_graph(i, self.gmaps[i], ob)

    def begin_nav(self):
        """second half: map / local tokens, embedding sources, fusion map, expert action"""
        self._flush_graph_updates()
        if self.native is not None:
            return self._begin_nav_native()
        t, B, obs, gmaps = self.t, self.B, self.obs, self.gmaps
        V, cand_vpids, view_lens, nav_types = self._half
        # -- log layout of this step: B*V view rows, B fused rows, B [cls] rows
        base = self.log_rows
        self.log_base.append(base)
        self.log_V.append(V)
        self.log_rows += B * (V + 2)
        fused0, cls0 = base + B * V, base + B * V + B
        prev_cls0 = (self.log_base[t - 1] + B * self.log_V[t - 1] + B) if t > 0 else -1
        # -- node-embedding bookkeeping (agent.py:905-924)
        for i, g in enumerate(gmaps):
            if self.ended[i]:
                continue
            cur = obs[i]["viewpoint"]
            self.fused_row[i][cur] = fused0 + i
            self.view_rows[i].pop(cur, None)
            for j, cv in enumerate(cand_vpids[i]):
                if not g.graph.visited(cv):
                    self.view_rows[i].setdefault(cv, []).append(base + i * V + j)
        # -- map tokens (agent.py:175-251)
        vpid_lists, lens = [], np.zeros(B, np.int64)
        node_ids = []
        no_left = np.zeros(B, bool)
        for i, g in enumerate(gmaps):
            names = list(g.node_positions.keys())
            seen = [g.graph.visited(k) for k in names]
            vis = [k for k, s in zip(names, seen) if s]
            unv = [k for k, s in zip(names, seen) if not s]
            no_left[i] = len(unv) == 0
            vpid_lists.append([None, None] + vis + unv)
            node_ids.append((len(vis), len(unv)))
            lens[i] = 2 + len(vis) + len(unv)
        K = max((int(lens.max()) + self.k_bucket - 1) // self.k_bucket * self.k_bucket, self.pad_K)
        Vp = V + 2
        step_ids = np.zeros((B, K), np.int64)
        pos = np.zeros((B, K, 7), np.float32)
        pair = np.zeros((B, K, K), np.float32)
        visited = np.zeros((B, K), bool)
        gmask = np.arange(K)[None] < lens[:, None]
        gmask[:, 1] = False                                            # [mem] can never be chosen (agent.py:233)
        vp_pos = np.zeros((B, Vp, 14), np.float32)
        coo_o, coo_s, coo_w = [], [], []
        # positions: the targets of ALL samples (map nodes, start viewpoint, candidates) go through ONE vectorised
        # relative-position / angle-feature evaluation; graph distances and hop counts are per-sample row gathers
        tp, gd, hp, rep, segs = [], [], [], [], []
        cur_pos, cur_head = np.zeros((B, 3)), np.zeros(B)
        for i, g in enumerate(gmaps):
            ids = vpid_lists[i]
            n, (nv, nu) = int(lens[i]), node_ids[i]
            visited[i, 1:2 + nv] = True
            sid, gix = g.node_step_ids, g.graph.index
            step_ids[i, 2:n] = [sid.get(vp, 0) for vp in ids[2:]]
            gi = [gix[vp] for vp in ids[2:]]
            cur = obs[i]["viewpoint"]
            ci = gix[cur]
            nc = len(cand_vpids[i])
            idx_all = np.array(gi + [gix[g.start_vp]] + [gix[vp] for vp in cand_vpids[i]], np.int64)
            tp.append(g.pos_by_id[idx_all])
            gd.append(g.graph.dist_row(ci)[idx_all])
            hp.append(g.graph.hops(ci)[idx_all])
            rep.append(np.full(len(idx_all), i))
            segs.append((n - 2, nc))
            cur_pos[i], cur_head[i] = g.node_positions[cur], obs[i]["heading"]
            gia = idx_all[:n - 2]
            sub = g.graph._d[np.ix_(gia, gia)].astype(np.float32)
            np.fill_diagonal(sub, 0.0)
            pair[i, 2:n, 2:n] = sub
            # embedding sources
            if t > 0:
                coo_o += [i * K + 1, B * K + i * Vp + 1]
                coo_s += [prev_cls0 + i, prev_cls0 + i]
                coo_w += [1.0, 1.0]
            fr, vr = self.fused_row[i], self.view_rows[i]
            for k, vp in enumerate(ids[2:], start=2):
                if k < 2 + nv:
                    coo_o.append(i * K + k)
                    coo_s.append(fr[vp])
                    coo_w.append(1.0)
                else:
                    rows = vr[vp]
                    coo_o += [i * K + k] * len(rows)
                    coo_s += rows
                    coo_w += [1.0 / len(rows)] * len(rows)
        rep = np.concatenate(rep)
        h, e, d = rel_pos(cur_pos[rep], np.concatenate(tp), base_heading=cur_head[rep], base_elevation=0)
        feats = np.concatenate([angle_fts(h.astype(np.float32), e.astype(np.float32)),
                                np.stack([d / MAX_DIST, np.concatenate(gd) / MAX_DIST, np.concatenate(hp) / MAX_STEP], 1).astype(np.float32)], 1)
        pos[:, :2, 1] = pos[:, :2, 3] = 1.0                           # None tokens: angle (0, 0) -> cos = 1, distances 0
        o = 0
        for i, (ng, nc) in enumerate(segs):                           # (agent.py:290-328 for the local tokens: start | candidate)
            pos[i, 2:2 + ng] = feats[o:o + ng]
            vp_pos[i, :, :7] = feats[o + ng]
            vp_pos[i, 2:2 + nc, 7:] = feats[o + ng + 1:o + ng + 1 + nc]
            o += ng + 1 + nc
        # views of the current panorama -> local tokens 2..V+1 (padded rows included, like torch.cat in :294-297)
        vo = (B * K + np.arange(B)[:, None] * Vp + 2 + np.arange(V)[None]).reshape(-1)
        vs = (base + np.arange(B)[:, None] * V + np.arange(V)[None]).reshape(-1)
        out_rows = np.concatenate([np.array(coo_o, np.int64), vo])
        src_rows = np.concatenate([np.array(coo_s, np.int64), vs])
        w = np.concatenate([np.array(coo_w, np.float32), np.ones(len(vo), np.float32)])
        n_out = B * K + B * Vp
        csr = _csr_from_coo(out_rows, src_rows, w, n_out)
        csr_t = _csr_from_coo(src_rows, out_rows, w, cls0) if self.train else None     # sources are rows < cls0
        vp_nav = np.concatenate([np.ones((B, 1), bool), np.zeros((B, 1), bool), nav_types == 1], 1)
        vp_masks = np.arange(Vp)[None] < (view_lens + 2)[:, None]
        vp_cand = [[None, None] + c for c in cand_vpids]
        fsrc, bw = fusion_map(vpid_lists, visited, vp_cand, K, Vp)
        targets = self._teacher_action(vpid_lists, visited)
        self._cur = dict(vpids=vpid_lists, no_left=no_left, targets=targets)
        return dict(K=K, Vp=Vp, K_valid=int(lens.max()), log_base=base, log_fused=fused0, log_cls=cls0, log_rows=self.log_rows,
                    gmap_vpids=vpid_lists, gmap_lens=lens, gmap_step_ids=step_ids, gmap_pos_fts=pos, gmap_pair_dists=pair,
                    gmap_visited_masks=visited, gmap_masks=gmask, no_vp_left=no_left,
                    vp_pos_fts=vp_pos, vp_nav_masks=vp_nav, vp_masks=vp_masks, vp_cand_vpids=vp_cand,
                    csr=csr, csr_t=csr_t, n_out=n_out, fsrc=fsrc, bw=bw, targets=targets)

    # ---- expert (agent.py:330-373) -----------------------------------------------------------------------------
    def _teacher_action(self, vpids, visited):
        a = np.zeros(self.B, np.int64)
        t, env = self.t, self.env
        for i, ob in enumerate(self.obs):
            if self.ended[i]:
                a[i] = IGNORE
            elif self.fb[i] == "teacher":
                if ob["viewpoint"] != ob["gt_path"][t]:
                    raise AssertionError("teacher forcing left the ground-truth path")
                if t < len(ob["gt_path"]) - 1:
                    goal = ob["gt_path"][t + 1]
                    a[i] = next((j for j, v in enumerate(vpids[i]) if v == goal), 0)
            elif ob["viewpoint"] != ob["gt_path"][-1]:
                scan, cur = ob["scan"], ob["viewpoint"]
                best, best_d = IGNORE, float("inf")
                dist, gt = env.shortest_distances[scan], ob["gt_path"]
                if self.expert == "ndtw":
                    from . import hostplan
                    segs = self.traj[i]["path"]                      # the walked path, flattened incrementally (was sum(segs, []): quadratic in the episode's length)
                    walked, k = self._flat.setdefault(i, ([], [0]))
                    for p in segs[k[0]:]:
                        walked.extend(p)
                    k[0] = len(segs)
                    n0, row = self._dtw.get(i, (0, [0.0] + [math.inf] * len(gt)))
                    if hostplan.lib() is not None:
                        # native rows (csrc/hostplan.c: the same recurrence, the same order of additions and comparisons): the walked prefix, then
                        # every unvisited candidate's connecting path in one call
                        dd = self._dense_dist(scan, dist)
                        ref = self._ref_idx.get(i)
                        if ref is None:
                            ref = self._ref_idx[i] = np.array([dd.index[v] for v in gt], np.int32)
                        if len(walked) > n0:
                            row = hostplan.dtw_extend(dd, row, walked[n0:], ref)
                        self._dtw[i] = (len(walked), row)
                        cj = [j for j, v in enumerate(vpids[i]) if j > 1 and not visited[i, j]]
                        if cj:
                            sp, pc, ix, arrs = env.shortest_paths[scan][cur], dd.paths, dd.index, []
                            for j in cj:
                                v = vpids[i][j]
                                pa = pc.get((cur, v))
                                if pa is None:
                                    pa = pc[(cur, v)] = np.array([ix[x] for x in sp[v][1:]], np.int32)
                                arrs.append(pa)
                            last = hostplan.dtw_cands_idx(dd, row, arrs, ref)
                            for j, t_last in zip(cj, last):
                                d = -math.exp(-float(t_last) / (3.0 * len(gt)))
                                if d < best_d:
                                    best, best_d = j, d
                        a[i] = best
                        continue
                    row = dtw_rows(dist, row, walked[n0:], gt)
                    self._dtw[i] = (len(walked), row)
                for j, v in enumerate(vpids[i]):
                    if j > 1 and not visited[i, j]:
                        if self.expert == "ndtw":
                            tail = dtw_rows(dist, row, env.shortest_paths[scan][cur][v][1:], gt)
                            d = -math.exp(-tail[-1] / (3.0 * len(gt)))
                        else:
                            d = dist[v][gt[-1]] + dist[cur][v]
                        if d < best_d:
                            best, best_d = j, d
                a[i] = best
        return a

    # ---- act + observe (agent.py:1056-1107) ----------------------------------------------------------------------
    def end_step(self, a_t=None, features=False):
        """a_t: chosen map-token index per sample (None under teacher forcing = the expert's).  Returns True when all ended."""
        t, B, obs = self.t, self.B, self.obs
        cur = self._cur
        a_t = cur["targets"] if a_t is None else [cur["targets"][i] if self.fb[i] == "teacher" else a_t[i] for i in range(B)]
        for i in range(B):
            if not self.ended[i]:
                self.stop_refs[i].append((t, obs[i]["viewpoint"]))
        stop = [(ob["viewpoint"] == ob["gt_path"][-1]) if self.fb[i] in ("teacher", "sample") else int(a_t[i]) == 0
                for i, ob in enumerate(obs)]
        acts, hops_from = [], [None] * B
        for i in range(B):
            if stop[i] or self.ended[i] or cur["no_left"][i] or t == self.T - 1:
                acts.append(None)
                self.just_ended[i] = True
            else:
                acts.append(cur["vpids"][i][int(a_t[i])])
        for i, ob in enumerate(obs):
            if acts[i] is not None:
                p = self.gmaps[i].graph.path(ob["viewpoint"], acts[i])
                self.traj[i]["path"].append(p)
                hops_from[i] = self.traj[i]["path"][-2][-1] if len(p) == 1 else p[-2]
        self.env.step(acts, hops_from)
        for i in range(B):
            if not self.ended[i] and self.just_ended[i]:
                self.end_obs_vp[i] = obs[i]["viewpoint"]
        self.obs = obs = self.env._get_obs(features)
        # the new observations enter the episodes' graphs at the start of `begin_nav` (or `finish`), not here: `begin_pano` does not read the graphs, and
        # the GPU idles from the moment the actions arrive until the next step's panorama launch -- the edge / relaxation work waits under that launch
        self._pending = [(i, ob) for i, ob in enumerate(obs) if not self.ended[i]]
        if not DEFER_GRAPH:
            self._flush_graph_updates()
        self.ended |= np.array([a is None for a in acts])
        self.actions = acts
        self.t += 1
        return bool(self.ended.all())

    def finish(self, stop_probs):
        """stop_probs[t][b] = softmax(fused_logits)[b, 0] of every step: back-track each episode to its most stop-worthy
        visited node (agent.py:1080-1089; the reference reads the scores with .item() every step -- deferred here, the
        episode's graph does not change after it ended)"""
        self._flush_graph_updates()
        for i in range(self.B):
            if self.end_obs_vp[i] is None:
                continue
            node, best = None, -float("inf")
            for (t, vp) in self.stop_refs[i]:
                s = float(stop_probs[t][i])
                self.gmaps[i].node_stop_scores[vp] = {"stop": s}
            for vp, v in self.gmaps[i].node_stop_scores.items():
                if v["stop"] > best:
                    node, best = vp, v["stop"]
            if node is not None and self.end_obs_vp[i] != node:
                self.traj[i]["path"].append(self.gmaps[i].graph.path(self.end_obs_vp[i], node))
        return self.traj


def fusion_map(gmap_vpids, visited_masks, vp_cand_vpids, K, Vp):
    """local -> global logit fusion as index arrays ([LINEAGE] DUET navigation forward): map token k of an
    unvisited node receives the local logit of the view that shows it (fsrc = that local token), or, when no current view
    shows it, the sum of the local logits of views that lead back to visited nodes (fsrc = -2, bw marks those views)."""
    B = len(gmap_vpids)
    fsrc = np.full((B, K), -1, np.int32)
    bw = np.zeros((B, Vp), np.uint8)
    for b in range(B):
        ids = gmap_vpids[b]
        seen = set(vp for j, vp in enumerate(ids) if vp is not None and visited_masks[b][j])
        shown = {}
        for j, c in enumerate(vp_cand_vpids[b]):
            if c is None or j == 0:
                continue
            if c in seen:
                bw[b, j] = 1
            else:
                shown[c] = j
        fsrc[b, 0] = 0
        for j, vp in enumerate(ids):
            if j > 0 and vp is not None and vp not in seen:
                fsrc[b, j] = shown.get(vp, -2)
    return fsrc, bw
