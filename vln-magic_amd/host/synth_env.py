"""Synthetic stand-in for the reference's `R2RNavBatch` + MatterSim (map_nav_src/r2r/env.py:87-400): random building-like
connectivity graphs, seeded 36-view x 768-d features, instructions of random token ids and shortest-path ground truth.
There is no simulator, dataset or network in this environment, so the navigator loop (BASELINE config 5) is
driven by this stepper; it produces observations with exactly the keys and conventions the agent reads (env.py:337-368):

  viewpoint / scan / position / heading / elevation / viewIndex   discretised 12 x 3 viewing angles
  feature   [36, 768 + 4]   view features + angle features relative to the agent's view (env.py:345, utils/data.py:129-151)
  candidate [{viewpointId, pointId, position, heading, elevation, feature[768+4], ...}]   one per neighbour, represented by
            the view closest to it (env.py:283-300), heading / elevation relative to the agent's view (env.py:317-318)
  instr_encoding, gt_path, instr_id, path_id

Movement follows `make_equiv_action` (agent.py:375-403): the agent lands on the chosen viewpoint facing the discretised
direction of the last hop.  `features=False` observations skip the per-observation host feature copies: the index-plan
rollout reads features from the HBM-resident table (`feature_table`, row = `vp_row[scan, viewpoint]`) instead.
"""
import math

import numpy as np

R30 = math.radians(30)


def _angle_feature(h, e):
    return np.array([math.sin(h), math.cos(h), math.sin(e), math.cos(e)], np.float32)


def view_angle_table():
    """[base view 36][view 36][4]: angle feature of every view relative to the agent's view (utils/data.py:129-154)"""
    t = np.zeros((36, 36, 4), np.float32)
    for base in range(36):
        bh, be = (base % 12) * R30, (base // 12 - 1) * R30
        for ix in range(36):
            t[base, ix] = _angle_feature((ix % 12) * R30 - bh, (ix // 12 - 1) * R30 - be)
    return t


class Scan:
    def __init__(self, name, rng, n_nodes, spacing=2.2):
        self.name = name
        side = int(math.ceil(math.sqrt(n_nodes)))
        cells = [(i, j) for i in range(side) for j in range(side)]
        pick = rng.permutation(len(cells))[:n_nodes]
        pos = np.array([[cells[c][0] * spacing, cells[c][1] * spacing, 0.0] for c in pick]) + \
            np.concatenate([rng.uniform(-0.5, 0.5, (n_nodes, 2)), rng.normal(0, 0.12, (n_nodes, 1))], 1)
        self.pos = pos
        self.vps = [f"{name}_{i:03d}" for i in range(n_nodes)]
        self.idx = {v: i for i, v in enumerate(self.vps)}
        d = np.sqrt(((pos[:, None] - pos[None]) ** 2).sum(-1))
        adj = (d < 1.55 * spacing) & ~np.eye(n_nodes, dtype=bool)
        # connect the components through their closest pairs
        comp = np.arange(n_nodes)

        def find(x):
            while comp[x] != x:
                comp[x] = comp[comp[x]]
                x = comp[x]
            return x
        for a, b in zip(*np.nonzero(adj)):
            comp[find(a)] = find(b)
        order = np.dstack(np.unravel_index(np.argsort(d, axis=None), d.shape))[0]
        for a, b in order:
            if a < b and find(a) != find(b):
                adj[a, b] = adj[b, a] = True
                comp[find(a)] = find(b)
        self.adj = adj
        # all-pairs shortest paths (Floyd-Warshall with next-hop table)
        sd = np.where(adj, d, np.inf)
        np.fill_diagonal(sd, 0.0)
        nxt = np.where(adj, np.arange(n_nodes)[None, :], -1)
        for k in range(n_nodes):
            via = sd[:, k][:, None] + sd[k, :][None, :]
            better = via < sd
            sd = np.where(better, via, sd)
            nxt = np.where(better, nxt[:, k][:, None], nxt)
        self.sdist, self.nxt = sd, nxt
        self.hops = np.zeros((n_nodes, n_nodes), np.int64)          # hop count along the shortest-distance path
        for a in range(n_nodes):
            for b in range(n_nodes):
                self.hops[a, b] = len(self.path(a, b)) - 1
        # candidates of every viewpoint: neighbour -> (absolute heading, elevation, nearest view)
        self.cands = []
        for a in range(n_nodes):
            cl = []
            for b in np.nonzero(adj[a])[0]:
                dx, dy, dz = pos[b] - pos[a]
                xy = max(math.hypot(dx, dy), 1e-8)
                h = math.asin(dx / xy)
                if pos[b][1] < pos[a][1]:
                    h = math.pi - h
                h = h % (2 * math.pi)
                e = math.asin(dz / max(math.sqrt(dx * dx + dy * dy + dz * dz), 1e-8))
                hb = int(round(h / R30)) % 12
                eb = 0 if e < -R30 / 2 else (2 if e > R30 / 2 else 1)
                cl.append(dict(viewpointId=self.vps[b], pointId=12 * eb + hb, normalized_heading=h, normalized_elevation=e,
                               position=tuple(float(x) for x in pos[b]), scanId=name, idx=len(cl) + 1))
            self.cands.append(cl)

    def path(self, a, b):
        out = [a]
        while a != b:
            a = int(self.nxt[a, b])
            out.append(a)
        return out


def _tables(sc):
    """env.shortest_distances[scan][a][b] / env.shortest_paths[scan][a][b] as the reference keeps them: dicts of dicts
    (r2r/env.py:115-119, networkx all-pairs results)"""
    n = len(sc.vps)
    dist = {sc.vps[a]: {sc.vps[b]: float(sc.sdist[a, b]) for b in range(n)} for a in range(n)}
    paths = {sc.vps[a]: {sc.vps[b]: [sc.vps[i] for i in sc.path(a, b)] for b in range(n)} for a in range(n)}
    return dist, paths


class SynthNavEnv:
    def __init__(self, batch_size=8, n_scans=3, nodes_per_scan=40, feat_dim=768, path_hops=(3, 6), instr_len=(20, 80),
                 vocab=(3, 50264), seed=0):
        self.batch_size, self.feat_dim = batch_size, feat_dim
        self.path_hops, self.instr_len, self.vocab = path_hops, instr_len, vocab
        self.rng = np.random.default_rng(seed)
        self.scans = {}
        rows = 0
        self.vp_row = {}
        for s in range(n_scans):
            sc = Scan(f"s{s}", self.rng, nodes_per_scan)
            self.scans[sc.name] = sc
            for v in sc.vps:
                self.vp_row[(sc.name, v)] = rows
                rows += 1
        self.feature_table = self.rng.standard_normal((rows, 36, feat_dim), dtype=np.float32)
        self.angle_table = view_angle_table()
        self.shortest_distances, self.shortest_paths = {}, {}
        for n, sc in self.scans.items():
            self.shortest_distances[n], self.shortest_paths[n] = _tables(sc)
        self.batch, self.state = None, None
        self._n_ep = 0

    # ---- episodes ---------------------------------------------------------------------------------------------
    def _draw_episode(self):
        rng = self.rng
        names = list(self.scans)
        while True:
            sc = self.scans[names[int(rng.integers(len(names)))]]
            a = int(rng.integers(len(sc.vps)))
            hops = int(rng.integers(self.path_hops[0], self.path_hops[1] + 1))
            ends = np.nonzero(sc.hops[a] == hops)[0]
            if len(ends):
                b = int(ends[int(rng.integers(len(ends)))])
                break
        n = int(rng.integers(self.instr_len[0], self.instr_len[1] + 1))
        ids = [0] + [int(x) for x in rng.integers(self.vocab[0], self.vocab[1] + 1, n - 2)] + [2]
        self._n_ep += 1
        return dict(instr_id=f"ep{self._n_ep}", path_id=self._n_ep, scan=sc.name, path=[sc.vps[i] for i in sc.path(a, b)],
                    heading=float(int(rng.integers(12)) * R30), instr_encoding=ids, instruction="")

    def reset(self, batch=None, features=True):
        self.batch = batch if batch is not None else [self._draw_episode() for _ in range(self.batch_size)]
        self.state = [dict(scan=it["scan"], vp=it["path"][0], heading=it["heading"], elevation=0.0) for it in self.batch]
        return self._get_obs(features)

    def new_episode(self, i, vp, heading, elevation):
        """sims.newEpisode for one slot (agent.py:403): the discretised simulator snaps to the 12 x 3 viewing angles"""
        hb = int(round((heading % (2 * math.pi)) / R30)) % 12
        eb = min(2, max(0, int(round(elevation / R30)) + 1))
        self.state[i].update(vp=vp, heading=hb * R30, elevation=(eb - 1) * R30)

    def step(self, targets, hops_from):
        """move slot i to targets[i] (None = stay); hops_from[i] = the viewpoint the last hop starts from (agent.py:386-391)"""
        for i, vp in enumerate(targets):
            if vp is None:
                continue
            st = self.state[i]
            sc = self.scans[st["scan"]]
            vi = next(c["pointId"] for c in sc.cands[sc.idx[hops_from[i]]] if c["viewpointId"] == vp)
            self.new_episode(i, vp, (vi % 12) * R30, (vi // 12 - 1) * R30)

    # ---- observations -----------------------------------------------------------------------------------------
    def view_index(self, i):
        st = self.state[i]
        return 12 * (int(round(st["elevation"] / R30)) + 1) + int(round(st["heading"] / R30)) % 12

    def _get_obs(self, features=True):
        obs = []
        for i, (it, st) in enumerate(zip(self.batch, self.state)):
            sc = self.scans[st["scan"]]
            a = sc.idx[st["vp"]]
            base = self.view_index(i)
            bh, be = (base % 12) * R30, (base // 12 - 1) * R30
            feat = self.feature_table[self.vp_row[(sc.name, st["vp"])]] if features else None
            cands = []
            for c in sc.cands[a]:
                cc = dict(c)
                cc["heading"] = c["normalized_heading"] - bh
                cc["elevation"] = c["normalized_elevation"] - be
                if features:
                    cc["feature"] = np.concatenate([feat[c["pointId"]], _angle_feature(cc["heading"], cc["elevation"])], -1)
                cands.append(cc)
            ob = dict(instr_id=it["instr_id"], path_id=it["path_id"], scan=sc.name, viewpoint=st["vp"], viewIndex=base,
                      position=tuple(float(x) for x in sc.pos[a]), heading=st["heading"], elevation=st["elevation"],
                      candidate=cands, instr_encoding=it["instr_encoding"], gt_path=it["path"], instruction=it["instruction"],
                      distance=float(sc.sdist[a, sc.idx[it["path"][-1]]]), row=self.vp_row[(sc.name, st["vp"])])
            if features:
                ob["feature"] = np.concatenate([feat, self.angle_table[base]], -1)
            obs.append(ob)
        return obs
