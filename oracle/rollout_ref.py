"""ORACLE (test infrastructure): the navigator step loop of `GMapNavAgent.rollout`, restated with the reference's own
per-sample host loops (map_nav_src/r2r/agent.py):

  language_variable        :63-90      panorama_variable      :111-173 (candidate views first, then the other views)
  nav_gmap_variable        :175-251    nav_vp_variable_mem    :290-328
  teacher_action           :330-373    (imitation learning, and the 'spl' / 'ndtw' expert of the DAgger rollouts)
  make_equiv_action        :375-403    rollout                :722-1160 (feedback teacher | argmax | sample; MAKD t2s)

and a dict-of-dict `RefFloyd` (speaker_utils.py:501-546) / `RefGraphMap` ([LINEAGE] DUET models/graph_utils.py -- the
reference withholds it; API from the call sites listed in vln-magic_amd/host/graph_map.py).

PINNING: RefFloyd, and the four *_variable builders + teacher_action driven with RefGraphMap, are checked against the
reference's own FloydGraph / GMapNavAgent methods by tests/golden/nav_loop.pt (minted by tests/golden/mint_golden.py).  The
loop as a whole runs against MatterSim in the reference and cannot run here: parity of the full rollout is unpinned.
The model is a parameter (any object with the VLNBert call contract), so the same loop drives the CPU oracle model
(oracle/nav_ref.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math
from collections import defaultdict

import numpy as np
import torch
import torch.nn.functional as F

from . import makd_ref as M

MAX_DIST, MAX_STEP = 30, 10                                  # r2r/env.py:22-23
IGNORE = -100                                                # r2r/parser.py:36


class RefFloyd:
    def __init__(self):
        big = 95959595
        self.dis = defaultdict(lambda: defaultdict(lambda: big))
        self.mid = defaultdict(lambda: defaultdict(lambda: ""))
        self.done = set()

    def distance(self, x, y):
        return 0 if x == y else self.dis[x][y]

    def add_edge(self, x, y, d):
        if d < self.dis[x][y]:
            self.dis[x][y] = self.dis[y][x] = d
            self.mid[x][y] = self.mid[y][x] = ""

    def update(self, k):
        for x in self.dis:
            for y in self.dis:
                if x != y and self.dis[x][k] + self.dis[k][y] < self.dis[x][y]:
                    self.dis[x][y] = self.dis[y][x] = self.dis[x][k] + self.dis[k][y]
                    self.mid[x][y] = self.mid[y][x] = k
        self.done.add(k)

    def visited(self, k):
        return k in self.done

    def path(self, x, y):
        if x == y:
            return []
        k = self.mid[x][y]
        return [y] if k == "" else self.path(x, k) + self.path(k, y)


def rel_pos_fts(a, b, base_heading=0, base_elevation=0):     # utils/data.py:157-174
    dx, dy, dz = b[0] - a[0], b[1] - a[1], b[2] - a[2]
    xy = max(np.sqrt(dx ** 2 + dy ** 2), 1e-8)
    xyz = max(np.sqrt(dx ** 2 + dy ** 2 + dz ** 2), 1e-8)
    heading = np.arcsin(dx / xy)
    if b[1] < a[1]:
        heading = np.pi - heading
    return heading - base_heading, np.arcsin(dz / xyz) - base_elevation, xyz


def angle_fts(h, e, size=4):                                 # utils/data.py:176-182
    f = np.vstack([np.sin(h), np.cos(h), np.sin(e), np.cos(e)]).transpose().astype(np.float32)
    return np.concatenate([f] * (size // 4), 1) if size // 4 > 1 else f


class RefGraphMap:
    def __init__(self, start_vp):
        self.start_vp = start_vp
        self.node_positions, self.graph = {}, RefFloyd()
        self.embeds = {False: {}, True: {}}
        self.node_stop_scores, self.node_step_ids = {}, {}

    def update_graph(self, ob):
        self.node_positions[ob["viewpoint"]] = ob["position"]
        for cc in ob["candidate"]:
            self.node_positions[cc["viewpointId"]] = cc["position"]
            d = math.sqrt(sum((p - q) ** 2 for p, q in zip(ob["position"], cc["position"])))
            self.graph.add_edge(ob["viewpoint"], cc["viewpointId"], d)
        self.graph.update(ob["viewpoint"])

    def update_node_embed(self, vp, embed, rewrite=False, teacher=False):
        st = self.embeds[bool(teacher)]
        if rewrite or vp not in st:
            st[vp] = [embed, 1]
        else:
            st[vp] = [st[vp][0] + embed, st[vp][1] + 1]

    def get_node_embed(self, vp, teacher=False):
        e, n = self.embeds[bool(teacher)][vp]
        return e / n

    def get_pos_fts(self, cur_vp, vpids, cur_heading, cur_elevation, angle_feat_size=4):
        ang, dist = [], []
        for vp in vpids:
            if vp is None:
                ang.append([0, 0])
                dist.append([0, 0, 0])
            else:
                h, e, d = rel_pos_fts(self.node_positions[cur_vp], self.node_positions[vp], base_heading=cur_heading, base_elevation=0)
                ang.append([h, e])
                dist.append([d / MAX_DIST, self.graph.distance(cur_vp, vp) / MAX_DIST, len(self.graph.path(cur_vp, vp)) / MAX_STEP])
        ang, dist = np.array(ang).astype(np.float32), np.array(dist).astype(np.float32)
        return np.concatenate([angle_fts(ang[:, 0], ang[:, 1], angle_feat_size), dist], 1)


# ---- input assembly ------------------------------------------------------------------------------------------------
def pad_rows(ts, pad=0):
    n = max(t.shape[0] for t in ts)
    out = torch.full((len(ts), n) + tuple(ts[0].shape[1:]), pad, dtype=ts[0].dtype, device=ts[0].device)
    for i, t in enumerate(ts):
        out[i, :t.shape[0]] = t
    return out


def seq_masks(lens, n=None):
    lens = torch.as_tensor(lens)
    n = int(lens.max()) if n is None else n
    return torch.arange(n)[None] < lens[:, None]


def language_variable(obs):
    lens = [len(ob["instr_encoding"]) for ob in obs]
    ids = np.zeros((len(obs), max(lens)), dtype=np.int64)
    for i, ob in enumerate(obs):
        ids[i, :lens[i]] = ob["instr_encoding"]
    return dict(txt_ids=torch.from_numpy(ids), txt_masks=seq_masks(lens, ids.shape[1]))


def panorama_variable(obs, feat=768):
    img, loc, nav, lens, cands = [], [], [], [], []
    for ob in obs:
        vi, va, nt, cv, used = [], [], [], [], set()
        for cc in ob["candidate"]:
            vi.append(cc["feature"][:feat])
            va.append(cc["feature"][feat:])
            nt.append(1)
            cv.append(cc["viewpointId"])
            used.add(cc["pointId"])
        vi.extend(x[:feat] for k, x in enumerate(ob["feature"]) if k not in used)
        va.extend(x[feat:] for k, x in enumerate(ob["feature"]) if k not in used)
        nt.extend([0] * (36 - len(used)))
        vi, va = np.stack(vi, 0), np.stack(va, 0)
        img.append(torch.from_numpy(vi))
        loc.append(torch.from_numpy(np.concatenate([va, np.ones((len(vi), 3), np.float32)], 1)))
        nav.append(torch.LongTensor(nt))
        cands.append(cv)
        lens.append(len(vi))
    return dict(view_img_fts=pad_rows(img), loc_fts=pad_rows(loc), nav_types=pad_rows(nav), view_lens=torch.LongTensor(lens),
                cand_vpids=cands, already_dropout=False)


def nav_gmap_variable(obs, gmaps, last_embeds=None, teacher=False):
    B = len(obs)
    vpids_all, lens, embeds, steps, pos, dists, vis, no_left = [], [], [], [], [], [], [], []
    for i, gmap in enumerate(gmaps):
        visited = [k for k in gmap.node_positions.keys() if gmap.graph.visited(k)]
        unvisited = [k for k in gmap.node_positions.keys() if not gmap.graph.visited(k)]
        no_left.append(len(unvisited) == 0)
        vpids = [None, None] + visited + unvisited              # stop, memory, visited, unvisited (enc_full_graph)
        vmask = [0, 1] + [1] * len(visited) + [0] * len(unvisited)
        sid = [gmap.node_step_ids.get(vp, 0) for vp in vpids]
        node = [gmap.get_node_embed(vp, teacher) for vp in vpids[2:]]
        mem = torch.zeros_like(node[0]) if last_embeds is None else last_embeds[i]
        embeds.append(torch.stack([torch.zeros_like(node[0]), mem] + node, 0))
        pos.append(torch.from_numpy(gmap.get_pos_fts(obs[i]["viewpoint"], vpids, obs[i]["heading"], obs[i]["elevation"])))
        pd = np.zeros((len(vpids), len(vpids)), dtype=np.float32)
        for a in range(2, len(vpids)):
            for b in range(a + 1, len(vpids)):
                pd[a, b] = pd[b, a] = gmap.graph.distance(vpids[a], vpids[b])
        dists.append(torch.from_numpy(pd))
        steps.append(torch.LongTensor(sid))
        vis.append(torch.BoolTensor(vmask))
        vpids_all.append(vpids)
        lens.append(len(vpids))
    masks = seq_masks(lens)
    masks[:, 1] = False                                          # the memory token is never a target
    K = max(lens)
    pair = torch.zeros(B, K, K)
    for i in range(B):
        pair[i, :lens[i], :lens[i]] = dists[i]
    return dict(gmap_vpids=vpids_all, gmap_img_embeds=pad_rows(embeds), gmap_step_ids=pad_rows(steps), gmap_pos_fts=pad_rows(pos),
                gmap_visited_masks=pad_rows(vis), gmap_pair_dists=pair, gmap_masks=masks, no_vp_left=no_left)


def nav_vp_variable_mem(obs, gmaps, pano_embeds, cand_vpids, view_lens, nav_types, last_embeds=None):
    B = len(obs)
    mem = torch.zeros_like(pano_embeds[:, :1]) if last_embeds is None else last_embeds.unsqueeze(1)
    vp_img = torch.cat([torch.zeros_like(pano_embeds[:, :1]), mem, pano_embeds], 1)
    pos = []
    for i, gmap in enumerate(gmaps):
        cand = gmap.get_pos_fts(obs[i]["viewpoint"], cand_vpids[i], obs[i]["heading"], obs[i]["elevation"])
        start = gmap.get_pos_fts(obs[i]["viewpoint"], [gmap.start_vp], obs[i]["heading"], obs[i]["elevation"])
        p = np.zeros((vp_img.size(1), 14), dtype=np.float32)
        p[:, :7] = start
        p[2:len(cand) + 2, 7:] = cand
        pos.append(torch.from_numpy(p))
    nav_masks = torch.cat([torch.ones(B, 1, dtype=torch.bool), torch.zeros(B, 1, dtype=torch.bool), nav_types == 1], 1)
    return dict(vp_img_embeds=vp_img, vp_pos_fts=pad_rows(pos), vp_masks=seq_masks(view_lens + 2, vp_img.size(1)),
                vp_nav_masks=nav_masks, vp_cand_vpids=[[None, None] + x for x in cand_vpids])


def ndtw(dist, pred, ref, threshold=3.0):                    # r2r/eval_utils.py cal_dtw
    n, m = len(pred), len(ref)
    dtw = np.inf * np.ones((n + 1, m + 1))
    dtw[0][0] = 0
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            dtw[i][j] = dist[pred[i - 1]][ref[j - 1]] + min(dtw[i - 1][j], dtw[i][j - 1], dtw[i - 1][j - 1])
    return np.exp(-dtw[n][m] / (threshold * m))


def teacher_action(env, obs, vpids, ended, visited_masks, imitation_learning, t, traj, expert_policy="spl"):
    a = np.zeros(len(obs), dtype=np.int64)
    for i, ob in enumerate(obs):
        if ended[i]:
            a[i] = IGNORE
        elif imitation_learning:
            assert ob["viewpoint"] == ob["gt_path"][t]
            if t == len(ob["gt_path"]) - 1:
                a[i] = 0
            else:
                for j, vpid in enumerate(vpids[i]):
                    if ob["gt_path"][t + 1] == vpid:
                        a[i] = j
                        break
        elif ob["viewpoint"] == ob["gt_path"][-1]:
            a[i] = 0
        else:
            scan, cur = ob["scan"], ob["viewpoint"]
            best, best_d = IGNORE, float("inf")
            for j, vpid in enumerate(vpids[i]):
                if j > 1 and not visited_masks[i][j]:
                    if expert_policy == "ndtw":
                        d = -ndtw(env.shortest_distances[scan], sum(traj[i]["path"], []) + env.shortest_paths[scan][cur][vpid][1:],
                                  ob["gt_path"])
                    else:
                        d = env.shortest_distances[scan][vpid][ob["gt_path"][-1]] + env.shortest_distances[scan][cur][vpid]
                    if d < best_d:
                        best, best_d = j, d
            a[i] = best
    return torch.from_numpy(a)


# ---- the loop ------------------------------------------------------------------------------------------------------
def rollout(env, student, obs, *, feedback="teacher", train_ml=1.0, max_action_len=15, teacher=None, kd=None, rw_seq=None,
            expert_policy="spl", sample_draws=None, record=None, train_teacher=False):
    """One episode batch.  `obs` = env.reset(...).  teacher: frozen teacher model (MAKD t2s, agent.py:1024) with
    kd = dict(heads=<5 projection heads>, alpha, temperature, decay); rw_seq[t] = the 5 MKRW weights of step t
    (the reference draws them per step, :866-871; passed in so both sides of a parity test use the same draw).
    sample_draws[t] = uniform numbers standing in for Categorical.sample() under feedback='sample'.
    train_teacher: ICoD co-training (args.train_kdl_teacher): the teacher runs with gradients and gets its own loss
    `t_loss` = t_alpha * (sum of the reverse 's2t' MAKD terms * train_ml) + (1 - t_alpha) * (teacher CE * train_ml / B)
    (agent.py:1013-1026,1138-1149; the reverse terms use the STUDENT's sample weights and the student's projection heads on the
    detached student side, reduction 'mean').
    Returns dict(loss, ml_loss, kdl, traj, steps=[per-step records][, t_loss])."""
    import contextlib
    tgrad = contextlib.nullcontext if train_teacher else torch.no_grad
    B = len(obs)
    scanvp_cands = {}

    def note_cands(obs):
        for ob in obs:
            d = scanvp_cands.setdefault(f"{ob['scan']}_{ob['viewpoint']}", {})
            for c in ob["candidate"]:
                d[c["viewpointId"]] = c["pointId"]
    note_cands(obs)
    gmaps = [RefGraphMap(ob["viewpoint"]) for ob in obs]
    for i, ob in enumerate(obs):
        gmaps[i].update_graph(ob)
    traj = [dict(instr_id=ob["instr_id"], path=[[ob["viewpoint"]]]) for ob in obs]
    lang = language_variable(obs)
    txt_embeds, txt_attns = student("language", lang)
    s_out = dict(txt_embeds=txt_embeds, txt_attns=txt_attns)
    t_out = {}
    if teacher is not None:
        with tgrad():
            t_txt, t_txt_attns = teacher("language", lang)
        t_out = dict(txt_embeds=t_txt, txt_attns=t_txt_attns)
    ended, just_ended = np.array([False] * B), np.array([False] * B)
    last, t_last = None, None
    ml_loss = 0.0
    kdl = defaultdict(float)
    t_ml_loss, t_kdl = 0.0, defaultdict(float)
    steps = []
    for t in range(max_action_len):
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.node_step_ids[obs[i]["viewpoint"]] = t + 1
        pano = panorama_variable(obs)
        pe, pm, pf, pa = student("panorama", pano)
        s_out.update(pano_embeds=pe, pano_fused_embeds=pf, img_attns=pa)
        if teacher is not None:
            with tgrad():
                tpe, _, tpf, tpa = teacher("panorama", pano)
            t_out.update(pano_embeds=tpe, pano_fused_embeds=tpf, img_attns=tpa)
        for i, g in enumerate(gmaps):
            if ended[i]:
                continue
            g.update_node_embed(obs[i]["viewpoint"], pf[i], rewrite=True)
            if teacher is not None:
                g.update_node_embed(obs[i]["viewpoint"], tpf[i], rewrite=True, teacher=True)
            for j, cv in enumerate(pano["cand_vpids"][i]):
                if not g.graph.visited(cv):
                    g.update_node_embed(cv, pe[i, j])
                    if teacher is not None:
                        g.update_node_embed(cv, tpe[i, j], teacher=True)
        nav = nav_gmap_variable(obs, gmaps, last, teacher=False)
        nav.update(nav_vp_variable_mem(obs, gmaps, pe, pano["cand_vpids"], pano["view_lens"], pano["nav_types"], last))
        nav.update(txt_embeds=txt_embeds, txt_masks=lang["txt_masks"])
        outs = student("navigation", nav)
        s_out.update(nav_outs=outs, nav_logits=outs["fused_logits"])
        last = outs["cls_embeds"]
        logits, vpids = outs["fused_logits"], nav["gmap_vpids"]
        probs = torch.softmax(logits, 1)
        if teacher is not None:
            t_nav = nav_gmap_variable(obs, gmaps, t_last, teacher=True)
            t_nav.update(nav_vp_variable_mem(obs, gmaps, tpe, pano["cand_vpids"], pano["view_lens"], pano["nav_types"], t_last))
            t_nav.update(txt_embeds=t_txt, txt_masks=lang["txt_masks"])
            with tgrad():
                t_outs = teacher("navigation", t_nav)
            t_out.update(nav_outs=t_outs, nav_logits=t_outs["fused_logits"])
            t_last = t_outs["cls_embeds"]
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.node_stop_scores[obs[i]["viewpoint"]] = {"stop": probs[i, 0].item()}
        targets = teacher_action(env, obs, vpids, ended, nav["gmap_visited_masks"], feedback == "teacher", t, traj, expert_policy)
        ce = F.cross_entropy(logits, targets, ignore_index=IGNORE, reduction="none")
        ml_loss = ml_loss + ce.sum()
        if teacher is not None:
            with tgrad():
                t_ce = F.cross_entropy(t_out["nav_logits"], targets, ignore_index=IGNORE, reduction="none")
            t_out["sample_weights"] = M.exponential_decay(t_ce.detach(), kd["decay"]).detach()
            if train_teacher:
                t_ml_loss = t_ml_loss + t_ce.sum()
                s_out["sample_weights"] = M.exponential_decay(ce.detach(), kd["decay"]).detach()
                t_acc = defaultdict(float, t_kdl)
                t_kdl = M.nav_makd(t, t_out, s_out, kd["heads"], t_acc, role="s2t", temperature=kd["temperature"],
                                   weights=None if rw_seq is None else rw_seq[t], weight_mode="RW" if rw_seq is not None else None)
            acc = defaultdict(float, kdl)
            kdl = M.nav_makd(t, s_out, t_out, kd["heads"], acc, role="t2s", loss_type="sum", temperature=kd["temperature"],
                             weights=None if rw_seq is None else rw_seq[t], weight_mode="RW" if rw_seq is not None else None)
        if feedback == "teacher":
            a_t = targets
        elif feedback == "argmax":
            a_t = logits.max(1)[1].detach()
        elif feedback == "sample":
            cdf = probs.detach().double().cumsum(1)
            u = torch.as_tensor(sample_draws[t], dtype=torch.float64)
            a_t = (cdf < (u * cdf[:, -1])[:, None]).sum(1).clamp(max=probs.shape[1] - 1)
        else:
            raise ValueError(feedback)
        if feedback in ("teacher", "sample"):
            stop = [ob["viewpoint"] == ob["gt_path"][-1] for ob in obs]
        else:
            stop = (a_t == 0).tolist()
        cpu_a = []
        for i in range(B):
            if stop[i] or ended[i] or nav["no_vp_left"][i] or t == max_action_len - 1:
                cpu_a.append(None)
                just_ended[i] = True
            else:
                cpu_a.append(vpids[i][int(a_t[i])])
        steps.append(dict(logits=logits.detach().clone(), targets=targets.clone(), actions=list(cpu_a), a_t=a_t.clone(),
                          vpids=[list(v) for v in vpids], nav=nav if record == "nav" else None,
                          pano=pano if record == "nav" else None,
                          embeds=(pe.detach(), pf.detach(), outs["cls_embeds"].detach()) if record == "nav" else None))
        # make_equiv_action
        hops_from = [None] * B
        for i, ob in enumerate(obs):
            if cpu_a[i] is not None:
                traj[i]["path"].append(gmaps[i].graph.path(ob["viewpoint"], cpu_a[i]))
                hops_from[i] = traj[i]["path"][-2][-1] if len(traj[i]["path"][-1]) == 1 else traj[i]["path"][-1][-2]
        env.step(cpu_a, hops_from)
        for i in range(B):
            if not ended[i] and just_ended[i]:
                node, best = None, -float("inf")
                for k, v in gmaps[i].node_stop_scores.items():
                    if v["stop"] > best:
                        node, best = k, v["stop"]
                if node is not None and obs[i]["viewpoint"] != node:
                    traj[i]["path"].append(gmaps[i].graph.path(obs[i]["viewpoint"], node))
        obs = env._get_obs()
        note_cands(obs)
        for i, ob in enumerate(obs):
            if not ended[i]:
                gmaps[i].update_graph(ob)
        ended[:] = np.logical_or(ended, np.array([x is None for x in cpu_a]))
        if ended.all():
            break
    ml = ml_loss * train_ml / B
    if teacher is not None:
        kd_sum = sum(kdl.values()) / B
        total = kd["alpha"] * kd_sum + (1 - kd["alpha"]) * ml
    else:
        kd_sum, total = None, ml
    out = dict(loss=total, ml_loss=ml, kdl=kd_sum, kdl_terms=dict(kdl), traj=traj, steps=steps, gmaps=gmaps)
    if teacher is not None and train_teacher:
        ta = kd.get("t_alpha", kd["alpha"])
        out["t_kdl_terms"] = dict(t_kdl)
        out["t_loss"] = ta * (sum(t_kdl.values()) * train_ml) + (1 - ta) * (t_ml_loss * train_ml / B)
    return out
