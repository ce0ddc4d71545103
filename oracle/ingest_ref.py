"""ORACLE (test infrastructure, never shipped): numpy restatement of the reference's per-trajectory panorama token assembly,
/root/reference/pretrain_src/data/dataset.py:729-772 (`get_traj_pano_fts`), with the helpers it calls from
data/common.py:77-103 (`get_angle_fts`, `get_view_rel_angles`).  Pinned by tests/golden/ingest.pt, minted by running the
reference function itself on synthetic candidate tables (tests/golden/mint_golden.py::mint_ingest); `gmap_pos_fts` (dataset.py:553-575)
by tests/golden/gmap_pos.pt (mint_gmap_pos).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file."""
import math

import numpy as np


def get_view_rel_angles(base_view_id=0):
    """common.py:85-103: heading/elevation of the 36 discretised views relative to view `base_view_id`"""
    rel = np.zeros((36, 2), dtype=np.float32)
    base_heading = (base_view_id % 12) * math.radians(30)
    base_elevation = (base_view_id // 12 - 1) * math.radians(30)
    heading = elevation = 0.0
    for ix in range(36):
        if ix == 0:
            heading, elevation = 0.0, math.radians(-30)
        elif ix % 12 == 0:
            heading = 0.0
            elevation += math.radians(30)
        else:
            heading += math.radians(30)
        rel[ix, 0] = heading - base_heading
        rel[ix, 1] = elevation - base_elevation
    return rel


def get_angle_fts(headings, elevations, angle_feat_size):
    """common.py:77-83"""
    f = np.vstack([np.sin(headings), np.cos(headings), np.sin(elevations), np.cos(elevations)]).transpose().astype(np.float32)
    rep = angle_feat_size // 4
    return np.concatenate([f] * rep, 1) if rep > 1 else f


def vp_rel_pos(a, b, base_heading=0.0, base_elevation=0.0):
    """common.py:145-162: heading / elevation / distance of position b seen from position a (the simulator's x-y axes are transposed)"""
    dx, dy, dz = b[0] - a[0], b[1] - a[1], b[2] - a[2]
    xy = max(np.sqrt(dx ** 2 + dy ** 2), 1e-8)
    xyz = max(np.sqrt(dx ** 2 + dy ** 2 + dz ** 2), 1e-8)
    heading = np.arcsin(dx / xy)
    if b[1] < a[1]:
        heading = np.pi - heading
    return heading - base_heading, np.arcsin(dz / xyz) - base_elevation, xyz


def gmap_pos_fts(pos_of, dist_of, path_len_of, cur_vp, gmap_vpids, cur_heading, cur_elevation, angle_feat_size=4, max_dist=30, max_step=10):
    """dataset.py:553-575 (`get_gmap_pos_fts`): 7 features per map node relative to the current viewpoint -- angle features of the relative
    heading / elevation, then line distance, shortest distance and shortest-path hops, normalised.  The [stop] node (vpid None) takes
    rel_angles [0, 0] and rel_dists [0, 0, 0], i.e. the row [sin 0, cos 0, sin 0, cos 0, 0, 0, 0] = [0, 1, 0, 1, 0, 0, 0]: NOT a zero row."""
    ang, dst = [], []
    for vp in gmap_vpids:
        if vp is None:
            ang.append([0, 0])
            dst.append([0, 0, 0])
        else:
            h, e, d = vp_rel_pos(pos_of(cur_vp), pos_of(vp), cur_heading, cur_elevation)
            ang.append([h, e])
            dst.append([d / max_dist, dist_of(cur_vp, vp) / max_dist, (path_len_of(cur_vp, vp) - 1) / max_step])
    ang, dst = np.array(ang).astype(np.float32), np.array(dst).astype(np.float32)
    return np.concatenate([get_angle_fts(ang[:, 0], ang[:, 1], angle_feat_size), dst], 1)


def gmap_inputs(cands_of, pos_of, dist_of, path_len_of, path, cur_heading, cur_elevation, act_visited_node=False, **kw):
    """dataset.py:520-552 (`get_gmap_inputs`): the map a trajectory prefix has seen -- [stop] first, then the visited viewpoints in visiting order
    (step id t + 1), then every candidate seen from them that was never visited, in first-sighting order (step id 0; a candidate visited later is
    dropped from the frontier); visited mask [0] + [1] * visited + [0] * unvisited (R2R: act_visited_node False); position features relative to
    the LAST viewpoint; pairwise shortest distances with row / column 0 ([stop]) and the diagonal zero.  cands_of(vp): candidate ids in the
    reference's dict order."""
    visited, unvisited = {}, {}
    for t, vp in enumerate(path):
        visited[vp] = t + 1
        unvisited.pop(vp, None)
        for nxt in cands_of(vp):
            if nxt not in visited:
                unvisited[nxt] = 0
    vpids = [None] + list(visited) + list(unvisited)
    step_ids = [0] + list(visited.values()) + list(unvisited.values())
    if act_visited_node:
        vis = [0] + [1 if vp == path[-1] else 0 for vp in vpids[1:]]
    else:
        vis = [0] + [1] * len(visited) + [0] * len(unvisited)
    pos = gmap_pos_fts(pos_of, dist_of, path_len_of, path[-1], vpids, cur_heading, cur_elevation, **kw)
    pair = np.zeros((len(vpids), len(vpids)), np.float32)
    for i in range(1, len(vpids)):
        for j in range(i + 1, len(vpids)):
            pair[i, j] = pair[j, i] = dist_of(vpids[i], vpids[j])
    return vpids, step_ids, vis, pos, pair


def vp_pos_fts(pos_of, dist_of, path_len_of, start_vp, cur_vp, cand_vpids, cur_heading, cur_elevation, vp_ft_len, **kw):
    """dataset.py:555-565 (`get_vp_pos_fts`): [vp_ft_len + 1, 14] -- columns 0..6 of EVERY row ([stop] row 0 included) = the start viewpoint seen from
    the current one, columns 7..13 of rows 1..len(cand) = the candidates seen from the current one, zeros elsewhere"""
    cand = gmap_pos_fts(pos_of, dist_of, path_len_of, cur_vp, cand_vpids, cur_heading, cur_elevation, **kw)
    start = gmap_pos_fts(pos_of, dist_of, path_len_of, cur_vp, [start_vp], cur_heading, cur_elevation, **kw)
    out = np.zeros((vp_ft_len + 1, 14), np.float32)
    out[:, :7] = start
    out[1:len(cand) + 1, 7:] = cand
    return out


def traj_pano_tokens(view_fts_of, path, cands_of, angle_feat_size=4):
    """dataset.py:729-772 with `correct_heading` off (the shipped default).
    view_fts_of(vp) -> [36, D]; cands_of(vp) -> ordered dict {cand_vp: (viewidx, _, d_heading, d_elevation)}.
    Returns (traj_view_img_fts, traj_loc_fts, traj_nav_types, traj_cand_vpids, last_vp_angles)."""
    rel12 = get_view_rel_angles(12)
    out_fts, out_loc, out_nav, out_cand = [], [], [], []
    last = None
    for vp in path:
        view_fts = view_fts_of(vp)
        img, ang, cand_vpids = [], [], []
        used = set()
        for k, v in cands_of(vp).items():
            used.add(v[0])
            img.append(view_fts[v[0]])
            va = rel12[v[0]]
            ang.append([va[0] + v[2], va[1] + v[3]])
            cand_vpids.append(k)
        img.extend(view_fts[i] for i in range(36) if i not in used)
        ang.extend(rel12[i] for i in range(36) if i not in used)
        img, ang = np.stack(img, 0), np.stack(ang, 0)
        loc = np.concatenate([get_angle_fts(ang[:, 0], ang[:, 1], angle_feat_size), np.ones((len(img), 3), np.float32)], 1)
        out_fts.append(img)
        out_loc.append(loc)
        out_nav.append([1] * len(cand_vpids) + [0] * (36 - len(used)))
        out_cand.append(cand_vpids)
        last = ang
    return out_fts, out_loc, out_nav, out_cand, last


def view_gather(table, vp_row, order):
    """the device gather restated: out[p, j] = table[vp_row[p], order[p, j]] (order < 0 -> zeros)"""
    Np, V = order.shape
    out = np.zeros((Np, V, table.shape[2]), table.dtype)
    for p in range(Np):
        for j in range(V):
            if order[p, j] >= 0:
                out[p, j] = table[vp_row[p], order[p, j]]
    return out
