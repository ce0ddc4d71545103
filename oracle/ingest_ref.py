"""ORACLE (test infrastructure, never shipped): numpy restatement of the reference's per-trajectory panorama token assembly,
/root/reference/pretrain_src/data/dataset.py:729-772 (`get_traj_pano_fts`), with the helpers it calls from
data/common.py:77-103 (`get_angle_fts`, `get_view_rel_angles`).  Pinned by tests/golden/ingest.pt, minted by running the
reference function itself on synthetic candidate tables (tests/golden/mint_golden.py::mint_ingest).
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
