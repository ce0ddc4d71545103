"""Feature ingest: the precomputed CLIP view features as ONE packed table in HBM plus index-only
batches, instead of the reference's host dict of `[36, >=768]` float32 arrays that is re-stacked, padded and uploaded for
every batch (pretrain_src/data/dataset.py:246-254 `get_image_feature_from_h5py`; `get_traj_pano_fts` :729-772;
`pad_tensors` + H2D in tasks.py collates / loader.py:78-120).

R2R has ~10.6 k viewpoints: 36 x 768 bf16 = 55 KB each, 584 MB in all -- 0.2 % of one MI355X's 288 GB.  Per trajectory step
the host now sends 1 + V int32 (the viewpoint's table row and its view order: candidate views first, then the rest,
dataset.py:742-756) and `magic_view_gather` streams the 55 KB row block HBM -> HBM in token order; loc_fts / nav_types are
tiny and stay on the existing path.  The HDF5 reader itself is out of scope (no h5py here): `from_arrays` takes what
`f[key][...][:, :image_feat_size]` would return.
"""
import math

import numpy as np
import torch

from . import ops as O

V_VIEWS = 36


def get_view_rel_angles(base_view_id=0):
    """data/common.py:85-103"""
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


_REL12 = get_view_rel_angles(12)


def pano_view_order(nav_cands, angle_feat_size=4):
    """Token order of one panorama (dataset.py:738-768, correct_heading off): candidates first, in the dict's order, each
    contributing the view it is seen in (duplicates allowed), then every view no candidate used.
    nav_cands: {cand_vp: (viewidx, _, d_heading, d_elevation)}.  Returns (order int32 [n_views], loc_fts [n_views, A+3],
    nav_types [n_views], cand_vpids)."""
    order, ang, cand_vpids = [], [], []
    used = set()
    for k, v in nav_cands.items():
        used.add(v[0])
        order.append(v[0])
        ang.append((_REL12[v[0], 0] + v[2], _REL12[v[0], 1] + v[3]))
        cand_vpids.append(k)
    rest = [i for i in range(V_VIEWS) if i not in used]
    order.extend(rest)
    ang.extend((_REL12[i, 0], _REL12[i, 1]) for i in rest)
    ang = np.asarray(ang, np.float32)
    f = np.stack([np.sin(ang[:, 0]), np.cos(ang[:, 0]), np.sin(ang[:, 1]), np.cos(ang[:, 1])], 1).astype(np.float32)
    rep = angle_feat_size // 4
    if rep > 1:
        f = np.concatenate([f] * rep, 1)
    loc = np.concatenate([f, np.ones((len(order), 3), np.float32)], 1)
    nav = np.array([1] * len(cand_vpids) + [0] * len(rest), np.int64)
    return np.asarray(order, np.int32), loc, nav, cand_vpids


class FeatureTable:
    """[n_viewpoints, 36, D] in HBM in the compute dtype + key -> row index."""

    def __init__(self, keys, table, n_base=None):
        self.index = {k: i for i, k in enumerate(keys)}
        self.table = table
        self.n_base = n_base if n_base is not None else len(self.index)      # rows [n_base, 2 n_base) = the augmented copies
        self.has_aug = table.shape[0] == 2 * self.n_base

    @classmethod
    def from_arrays(cls, keys, arrays, device="cuda", dtype=torch.bfloat16, image_feat_size=768, aug_arrays=None):
        """arrays[i]: the `[36, >= image_feat_size]` float32 block stored under keys[i] = "{scan}_{viewpoint}".
        aug_arrays: the same viewpoints from the EnvEdit-style augmented feature file (`aug_img_file`,
        r2r_magic_pretrain.json; dataset.py:606-610), appended as rows [n, 2n)."""
        blocks = [a[:, :image_feat_size] for a in arrays]
        if aug_arrays is not None:
            assert len(aug_arrays) == len(arrays)
            blocks += [a[:, :image_feat_size] for a in aug_arrays]
        t = torch.from_numpy(np.stack(blocks).astype(np.float32))
        return cls(keys, t.to(device).to(dtype).contiguous(), n_base=len(arrays))

    def row(self, scan, vp):
        return self.index[f"{scan}_{vp}"]

    def batch_indices(self, scans, paths, cands_of, angle_feat_size=4, pad_views=None, aug_coin=None):
        """aug_coin: callable returning a uniform [0,1) draw; when given (and the table holds augmented copies) every panorama
        independently reads the augmented row with probability 1/2 -- `get_scanvp_feature` (dataset.py:606-610) flips
        `np.random.rand() > 0.5` once per viewpoint visit, in path order; pass `np.random.rand` for the same stream.

        Index-only description of a batch of trajectories (what `get_traj_pano_fts` + the collates' pad/stack produce as
        tensors): returns dict(vp_row int32 [Np], order int32 [Np, V] (-1 = padded slot), traj_vp_view_lens, traj_loc_fts
        [Np, V, A+3] zero-padded, traj_nav_types [Np, V], traj_cand_vpids, traj_step_lens)."""
        rows, orders, locs, navs, cand_ids, step_lens = [], [], [], [], [], []
        for scan, path in zip(scans, paths):
            step_lens.append(len(path))
            cl = []
            for vp in path:
                o, loc, nav, cv = pano_view_order(cands_of(scan, vp), angle_feat_size)
                r = self.row(scan, vp)
                if aug_coin is not None and self.has_aug and aug_coin() > 0.5:
                    r += self.n_base
                rows.append(r)
                orders.append(o)
                locs.append(loc)
                navs.append(nav)
                cl.append(cv)
            cand_ids.append(cl)
        V = max(max(len(o) for o in orders), pad_views or 0)
        Np = len(rows)
        order = np.full((Np, V), -1, np.int32)
        loc = np.zeros((Np, V, locs[0].shape[1]), np.float32)
        nav = np.zeros((Np, V), np.int64)
        for i, (o, l_, n_) in enumerate(zip(orders, locs, navs)):
            order[i, :len(o)], loc[i, :len(o)], nav[i, :len(o)] = o, l_, n_
        return dict(vp_row=torch.tensor(rows, dtype=torch.int32), order=torch.from_numpy(order),
                    traj_vp_view_lens=torch.tensor([len(o) for o in orders], dtype=torch.long),
                    traj_loc_fts=torch.from_numpy(loc), traj_nav_types=torch.from_numpy(nav),
                    traj_cand_vpids=cand_ids, traj_step_lens=step_lens)

    def gather(self, vp_row, order, out=None):
        """traj_view_img_fts [Np, V, D] in the table's dtype, assembled on the device in the reference's token order"""
        dev = self.table.device
        vp_row, order = vp_row.to(dev, non_blocking=True), order.to(dev, non_blocking=True).contiguous()
        if out is None:
            out = torch.empty(order.shape[0], order.shape[1], self.table.shape[2], dtype=self.table.dtype, device=dev)
        return O.view_gather(self.table, vp_row, order, out)
