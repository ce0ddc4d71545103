"""Seeded synthetic R2R-shaped pretrain batches (SURVEY §8d "Synthetic inputs").

Produces exactly the batch dict that the reference's collate functions hand to the model
(/root/reference/pretrain_src/data/tasks.py: mlm_collate :110, sap_collate :392, cfp_collate :618;
key layout in SURVEY App. A.1), including the python-side viewpoint-id lists
(traj_vpids / traj_cand_vpids / gmap_vpids) that drive the map aggregation and the local->global
logit fusion.  No dataset is read: features are gaussian, graphs are random walks.
"""
import math
import random

import numpy as np
import torch

V_VIEWS = 36
IMG_DIM = 768
# the [stop] node's 7 map position features as the reference builds them (pretrain_src/data/dataset.py:557-560 + data/common.py:77-83:
# rel_angles [0, 0] -> [sin 0, cos 0, sin 0, cos 0], rel_dists [0, 0, 0]); pinned by tests/golden/gmap_pos.pt.  Rounds 1-5 wrote a zero row
# here, which together with zero-initialised biases made `gmap_pos_embeddings` a LayerNorm of the zero vector (rstd = 1e6) on every sample.
STOP_NODE_POS_FTS = (0.0, 1.0, 0.0, 1.0, 0.0, 0.0, 0.0)


def _angle_fts(rng, n):
    h = rng.uniform(-math.pi, math.pi, size=n)
    e = rng.uniform(-math.pi / 6, math.pi / 6, size=n)
    return np.stack([np.sin(h), np.cos(h), np.sin(e), np.cos(e)], 1).astype(np.float32)


def make_sample(rng, pyrng, *, min_len=20, max_len=80, min_steps=4, max_steps=7, vocab=50265,
                uid=0, dup_view_prob=0.0, img_dim=IMG_DIM):
    """One trajectory sample in the format of R2RTextPathData.get_input
    (pretrain_src/data/dataset.py:640-727).  dup_view_prob: probability that a panorama has two candidates seen in the same
    discretised view -- the reference then emits that view twice and the panorama has 37 tokens (dataset.py:742-756:
    candidate views are appended per candidate, the remaining views once), so batches become ragged in the view dimension."""
    L = int(rng.integers(min_len, max_len + 1))
    txt = rng.integers(3, vocab - 1, size=L).astype(np.int64)
    txt[0] = 0      # <s>
    txt[-1] = 2     # </s>
    T = int(rng.integers(min_steps, max_steps + 1))
    path = [f"s{uid}_p{t}" for t in range(T)]
    frontier = []          # unvisited vps seen so far (insertion ordered)
    n_new = 0
    traj_view, traj_loc, traj_nav, traj_cand = [], [], [], []
    for t in range(T):
        n_cand = int(rng.integers(2, 7))
        cands = []
        if t + 1 < T:
            cands.append(path[t + 1])
        if t > 0:
            cands.append(path[t - 1])          # backtrack candidate (visited)
        while len(cands) < n_cand:
            if frontier and rng.random() < 0.3:
                c = frontier[int(rng.integers(0, len(frontier)))]
                if c in cands:
                    continue
            else:
                c = f"s{uid}_f{n_new}"
                n_new += 1
            cands.append(c)
        pyrng.shuffle(cands)
        for c in cands:
            if c not in path and c not in frontier:
                frontier.append(c)
        traj_cand.append(cands)
        nv = V_VIEWS + (1 if (dup_view_prob > 0 and len(cands) >= 2 and rng.random() < dup_view_prob) else 0)
        traj_view.append(rng.standard_normal((nv, img_dim)).astype(np.float32))
        loc = np.concatenate([_angle_fts(rng, nv), np.ones((nv, 3), np.float32)], 1)
        traj_loc.append(loc)
        traj_nav.append(np.array([1] * len(cands) + [0] * (nv - len(cands)), np.int64))
    visited = set(path)
    unvisited = [c for c in frontier if c not in visited]
    gmap_vpids = [None] + path + unvisited
    K = len(gmap_vpids)
    gmap_step_ids = [0] + list(range(1, T + 1)) + [0] * len(unvisited)
    gmap_visited = [0] + [1] * T + [0] * len(unvisited)
    gmap_pos = np.concatenate([_angle_fts(rng, K), rng.uniform(0, 1, (K, 3)).astype(np.float32)], 1)
    gmap_pos[0] = STOP_NODE_POS_FTS
    d = rng.uniform(0, 30, (K, K)).astype(np.float32)
    d = (d + d.T) / 2
    np.fill_diagonal(d, 0)
    d[0, :] = 0
    d[:, 0] = 0
    n_last = len(traj_cand[-1])
    vp_pos = np.zeros((traj_view[-1].shape[0] + 1, 14), np.float32)
    vp_pos[:, :7] = np.concatenate([_angle_fts(rng, 1), rng.uniform(0, 1, (1, 3)).astype(np.float32)], 1)
    vp_pos[1:1 + n_last, 7:] = np.concatenate(
        [_angle_fts(rng, n_last), rng.uniform(0, 1, (n_last, 3)).astype(np.float32)], 1)
    # action labels (dataset.py:622-638): stop with p=0.2 else an unvisited node
    if rng.random() < 0.2 or not unvisited:
        g_label, l_label = 0, 0
    else:
        tgt = unvisited[int(rng.integers(0, len(unvisited)))]
        g_label = gmap_vpids.index(tgt)
        l_label = traj_cand[-1].index(tgt) + 1 if tgt in traj_cand[-1] else -100
    return dict(
        txt_ids=torch.from_numpy(txt),
        traj_view_img_fts=[torch.from_numpy(x) for x in traj_view],
        traj_loc_fts=[torch.from_numpy(x) for x in traj_loc],
        traj_nav_types=[torch.from_numpy(x) for x in traj_nav],
        traj_cand_vpids=traj_cand, traj_vpids=path,
        gmap_vpids=gmap_vpids, gmap_step_ids=torch.tensor(gmap_step_ids, dtype=torch.long),
        gmap_visited_masks=torch.tensor(gmap_visited, dtype=torch.bool),
        gmap_pos_fts=torch.from_numpy(gmap_pos), gmap_pair_dists=torch.from_numpy(d),
        vp_pos_fts=torch.from_numpy(vp_pos),
        local_act_labels=l_label, global_act_labels=g_label)


def random_word_mask(txt, rng, vocab, mask_id=None, prob=0.15):
    """MLM corruption rule of tasks.py:random_word :11-52 (15 %; 80/10/10), -1 = not predicted.
    At least one token is masked (tasks.py:46-50)."""
    mask_id = vocab - 1 if mask_id is None else mask_id      # RoBERTa <mask> = 50264
    src = txt.tolist()                          # plain lists inside the loop: element-wise tensor indexing was 2/3 of the mlm collate's time
    out, labels = list(src), [-1] * len(src)
    hit = False
    for i in range(1, len(src) - 1):
        p = rng.random()
        if p < prob:
            p /= prob
            labels[i] = src[i]
            hit = True
            if p < 0.8:
                out[i] = mask_id
            elif p < 0.9:
                out[i] = int(rng.integers(3, vocab - 1))
    if not hit:
        labels[1] = src[1]
        out[1] = mask_id
    return torch.tensor(out, dtype=txt.dtype), torch.tensor(labels, dtype=txt.dtype)


def _pad_stack(ts, pad=0):
    """pad_tensors / pad_sequence of the reference collates (common.py:9-24): [len(ts), max rows, ...] filled with `pad`.  One C++ call
    instead of a Python loop over ~300 panoramas (5 of a worker's ~16 ms per B=48 batch)."""
    n0 = ts[0].shape[0]
    if all(t.shape[0] == n0 for t in ts):
        return torch.stack(ts)
    return torch.nn.utils.rnn.pad_sequence(ts, batch_first=True, padding_value=pad)


def collate(samples, task, rng=None, vocab=50265, mrc_mask_prob=0.15, prob_size=1000):
    """Same layout as {mlm,mrc,sap,cfp}_collate (tasks.py:110-176, :263-310, :392-451, :618-678)."""
    b = {}
    txt = [s["txt_ids"] for s in samples]
    if task == "mlm":
        pairs = [random_word_mask(t, rng, vocab) for t in txt]
        txt = [p[0] for p in pairs]
        b["txt_labels"] = _pad_stack([p[1] for p in pairs], -1)
    b["txt_lens"] = torch.tensor([len(t) for t in txt], dtype=torch.long)
    b["txt_ids"] = _pad_stack(txt, 0)
    if task == "mrc":        # MrcDataset.__getitem__ (tasks.py:205-221): mask views of the LAST panorama, keep their soft labels
        samples = [dict(s) for s in samples]
        masks, probs = [], []
        for s in samples:
            nv = s["traj_view_img_fts"][-1].shape[0]
            m = rng.random(nv) < mrc_mask_prob
            if not m.any():
                m[int(rng.integers(0, nv))] = True          # at least one (tasks.py:170-175)
            m = torch.from_numpy(m)
            views = list(s["traj_view_img_fts"])
            views[-1] = views[-1].masked_fill(m[:, None], 0)     # _mask_img_feat (tasks.py:177-181)
            s["traj_view_img_fts"] = views
            masks.append(m)
            probs.append(torch.softmax(torch.from_numpy(rng.standard_normal((nv, prob_size)).astype(np.float32)) * 2, -1))
        b["vp_view_mrc_masks"] = _pad_stack(masks, False)
        b["vp_view_probs"] = _pad_stack(probs, 0)
    b["traj_step_lens"] = [len(s["traj_view_img_fts"]) for s in samples]
    b["traj_vp_view_lens"] = torch.tensor(
        sum([[len(y) for y in s["traj_view_img_fts"]] for s in samples], []), dtype=torch.long)
    b["traj_view_img_fts"] = _pad_stack(sum([s["traj_view_img_fts"] for s in samples], []))        # pad_tensors (common.py:9)
    b["traj_loc_fts"] = _pad_stack(sum([s["traj_loc_fts"] for s in samples], []))
    b["traj_nav_types"] = _pad_stack(sum([s["traj_nav_types"] for s in samples], []))             # pad_sequence, value 0
    b["traj_reverie_loc_fts"] = None
    b["traj_cand_vpids"] = [s["traj_cand_vpids"] for s in samples]
    b["traj_vpids"] = [s["traj_vpids"] for s in samples]
    b["gmap_vpids"] = [s["gmap_vpids"] for s in samples]
    b["gmap_lens"] = torch.tensor([len(s["gmap_step_ids"]) for s in samples], dtype=torch.long)
    b["gmap_step_ids"] = _pad_stack([s["gmap_step_ids"] for s in samples], 0)
    b["gmap_visited_masks"] = _pad_stack([s["gmap_visited_masks"] for s in samples], False)
    b["gmap_pos_fts"] = _pad_stack([s["gmap_pos_fts"] for s in samples], 0)
    K = int(b["gmap_lens"].max())
    pd = torch.zeros(len(samples), K, K)
    for i, s in enumerate(samples):
        k = len(s["gmap_step_ids"])
        pd[i, :k, :k] = s["gmap_pair_dists"]
    b["gmap_pair_dists"] = pd
    b["vp_lens"] = torch.tensor([s["vp_pos_fts"].shape[1] for s in samples], dtype=torch.long)  # sic: tasks.py:434
    b["vp_pos_fts"] = _pad_stack([s["vp_pos_fts"] for s in samples])
    if task in ("sap", "cfp"):
        b["local_act_labels"] = torch.tensor([s["local_act_labels"] for s in samples], dtype=torch.long)
        b["global_act_labels"] = torch.tensor([s["global_act_labels"] for s in samples], dtype=torch.long)
    return b


def make_batch(task, batch_size=48, seed=1234, step=0, vocab=50265, **kw):
    """Deterministic batch for (seed, step): per-rank streams use seed = 1234 + rank."""
    rng = np.random.default_rng([seed, step])
    pyrng = random.Random(seed * 1000003 + step)
    samples = [make_sample(rng, pyrng, vocab=vocab, uid=i, **kw) for i in range(batch_size)]
    return collate(samples, task, rng=rng, vocab=vocab)


def batch_to(batch, device):
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
