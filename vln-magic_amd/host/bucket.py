"""Shape buckets for streaming batches under HIP-graph replay.

A captured graph needs static shapes, a streamed batch has ragged ones: instruction length L (collate pads to the batch maximum,
pretrain_src/data/tasks.py:116), map size K (:142-143), total trajectory steps Np = sum T_b (:133), masked tokens (mlm).  `pad_batch` pads a
collated batch to its bucket -- L to the truncation length, K / Np / n_mask to multiples of a quantum -- and records the TRUE sizes;
`plan.build_plan_host(..., pad=...)` then builds an index plan whose every array depends on the bucket only, so a packed record
(loader.pack) of a bucket has ONE layout and can be copied into the buffer a captured graph reads.

What keeps the result identical to the unpadded batch (tests/test_stream_graph_gpu.py): padding is inert wherever the network masks it
already (attention keys, action logits, gathers, ignored labels); the MAKD terms, whose `mean` runs over the batch's own padded extent
(pretrain_src/optim/kd_loss.py:5-16), get their true extents and normalisers from device memory (magic_mse_multi valid_dev / norm_dev), and the
MLM and MRC losses their 1 / n_mask through per-row weights (padded MRC rows carry an all-zero target distribution: no loss, no gradient).
"""
import torch


def rup(x, q):
    return (int(x) + q - 1) // q * q


def bucket_of(batch, task, L=80, q_k=8, q_np=32, q_mask=64, q_mrc=32):
    """bucket sizes of a collated batch"""
    bk = dict(L=max(L, int(batch["txt_ids"].shape[1])), K=rup(batch["gmap_step_ids"].shape[1], q_k), Np=rup(sum(batch["traj_step_lens"]), q_np), n_mask=0)
    if task == "mlm":
        bk["n_mask"] = rup(max(int((batch["txt_labels"] != -1).sum()), 1), q_mask)
    if task == "mrc":               # masked views of the current viewpoint (tasks.py:205-221: ~15 % of <= 36 views per sample)
        bk["n_mask"] = rup(max(int(batch["vp_view_mrc_masks"].sum()), 1), q_mrc)
    return bk


def _pad(t, dim, size, value=0):
    if t is None or t.shape[dim] == size:
        return t
    shape = list(t.shape)
    shape[dim] = size - t.shape[dim]
    return torch.cat([t, torch.full(shape, value, dtype=t.dtype)], dim)


def pad_batch(batch, task, bk):
    """-> (padded copy of the collated batch, true sizes).  Lists (vpids, step lens) and per-sample lengths stay as they are."""
    true = dict(L=int(batch["txt_ids"].shape[1]), K=int(batch["gmap_step_ids"].shape[1]), Np=int(sum(batch["traj_step_lens"])),
                n_mask=int((batch["txt_labels"] != -1).sum()) if task == "mlm" else int(batch["vp_view_mrc_masks"].sum()) if task == "mrc" else 0)
    if true["L"] > bk["L"] or true["K"] > bk["K"] or true["Np"] > bk["Np"] or true["n_mask"] > bk["n_mask"]:
        raise ValueError(f"batch {true} does not fit bucket {bk}")
    b = dict(batch)
    b["txt_ids"] = _pad(batch["txt_ids"], 1, bk["L"], 0)
    if batch.get("txt_labels") is not None:
        b["txt_labels"] = _pad(batch["txt_labels"], 1, bk["L"], -1)
    for k in ("traj_vp_view_lens", "traj_view_img_fts", "traj_vp_row", "traj_view_order", "traj_loc_fts", "traj_nav_types"):
        if torch.is_tensor(batch.get(k)):
            b[k] = _pad(batch[k], 0, bk["Np"], 0)
    for k in ("gmap_step_ids", "gmap_visited_masks", "gmap_pos_fts"):
        b[k] = _pad(batch[k], 1, bk["K"], 0)
    b["gmap_pair_dists"] = _pad(_pad(batch["gmap_pair_dists"], 1, bk["K"], 0), 2, bk["K"], 0)
    return b, true
