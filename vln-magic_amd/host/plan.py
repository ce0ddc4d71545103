"""Host-side batch plan: turns the python-side structure of a collated batch (viewpoint-id lists,
ragged lengths, masks) into the small int32/uint8 device arrays the kernels consume.

This is the device-friendly restatement of the host loops the withheld model runs per batch
([LINEAGE] DUET `_aggregate_gmap_features` / `forward_sap` fusion loops; semantics pinned by
map_nav_src/r2r/agent.py:905-924 and SURVEY App. B.3-B.4): a CSR "averaging matrix" from panorama rows
to map nodes (and its transpose for the backward), a row-selection CSR for the current viewpoint's
candidate views, and the local->global logit fusion index map.
"""
import numpy as np
import torch

def _csr(rows, n_out):
    """rows: list of (out_row, src_row, weight) -> ptr[n_out+1], idx, w sorted by out_row (stable)."""
    if not rows:
        return np.zeros(n_out + 1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.float32)
    a = np.asarray(rows, np.float64)                    # (row ids < 2^24 and the weights 1 / n are exact in float64)
    o = a[:, 0].astype(np.int64)
    order = np.argsort(o, kind="stable")
    ptr = np.zeros(n_out + 1, np.int64)
    np.add.at(ptr, o + 1, 1)
    return np.cumsum(ptr).astype(np.int32), a[order, 1].astype(np.int32), a[order, 2].astype(np.float32)


def csr_pair(entries, n_out, n_src):
    """forward CSR (out <- src) and its transpose (src <- out)."""
    f = _csr(entries, n_out)
    t = _csr([(s, o, w) for (o, s, w) in entries], n_src)
    return f, t


def build_plan(batch, task, device, ld_round=8):
    """host half + device half in one call (resident / test batches).  A streaming loader runs `build_plan_host` in its worker
    processes (it is pure CPU work, 6-28 ms per B=48 batch) and `plan_to_device` on the training process's copy stream."""
    return plan_to_device(build_plan_host(batch, task), device)


def plan_to_device(hp, device):
    """ship every index array of a host plan in ONE pinned staging buffer and ONE async copy (the plan has ~60 small arrays:
    copied one by one they cost ~10 us of host time each)"""
    device = torch.device(device)
    arrays = dict(hp["cpu"])
    for name, (f, t) in hp["csr"].items():
        for j, a in enumerate(f):
            arrays[f"{name}/{j}"] = torch.as_tensor(a)
        for j, a in enumerate(t):
            arrays[f"{name}_T/{j}"] = torch.as_tensor(a)
    metas, off = [], 0
    for k, a in arrays.items():
        a = a.contiguous()
        off = (off + 15) & ~15
        metas.append((k, a, off))
        off += a.numel() * a.element_size()
    stage = torch.empty(max(off, 16), dtype=torch.uint8, pin_memory=(device.type == "cuda"))
    for k, a, o in metas:
        n = a.numel() * a.element_size()
        if n:
            stage[o:o + n] = a.reshape(-1).view(torch.uint8)
    buf = stage.to(device, non_blocking=True)
    plan, csr = {}, {}
    for k, a, o in metas:
        n = a.numel() * a.element_size()
        t = buf[o:o + n].view(a.dtype).view(a.shape) if n else torch.empty(a.shape, dtype=a.dtype, device=device)
        if "/" in k:
            name, j = k.split("/")
            csr.setdefault(name, {})[int(j)] = t
        else:
            plan[k] = t
    for name, parts in csr.items():
        plan[name] = tuple(parts[j] for j in range(len(parts)))
    plan.update(hp["meta"])
    plan["_stage"] = stage
    return plan


def _pad_csr(pair, cap):
    """pad the idx / w arrays of a (forward, transposed) CSR pair to `cap` entries: the ptr arrays never reach the padding"""
    out = []
    for ptr, idx, w in pair:
        n = len(idx)
        if n > cap:
            raise ValueError(f"CSR with {n} entries does not fit its bucket capacity {cap}")
        out.append((ptr, np.concatenate([idx, np.zeros(cap - n, np.int32)]) if n < cap else idx,
                    np.concatenate([w, np.zeros(cap - n, np.float32)]) if n < cap else w))
    return tuple(out)


def build_plan_host(batch, task, ld_round=8, pad=None):
    """the CPU-only half: numpy / CPU tensors + scalars, picklable (DataLoader workers can build it next to the collate).
    pad: (bucket, true sizes) of a batch padded by bucket.pad_batch -- every array then depends on the bucket only."""
    B = len(batch["traj_step_lens"])
    L = batch["txt_ids"].shape[1]
    K = batch["gmap_step_ids"].shape[1]
    # view slots per panorama: 36, or more when a panorama shows two candidates in one view (dataset.py:742-756); padded per batch
    V = int(batch["traj_view_order"].shape[1] if batch.get("traj_view_img_fts") is None else batch["traj_view_img_fts"].shape[1])
    # local (viewpoint) tokens: [stop] + the CURRENT panorama's views, cut to the longest current panorama of the batch
    # (= vp_pos_fts.shape[1], tasks.py:434-435); equals V + 1 unless only earlier steps have a 37-token panorama
    Vp = int(batch["vp_pos_fts"].shape[1])
    step_lens = batch["traj_step_lens"]
    Np = int(sum(step_lens))
    if pad is not None:
        Np = int(pad[0]["Np"])                 # dummy panoramas at the end: no CSR entry ever points at them
    cpu = {}
    cpu["txt_ids"] = batch["txt_ids"].reshape(-1).to(torch.int32)
    # rows of the word-embedding table this batch reads (pad id included: padded positions are embedded too) -- on steps whose only
    # use of the table is this lookup they are the only rows with gradient: trainer.GradSync exchanges them instead of the dense table
    cpu["emb_rows"] = torch.unique(batch["txt_ids"]).to(torch.int64) if pad is None else torch.zeros(0, dtype=torch.int64)
    txt_lens = batch["txt_lens"]
    cpu["txt_mask"] = (torch.arange(L)[None] < txt_lens[:, None]).to(torch.uint8)
    view_lens = batch["traj_vp_view_lens"]
    cpu["view_lens"] = view_lens.to(torch.int32)
    cpu["pano_mask"] = (torch.arange(V)[None] < view_lens[:, None]).to(torch.uint8)
    cpu["nav_types"] = batch["traj_nav_types"].reshape(-1).to(torch.int32)
    cpu["gmap_mask"] = (torch.arange(K)[None] < batch["gmap_lens"][:, None]).to(torch.uint8)
    cpu["gmap_step_ids"] = batch["gmap_step_ids"].reshape(-1).to(torch.int32)

    # ---- map-node aggregation: visited node <- fused pano of its step; unvisited <- mean of cand views
    e_embed, e_fused = [], []
    first_row = np.concatenate([[0], np.cumsum(step_lens)]).astype(np.int64)
    last_rows = first_row[1:] - 1
    for b in range(B):
        vis, unv = {}, {}
        for t in range(step_lens[b]):
            row = int(first_row[b]) + t
            vis[batch["traj_vpids"][b][t]] = row
            for j, c in enumerate(batch["traj_cand_vpids"][b][t]):
                unv.setdefault(c, []).append(row * V + j)
        for k, vp in enumerate(batch["gmap_vpids"][b]):
            if k == 0:
                continue
            if vp in vis:
                e_fused.append((b * K + k, vis[vp], 1.0))
            else:
                srcs = unv[vp]
                for s in srcs:
                    e_embed.append((b * K + k, s, 1.0 / len(srcs)))
    plan_csr = {}
    plan_csr["gmap_from_embed"] = csr_pair(e_embed, B * K, Np * V)
    plan_csr["gmap_from_fused"] = csr_pair(e_fused, B * K, Np)
    # ---- current-viewpoint tokens: [stop] + the last step's 36 views
    e_vp = []
    for b in range(B):
        for j in range(Vp - 1):
            e_vp.append((b * Vp + 1 + j, int(last_rows[b]) * V + j, 1.0))
    plan_csr["vp_from_embed"] = csr_pair(e_vp, B * Vp, Np * V)
    vp_lens = view_lens[torch.from_numpy(last_rows)] + 1
    cpu["vp_mask"] = (torch.arange(Vp)[None] < vp_lens[:, None]).to(torch.uint8)
    # ---- first-token (CLS / [stop]) row selections
    plan_csr["g0"] = csr_pair([(b, b * K, 1.0) for b in range(B)], B, B * K)
    plan_csr["v0"] = csr_pair([(b, b * Vp, 1.0) for b in range(B)], B, B * Vp)
    plan_csr["t0"] = csr_pair([(b, b * L, 1.0) for b in range(B)], B, B * L)

    if task in ("sap", "cfp"):
        visited = batch["gmap_visited_masks"]
        cpu["gmask"] = ((~visited) & cpu["gmap_mask"].bool()).to(torch.uint8)
        nav_last = batch["traj_nav_types"][torch.from_numpy(last_rows)] == 1
        cpu["lmask"] = torch.cat([torch.ones(B, 1, dtype=torch.bool), nav_last], 1)[:, :Vp].to(torch.uint8)
        fsrc = np.full((B, K), -1, np.int32)
        bwmask = np.zeros((B, Vp), np.uint8)
        vis_host = visited.tolist()          # (indexing the tensor element by element cost 3.6 ms per batch)
        for b in range(B):
            vis_set = set(vp for j, vp in enumerate(batch["gmap_vpids"][b]) if vis_host[b][j])
            tmp = {}
            for j, c in enumerate(batch["traj_cand_vpids"][b][-1]):
                if c in vis_set:
                    bwmask[b, j + 1] = 1
                else:
                    tmp[c] = j + 1
            fsrc[b, 0] = 0
            for j, vp in enumerate(batch["gmap_vpids"][b]):
                if j > 0 and vp not in vis_set:
                    fsrc[b, j] = tmp[vp] if vp in tmp else -2
        cpu["fsrc"] = torch.from_numpy(fsrc)
        cpu["bwmask"] = torch.from_numpy(bwmask)
        cpu["global_act_labels"] = batch["global_act_labels"].to(torch.int32)
        cpu["local_act_labels"] = batch["local_act_labels"].to(torch.int32)
        cpu["arange_b"] = torch.arange(B, dtype=torch.int32)
    if task == "mlm":
        lab = batch["txt_labels"].reshape(-1)
        sel = torch.nonzero(lab != -1).reshape(-1)
        n_mask = int(sel.numel())
        nm_rows = n_mask if pad is None else int(pad[0]["n_mask"])      # padded rows: no source row, label -1 (ignored by the loss kernel)
        plan_csr["mlm_rows"] = csr_pair([(i, int(r), 1.0) for i, r in enumerate(sel.tolist())], nm_rows, B * L)
        cpu["mlm_labels"] = torch.cat([lab[sel].to(torch.int32), torch.full((nm_rows - n_mask,), -1, dtype=torch.int32)])
        if pad is not None:
            cpu["mlm_row_w"] = torch.full((nm_rows,), 1.0 / max(n_mask, 1), dtype=torch.float32)      # the loss's 1 / n_mask, per row

    n_mrc = 0
    if task == "mrc":       # masked views of the current viewpoint: row b*Vp + 1 + v of the local encoder output ([stop] at 0)
        mm = batch["vp_view_mrc_masks"]
        sel = torch.nonzero(mm)                      # row-major == boolean-mask order of _compute_masked_hidden
        n_true = int(sel.shape[0])
        n_mrc = n_true if pad is None else int(pad[0]["n_mask"])      # padded rows: no source row, an all-zero target (no loss, no gradient)
        plan_csr["mrc_rows"] = csr_pair([(i, int(b_) * Vp + 1 + int(v_), 1.0) for i, (b_, v_) in enumerate(sel.tolist())], n_mrc, B * Vp)
        tg = batch["vp_view_probs"][mm].float()
        cpu["mrc_targets"] = torch.cat([tg, tg.new_zeros(n_mrc - n_true, tg.shape[1])]).contiguous()
        if pad is not None:
            cpu["mrc_row_w"] = torch.full((n_mrc,), 1.0 / max(n_true, 1), dtype=torch.float32)         # the loss's 1 / n_rows, per row

    if pad is not None:
        caps = dict(gmap_from_embed=Np * V, gmap_from_fused=B * K, mlm_rows=int(pad[0]["n_mask"]), mrc_rows=int(pad[0]["n_mask"]))
        for name, cap in caps.items():
            if name in plan_csr:
                plan_csr[name] = _pad_csr(plan_csr[name], cap)
    meta = {}
    # value ranges of everything the kernels use as a table index, taken on the host copies (the device never bounds-checks:
    # an out-of-range id would be an out-of-bounds read on the GPU) -- validated against the model config by check_plan()
    meta["limits"] = dict(txt_id=int(batch["txt_ids"].max()), txt_id_min=int(batch["txt_ids"].min()),
                          step_id=int(batch["gmap_step_ids"].max()), step_id_min=int(batch["gmap_step_ids"].min()),
                          nav_type=int(batch["traj_nav_types"].max()), nav_type_min=int(batch["traj_nav_types"].min()))
    meta.update(B=B, L=L, K=K, Vp=Vp, Np=Np, V=V, last_rows=last_rows,
                n_mask=(int(cpu["mlm_labels"].numel()) if task == "mlm" else 0), n_mrc=n_mrc,
                txt_tokens=int(txt_lens.sum()), gmap_nodes=int(batch["gmap_lens"].sum()), traj_steps=Np,
                lens=dict(txt=txt_lens.tolist(), gmap=batch["gmap_lens"].tolist(), steps=list(step_lens)))
    if pad is not None:
        meta["traj_steps"] = int(sum(step_lens))
        meta["true"] = dict(pad[1])
        meta["bucket"] = dict(pad[0])
        # (ragged, so it rides in the record's pickled header rather than in the bucket-shaped buffer; stream_graph.StreamStep hands it to
        # the data-parallel exchange of the word-embedding rows)
        meta["touched_ids"] = np.unique(batch["txt_ids"].numpy()).astype(np.int64)
    return dict(cpu=cpu, csr=plan_csr, meta=meta)


def check_plan(plan, cfg):
    """Raise (on the host, before any launch) if a batch would index a table out of range."""
    lim = plan.get("limits")
    if lim is None or plan.get("_checked") is cfg:
        return
    def bad(msg):
        raise ValueError(f"batch does not fit the model config: {msg}")
    if lim["txt_id_min"] < 0 or lim["txt_id"] >= cfg.vocab_size:
        bad(f"txt_ids in [{lim['txt_id_min']}, {lim['txt_id']}] vs vocab_size {cfg.vocab_size}")
    if plan["L"] + 2 > cfg.max_position_embeddings:
        bad(f"{plan['L']} tokens need position rows up to {plan['L'] + 1}, max_position_embeddings is {cfg.max_position_embeddings}")
    if lim["step_id_min"] < 0 or lim["step_id"] >= cfg.max_action_steps:
        bad(f"gmap_step_ids up to {lim['step_id']} vs max_action_steps {cfg.max_action_steps}")
    if lim["nav_type_min"] < 0 or lim["nav_type"] > 2:
        bad(f"traj_nav_types in [{lim['nav_type_min']}, {lim['nav_type']}] (nav_type_embedding has 3 rows)")
    if plan["V"] > 64:
        bad(f"{plan['V']} view tokens per panorama (the panorama kernels serve <= 64)")
    plan["_checked"] = cfg
