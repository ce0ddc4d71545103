"""SURVEY f-3: the bucket-padded packed record a DataLoader worker ships (loader.pack_bucketed: collate -> pad to the shape bucket -> host half of
the index plan -> ONE uint8 storage) against the REFERENCE's own collates: on the valid extent the record's tensors equal the tensors
`pretrain_src/data/tasks.py:{sap,cfp,mlm}_collate` produced for the same items (tests/golden/collate.pt, minted by running the reference),
and everything outside it is inert padding (0; ignored label -1).  VERDICT r2: the stream tests compared the engine with itself only."""
import os

import numpy as np
import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import synth
from magic_amd.host.bucket import bucket_of
from magic_amd.host.loader import pack_bucketed, unpack

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PAD = {"txt_labels": -1}


@pytest.mark.parametrize("task", ["sap", "cfp", "mlm"])
def test_bucket_padded_record_reproduces_reference_collate_on_its_valid_extent(task):
    fx = torch.load(os.path.join(GOLDEN, "collate.pt"), weights_only=False)
    want = fx[task]
    batch = synth.collate(fx["samples"], task, **({"rng": np.random.default_rng(5)} if task == "mlm" else {}))
    bk = bucket_of(batch, task)
    assert bk["L"] >= batch["txt_ids"].shape[1] and bk["K"] % 8 == 0 and bk["Np"] % 32 == 0
    got, plan = unpack(pack_bucketed(batch, task), torch.device("cpu"))
    checked = 0
    for k, v in want.items():
        if not torch.is_tensor(v) or k not in got:
            continue
        g = got[k]
        assert g.dtype == v.dtype and g.dim() == v.dim() and all(a >= b for a, b in zip(g.shape, v.shape)), (k, g.shape, v.shape)
        inner = tuple(slice(0, s) for s in v.shape)
        assert torch.equal(g[inner], v), f"{task} {k}: the valid extent differs from the reference collate"
        mask = torch.ones(g.shape, dtype=torch.bool)
        mask[inner] = False
        assert (g[mask] == PAD.get(k, 0)).all(), f"{task} {k}: padding is not inert"
        checked += 1
    # tensors the record carries re-encoded in the index plan (int32 / uint8, flattened): same values on the valid extent
    B, L0, K0 = len(want["traj_step_lens"]), want["txt_ids"].shape[1], want["gmap_step_ids"].shape[1]
    Np0, V0 = want["traj_nav_types"].shape[:2]
    ids = plan["txt_ids"].view(B, bk["L"]).long()
    assert torch.equal(ids[:, :L0], want["txt_ids"]) and (ids[:, L0:] == 0).all()
    tm = plan["txt_mask"].view(B, bk["L"]).bool()
    assert torch.equal(tm.sum(1), want["txt_lens"]) and torch.equal(tm, torch.arange(bk["L"])[None] < want["txt_lens"][:, None])
    assert torch.equal(plan["view_lens"][:Np0].long(), want["traj_vp_view_lens"]) and (plan["view_lens"][Np0:] == 0).all()
    nt = plan["nav_types"].view(bk["Np"], -1).long()
    assert torch.equal(nt[:Np0, :V0], want["traj_nav_types"]) and (nt[Np0:] == 0).all()
    gs = plan["gmap_step_ids"].view(B, bk["K"]).long()
    assert torch.equal(gs[:, :K0], want["gmap_step_ids"]) and (gs[:, K0:] == 0).all()
    gm = plan["gmap_mask"].view(B, bk["K"]).bool()
    assert torch.equal(gm, torch.arange(bk["K"])[None] < want["gmap_lens"][:, None])          # padded map slots are masked keys
    pm = plan["pano_mask"].view(bk["Np"], -1).bool()
    assert torch.equal(pm[:Np0].sum(1), want["traj_vp_view_lens"]) and not pm[Np0:].any()        # dummy panoramas: no valid view
    assert plan["lens"]["txt"] == want["txt_lens"].tolist() and plan["lens"]["gmap"] == want["gmap_lens"].tolist()
    checked += 8
    if task == "mlm":           # masked positions: labels in row-major order of the reference's txt_labels, padded rows ignored (-1) with zero weight
        lab = want["txt_labels"]
        n = int((lab != -1).sum())
        assert plan["n_mask"] == bk["n_mask"] >= n and plan["true"]["n_mask"] == n
        assert torch.equal(plan["mlm_labels"][:n].long(), lab[lab != -1]) and (plan["mlm_labels"][n:] == -1).all()
        assert torch.allclose(plan["mlm_row_w"], torch.full((bk["n_mask"],), 1.0 / n))        # the TRUE 1 / n_mask on every row; padded rows carry the ignored label
        checked += 3
    assert checked >= 14, checked
    # the plan's true sizes are the reference batch's sizes, its static sizes the bucket's
    assert plan["B"] == len(want["traj_step_lens"]) and plan["L"] == bk["L"] and plan["K"] == bk["K"] and plan["Np"] == bk["Np"]
    assert plan["traj_steps"] == sum(want["traj_step_lens"])
    # two different batches of one bucket have byte-identical record LAYOUTS (one captured graph serves both)
    other = synth.collate(fx["samples"][::-1], task, **({"rng": np.random.default_rng(6)} if task == "mlm" else {}))
    if bucket_of(other, task) == bk:
        import pickle
        r1, r2 = pack_bucketed(batch, task), pack_bucketed(other, task)
        m1, m2 = pickle.loads(r1["blob"])[0], pickle.loads(r2["blob"])[0]
        assert r1["buf"].numel() == r2["buf"].numel() and m1 == m2
