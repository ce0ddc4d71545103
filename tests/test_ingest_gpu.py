"""Feature ingest on the device: packed HBM feature table + `magic_view_gather` vs the reference's
get_traj_pano_fts output (fixture tests/golden/ingest.pt) and vs torch indexing at full R2R size (-m gpu)."""
import os

import numpy as np
import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import feature_table as FT
from magic_amd.host import ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gather_reproduces_reference_token_tensors(dtype):
    fx = torch.load(os.path.join(GOLD, "ingest.pt"), weights_only=False)
    D, scan, cands = fx["D"], fx["scan"], fx["cands"]
    keys = sorted(fx["store"])
    ft = FT.FeatureTable.from_arrays(keys, [fx["store"][k].numpy() for k in keys], device=DEV, dtype=dtype, image_feat_size=D)
    b = ft.batch_indices([scan] * 3, fx["paths"], lambda sc, vp: cands[f"{sc}_{vp}"])
    out = torch.full((9, 37, D), 7.0, dtype=dtype, device=DEV)
    ft.gather(b["vp_row"], b["order"], out=out)
    flat = sum([o["fts"] for o in fx["outs"]], [])
    for p, x in enumerate(flat):
        n = x.shape[0]
        assert torch.equal(out[p, :n].cpu(), x.to(dtype)), p          # bit-exact (a pure copy in the table's dtype)
        assert not out[p, n:].any()                                   # padded view slots are zero, as pad_tensors leaves them
    assert b["traj_vp_view_lens"].tolist() == [x.shape[0] for x in flat]


def test_gather_at_r2r_scale_matches_torch_indexing_and_is_a_permutation():
    """10 567 viewpoints x 36 x 768 bf16 (= the whole R2R feature file, 584 MB) resident; a B=48 batch worth of panoramas."""
    g = torch.Generator().manual_seed(0)
    n, D, Np = 10567, 768, 290
    table = torch.randn(n, 36, D, generator=g).to(torch.bfloat16).to(DEV)
    rows = torch.randint(0, n, (Np,), generator=g).to(torch.int32)
    order = torch.stack([torch.randperm(36, generator=g) for _ in range(Np)]).to(torch.int32)
    order[5, 30:] = -1
    ft = FT.FeatureTable([str(i) for i in range(n)], table)
    out = ft.gather(rows, order)
    o = order.to(DEV).long().clamp(min=0)
    want = table[rows.to(DEV).long()[:, None], o]
    want[5, 30:] = 0
    assert torch.equal(out, want)
    # size-independent property: gathering with the inverse order undoes the permutation
    inv = torch.argsort(order[0].long()).to(torch.int32)
    t2 = FT.FeatureTable(["x"], out[0:1].contiguous())
    back = t2.gather(torch.zeros(1, dtype=torch.int32), inv[None])
    assert torch.equal(back[0], table[rows[0].long()])


def test_gather_rejects_bad_arguments():
    from magic_amd.host import lib as L
    t = torch.zeros(2, 36, 20, dtype=torch.bfloat16, device=DEV)          # D % 8 != 0
    with pytest.raises(L.MagicHipError):
        O.view_gather(t, torch.zeros(1, dtype=torch.int32, device=DEV), torch.zeros(1, 36, dtype=torch.int32, device=DEV),
                      torch.empty(1, 36, 20, dtype=torch.bfloat16, device=DEV))


def test_model_accepts_index_only_batches():
    """A batch that carries (view_table, traj_vp_row, traj_view_order) instead of traj_view_img_fts gives the same forward."""
    from magic_amd.host import synth
    from magic_amd.host.config import make_config
    from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining
    kw = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, num_l_layers=1, num_x_layers=1, num_pano_layers=1)
    model = GlocalTextPathCMTPreTraining(make_config(128, **kw), device=DEV, compute_dtype=torch.bfloat16, seed=3).eval()
    batch = synth.make_batch("sap", batch_size=4, seed=2, vocab=300, min_len=6, max_len=10, min_steps=2, max_steps=3)
    with torch.no_grad():
        want = model(batch, "sap", compute_loss=False)
    feats = batch["traj_view_img_fts"]                       # [Np, 36, 768] fp32: pretend every panorama is its own viewpoint,
    Np = feats.shape[0]                                      # stored in canonical view order, presented permuted
    g = torch.Generator().manual_seed(1)
    perm = torch.stack([torch.randperm(36, generator=g) for _ in range(Np)])
    canon = torch.empty_like(feats)
    canon[torch.arange(Np)[:, None], perm] = feats           # canon[p, perm[p, j]] = feats[p, j]
    ft = FT.FeatureTable.from_arrays([f"s_{i}" for i in range(Np)], [canon[i].numpy() for i in range(Np)], device=DEV)
    b2 = {k: v for k, v in batch.items() if k != "traj_view_img_fts"}
    b2.update(view_table=ft, traj_vp_row=torch.arange(Np, dtype=torch.int32), traj_view_order=perm.to(torch.int32))
    with torch.no_grad():
        got = model(b2, "sap", compute_loss=False)
    for k in ("global_logits", "local_logits", "fused_logits"):
        assert torch.equal(got[k], want[k]), k
