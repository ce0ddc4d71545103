"""Streaming path (-m gpu): batches that went through the DataLoader-worker packing (host/loader.PlanCollate -> pack),
the copy-stream prefetcher and unpack give exactly the step the resident path gives -- same losses, same parameter update -- for a
feature-carrying batch and for an index-only (HBM feature table) batch."""
import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import synth
from magic_amd.host.config import make_config
from magic_amd.host.feature_table import FeatureTable
from magic_amd.host.loader import DevicePrefetcher, PlanCollate
from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining
from magic_amd.host.plan import build_plan
from magic_amd.host.trainer import PretrainStep
from tests.test_model_gpu import KDL

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
KW = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, num_l_layers=2, num_x_layers=1, num_pano_layers=1)


def _trainer(seed):
    t = GlocalTextPathCMTPreTraining(make_config(256, role="teacher", **KW), device=DEV, compute_dtype=torch.float32, seed=0)
    s = GlocalTextPathCMTPreTraining(make_config(128, role="student", teacher_hidden_size=256, kdl=KDL, **KW), device=DEV,
                                     compute_dtype=torch.float32, seed=seed)
    return PretrainStep(s, t, lr=1e-3, warmup_steps=1, num_train_steps=100), s


def _batches():
    out = []
    for i, task in enumerate(("sap", "mlm", "cfp", "sap")):
        out.append((task, synth.make_batch(task, batch_size=5, seed=7, step=i, vocab=300, min_len=6, max_len=12, min_steps=2, max_steps=4)))
    return out


@pytest.mark.parametrize("ingest", ["host", "table"])
def test_streamed_steps_equal_resident_steps(ingest):
    rw = [1.1, 0.9, 1.0, 1.2, 0.8]
    batches = _batches()
    table = None
    if ingest == "table":          # every panorama becomes a row of an HBM table, presented through a view permutation
        g = torch.Generator().manual_seed(0)
        rows, new = [], []
        for task, b in batches:
            feats = b.pop("traj_view_img_fts")
            Np, V = feats.shape[:2]
            perm = torch.stack([torch.randperm(36, generator=g) for _ in range(Np)])
            canon = torch.zeros(Np, 36, feats.shape[2])
            canon[torch.arange(Np)[:, None], perm] = feats[:, :36]
            b["traj_vp_row"] = torch.arange(len(rows), len(rows) + Np, dtype=torch.int32)
            order = torch.full((Np, V), -1, dtype=torch.int32)
            order[:, :36] = perm.to(torch.int32)
            b["traj_view_order"] = order
            rows += [canon[i] for i in range(Np)]
            new.append((task, b))
        batches = new
        table = FeatureTable([str(i) for i in range(len(rows))], torch.stack(rows).to(DEV))
    # resident path
    tr_a, s_a = _trainer(1)
    want = []
    for task, b in batches:
        bd = synth.batch_to(b, DEV)
        if table is not None:
            bd["view_table"] = table
        out = tr_a.step(bd, task, rw=rw, plan=build_plan(b, task, DEV))
        want.append(float(out["loss"].detach()))
    # streamed path: worker-side collate wrapper -> packed record -> prefetcher
    tr_b, s_b = _trainer(1)
    recs = [(task, PlanCollate(lambda inp, b=b: b, task)([None])) for task, b in batches]
    got = []
    for task, bd, plan in DevicePrefetcher(recs, DEV):
        if table is not None:
            bd["view_table"] = table
        out = tr_b.step(bd, task, rw=rw, plan=plan)
        got.append(float(out["loss"].detach()))
    torch.cuda.synchronize()
    assert got == pytest.approx(want, rel=1e-6, abs=1e-7)
    d = (s_a.store.flat - s_b.store.flat).abs().max().item()
    assert d <= 5e-5, d            # 4 AdamW steps of 1e-3; fp32-atomic summation order differs run to run
