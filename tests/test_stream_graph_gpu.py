"""Streamed batches under HIP-graph replay (host/stream_graph.py, -m gpu): a step replayed on the bucket-padded record of a batch must equal the
eager step on the exact (unpadded) batch -- every loss term and the weights after AdamW -- and a second batch of the same bucket must reuse
the captured graph."""
import pytest
import torch

import magic_amd  # noqa: F401
import bench
from magic_amd.host import synth
from magic_amd.host.bucket import bucket_of
from magic_amd.host.loader import pack_bucketed
from magic_amd.host.plan import build_plan
from magic_amd.host.stream_graph import StreamStep

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
RW = [1.3, 0.7, 1.1, 0.9, 1.0]


def _close(a, b, rtol):
    a, b = float(a), float(b)
    return abs(a - b) <= rtol * max(abs(b), 1e-6)


def test_bucketed_graph_step_equals_the_exact_eager_step():
    _, _, _, sA, tA = bench.build_models(torch.bfloat16, DEV, 0.0, 1, 16)
    _, _, _, sB, tB = bench.build_models(torch.bfloat16, DEV, 0.0, 1, 16)
    assert torch.equal(sA.store.flat, sB.store.flat)
    rw = torch.tensor(RW, dtype=torch.float32, device=DEV)
    ss = StreamStep(tB, rw=rw)
    seen = {}
    for i, (task, step) in enumerate((("sap", 0), ("mlm", 1), ("cfp", 2), ("sap", 3), ("mlm", 4), ("cfp", 5), ("sap", 6))):
        b = synth.make_batch(task, batch_size=16, seed=4242, step=step)
        key = (task, tuple(sorted(bucket_of(b, task).items())))
        plan = build_plan(b, task, DEV)
        outA = tA.step(synth.batch_to(b, DEV), task, rw=rw, plan=plan)
        before = ss.captures
        outB, meta = ss.step(task, pack_bucketed(b, task))
        torch.cuda.synchronize()
        assert ss.captures == before + (0 if key in seen else 1), "a bucket is captured once"
        seen[key] = True
        assert meta["true"]["L"] == b["txt_ids"].shape[1] <= meta["L"] == 80
        for k in ("loss", "supervised_loss", "kdl_loss"):
            assert _close(outB[k], outA[k], 2e-3), (task, step, k, float(outB[k]), float(outA[k]))
        for k, v in outA["kdl_terms"].items():
            assert _close(outB["kdl_terms"][k], v, 5e-3) or abs(float(v)) < 1e-7, (task, step, k, float(outB["kdl_terms"][k]), float(v))
        wa, wb = sA.store.flat, sB.store.flat
        rel = ((wa - wb).norm() / wa.norm()).item()
        worst = max(abs(float(outB["kdl_terms"][k]) - float(v)) / max(abs(float(v)), 1e-9) for k, v in outA["kdl_terms"].items() if abs(float(v)) > 1e-7)
        print(f"{task} step {step}: loss {float(outA['loss']):.6f} vs {float(outB['loss']):.6f}, worst distillation term rel. diff {worst:.1e}, weights rel. diff {rel:.1e}, "
              f"true sizes {meta['true']} in bucket {meta['bucket']}")
        assert rel < 2e-5, (task, step, rel)
        # the Adam first moment is a running mean of the (clipped) gradients: it compares the two backward passes far more sharply than the weights
        ma, mb = sA.store.m, sB.store.m
        mrel = ((ma - mb).norm() / ma.norm()).item()
        assert mrel < 2e-2 and torch.nn.functional.cosine_similarity(ma, mb, dim=0).item() > 0.9998, (task, step, mrel)
    assert len(seen) < 7 or ss.captures <= 7


def test_mrc_records_replay_under_their_bucket_graph():
    """the fourth proxy task (MrcDataset, tasks.py:189-310; listed in `tasks` it creates the image_classifier head): masked-view rows padded
    to the bucket with all-zero target distributions, the true 1 / n_rows as per-row weights (magic_softkl_rows row_w)"""
    from magic_amd.host.trainer import PretrainStep
    from tests.test_model_gpu import build

    def trainer():
        _, _, g_t, g_s = build(torch.bfloat16, pretrain_tasks={"mlm", "mrc", "sap", "cfp"})
        g_s.keep_mlm_logits = False
        return g_s, PretrainStep(g_s, g_t, lr=5e-5, warmup_steps=2, num_train_steps=40)
    (sA, tA), (sB, tB) = trainer(), trainer()
    rw = torch.tensor(RW, dtype=torch.float32, device=DEV)
    ss = StreamStep(tB, rw=rw)
    seen = set()
    for step in range(5):
        b = synth.make_batch("mrc", batch_size=8, seed=99, step=step, vocab=600, min_len=8, max_len=19, min_steps=2, max_steps=4)
        bk = bucket_of(b, "mrc")
        n = int(b["vp_view_mrc_masks"].sum())
        outA = tA.step(synth.batch_to(b, DEV), "mrc", rw=rw, plan=build_plan(b, "mrc", DEV))
        before = ss.captures
        key = tuple(sorted(bk.items()))
        outB, meta = ss.step("mrc", pack_bucketed(b, "mrc"))
        torch.cuda.synchronize()
        assert ss.captures == before + (0 if key in seen else 1)
        seen.add(key)
        assert meta["true"]["n_mask"] == n <= meta["n_mrc"] == bk["n_mask"]
        for k in ("loss", "supervised_loss", "kdl_loss"):
            assert _close(outB[k], outA[k], 2e-3), (step, k, float(outB[k]), float(outA[k]))
        rel = ((sA.store.flat - sB.store.flat).norm() / sA.store.flat.norm()).item()
        print(f"mrc step {step}: {n} masked views in a bucket of {bk['n_mask']}, loss {float(outA['loss']):.6f} vs {float(outB['loss']):.6f}, weights rel. diff {rel:.1e}")
        assert rel < 2e-5
    assert torch.nn.functional.cosine_similarity(sA.store.m, sB.store.m, dim=0).item() > 0.9995
    assert len(seen) < 5, "five batches, fewer buckets: at least one replay reused a captured graph"


def test_teacher_one_batch_ahead_on_streamed_records_equals_the_exact_eager_steps():
    """StreamStep.run: split teacher / student graphs, two record slots per bucket, the teacher's forward for batch i+1 under the student's step on
    batch i -- same losses as stepping eagerly through the exact batches"""
    _, _, _, sA, tA = bench.build_models(torch.bfloat16, DEV, 0.0, 1, 16)
    _, _, _, sB, tB = bench.build_models(torch.bfloat16, DEV, 0.0, 1, 16)
    rw = torch.tensor(RW, dtype=torch.float32, device=DEV)
    ss = StreamStep(tB, rw=rw)
    sched = [("sap", 0), ("sap", 3), ("mlm", 1), ("cfp", 2), ("sap", 6), ("mlm", 4), ("cfp", 5), ("cfp", 2)]
    batches = [synth.make_batch(task, batch_size=16, seed=4242, step=step) for task, step in sched]
    feed = ((task, pack_bucketed(b, task)) for (task, _), b in zip(sched, batches))
    got = []
    for out, meta in ss.run(feed):
        torch.cuda.synchronize()
        got.append({k: float(out[k]) for k in ("loss", "supervised_loss", "kdl_loss")})
    assert len(got) == len(sched)
    for (task, step), b, g in zip(sched, batches, got):
        outA = tA.step(synth.batch_to(b, DEV), task, rw=rw, plan=build_plan(b, task, DEV))
        for k in ("loss", "supervised_loss", "kdl_loss"):
            assert _close(g[k], outA[k], 2e-3), (task, step, k, g[k], float(outA[k]))
    ma, mb = sA.store.m, sB.store.m
    assert torch.nn.functional.cosine_similarity(ma, mb, dim=0).item() > 0.9995
    assert ((sA.store.flat - sB.store.flat).norm() / sA.store.flat.norm()).item() < 2e-5


def test_bucket_graphs_are_bounded_by_an_lru():
    _, _, _, sA, tA = bench.build_models(torch.bfloat16, DEV, 0.0, 1, 16)
    _, _, _, sB, tB = bench.build_models(torch.bfloat16, DEV, 0.0, 1, 16)
    rw = torch.tensor(RW, dtype=torch.float32, device=DEV)
    ss = StreamStep(tB, rw=rw, max_graphs=1)
    for task, step in (("sap", 0), ("mlm", 1), ("sap", 3)):          # sap, mlm, sap: the second sap batch finds its graph evicted and recaptures
        b = synth.make_batch(task, batch_size=16, seed=4242, step=step)
        outA = tA.step(synth.batch_to(b, DEV), task, rw=rw, plan=build_plan(b, task, DEV))
        outB, _ = ss.step(task, pack_bucketed(b, task))
        torch.cuda.synchronize()
        assert len(ss.cache) == 1
        assert _close(outB["loss"], outA["loss"], 2e-3)
    assert ss.captures == 3
