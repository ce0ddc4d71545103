"""SURVEY f-3, data-parallel form (-m gpu; 2 ranks on one card over gloo): every rank streams its OWN bucket-padded records through
StreamStep -- `run` = teacher one batch ahead + the student's step as three graphs with the gradient-bucket exchanges issued between them +
the optimizer's graph; `step` = one graph up to the end of the backward + the monolithic exchange.  Both must equal the eager data-parallel
step on the exact (unpadded) batches: losses per rank, weights after AdamW, and bitwise-equal replicas."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import magic_amd  # noqa: F401

pytestmark = pytest.mark.gpu
SCHED = [("sap", 0), ("mlm", 1), ("cfp", 2), ("sap", 3), ("mrc", 4), ("sap", 6), ("cfp", 5)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from magic_amd.host import synth
        from magic_amd.host.loader import pack_bucketed
        from magic_amd.host.plan import build_plan
        from magic_amd.host.stream_graph import StreamStep
        from magic_amd.host.trainer import PretrainStep
        from tests.test_model_gpu import RW, build
        dev = "cuda"
        torch.cuda.set_device(0)
        rw = torch.tensor(RW, device=dev)
        mk = lambda task, step: synth.make_batch(task, batch_size=4, seed=31 + rank, step=step, vocab=600, min_len=8, max_len=15, min_steps=2,
                                                 max_steps=3)                     # ranks see different data (and land in different buckets)
        batches = [mk(task, step) for task, step in SCHED]

        def trainer():
            _, _, g_t, g_s = build(torch.bfloat16, pretrain_tasks={"mlm", "mrc", "sap", "cfp"})          # same seeds on every rank = broadcast parameters
            g_s.keep_mlm_logits = False
            tr = PretrainStep(g_s, g_t, lr=5e-5, warmup_steps=2, num_train_steps=40, sparse_embedding_rows=4 * 80)
            assert tr.sync.world == 2 and tr.sync.overlap
            from magic_amd.host import ops as O
            assert tr.sync.card_shared and not O.ENC_ROW_SPLIT and not O.XENC_ROW_SPLIT, "both ranks sit on the one card: row-split launches must be off"
            return g_s, tr
        sA, tA = trainer()
        want = []
        for (task, _), b in zip(SCHED, batches):
            o = tA.step(synth.batch_to(b, dev), task, rw=rw, plan=build_plan(b, task, dev))
            want.append({k: float(o[k]) for k in ("loss", "supervised_loss", "kdl_loss")})
        res = {}
        for form in ("run", "step"):
            sB, tB = trainer()
            ss = StreamStep(tB, rw=rw)
            recs = [(task, pack_bucketed(b, task)) for (task, _), b in zip(SCHED, batches)]
            got = []
            if form == "run":
                for out, meta in ss.run(iter(recs)):
                    torch.cuda.synchronize()
                    got.append({k: float(out[k]) for k in ("loss", "supervised_loss", "kdl_loss")})
            else:
                for task, rec in recs:
                    out, meta = ss.step(task, rec)
                    torch.cuda.synchronize()
                    got.append({k: float(out[k]) for k in ("loss", "supervised_loss", "kdl_loss")})
            assert tB.global_step == len(SCHED)
            tB.check_health()
            worst = max(abs(g[k] - w[k]) / max(abs(w[k]), 1e-6) for g, w in zip(got, want) for k in g)
            wa, wb = sA.store.flat, sB.store.flat
            every = [torch.empty_like(wb) for _ in range(world)]
            dist.all_gather(every, wb)
            res[form] = dict(worst_loss_rel=worst, w_rel=float((wa - wb).norm() / wa.norm()),
                             m_cos=float(torch.nn.functional.cosine_similarity(sA.store.m, sB.store.m, dim=0)),
                             replicas_equal=bool(torch.equal(every[0], every[1])), captures=ss.captures,
                             graphs=3 if (form == "run" and next(iter(ss.cache.values())).cs.graph3 is not None) else 1)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_streamed_records_under_graph_replay_data_parallel_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    out, t0 = [], time.time()
    while len(out) < world and time.time() - t0 < 800:
        try:
            out.append(q.get(timeout=5))
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):      # a rank died: do not wait for its result
                break
    if len(out) < world:
        for p in procs:
            if p.is_alive():
                p.terminate()
        pytest.fail(f"a rank failed (exit codes {[p.exitcode for p in procs]}): see its traceback above")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in out:
        print(rank, res)
        assert res["run"]["graphs"] == 3, "the data-parallel `run` form replays the three backward-cut graphs"
        for form, r in res.items():
            assert r["worst_loss_rel"] < 5e-3, (rank, form, r)
            assert r["w_rel"] < 1e-4 and r["m_cos"] > 0.999, (rank, form, r)
            assert r["replicas_equal"], f"rank {rank} {form}: replicas hold different weights after streamed data-parallel steps"
