"""Data-parallel path on CPU with gloo, world_size 2 (the N>1 path of bench.py / trainer.GradSync / broadcast_task).
One process per rank, rendezvous on 127.0.0.1."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import magic_amd  # noqa: F401


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from magic_amd.host.config import make_config
        from magic_amd.host.model_pretrain import pretrain_specs
        from magic_amd.host.params import ParamStore
        from magic_amd.host.trainer import GradSync, broadcast_task
        cfg = make_config(128, teacher_hidden_size=256, vocab_size=200, num_l_layers=1, num_x_layers=1, num_pano_layers=1)
        store = ParamStore(pretrain_specs(cfg), "cpu", torch.float32, seed=100 + rank)      # ranks start different
        dist.broadcast(store.flat, src=0)                                                   # DDP-ctor semantics (utils/misc.py:62-63)
        ref = ParamStore(pretrain_specs(cfg), "cpu", torch.float32, seed=100)
        same_params = bool(torch.equal(store.flat, ref.flat))
        g = torch.Generator().manual_seed(7 + rank)
        store.grad.copy_(torch.randn(store.total, generator=g))
        mine = store.grad.clone()
        sync = GradSync(store, chunk_elems=10007)            # odd chunk size: exercises the chunk tail
        gscale = sync.all_reduce()
        others = [torch.randn(store.total, generator=torch.Generator().manual_seed(7 + r)) for r in range(world)]
        want_sum = sum(others)
        ok_sum = bool(torch.allclose(store.grad, want_sum, rtol=1e-6, atol=1e-6))
        ok_mean = bool(torch.allclose(store.grad * gscale, want_sum / world, rtol=1e-6, atol=1e-6))
        # unused-parameter semantics: a rank that did not touch a tensor contributes zeros, result stays consistent
        # unmodified-loop path: the end of loss.backward() averages the flat gradient buffer (trainer.auto_sync), DDP semantics
        from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining
        from magic_amd.host.trainer import auto_sync
        model = GlocalTextPathCMTPreTraining(cfg, device="cpu", compute_dtype=torch.float32, seed=5)
        model.store.grad.copy_(torch.randn(model.store.total, generator=torch.Generator().manual_seed(40 + rank)))
        auto_sync(model)
        want_mean = sum(torch.randn(model.store.total, generator=torch.Generator().manual_seed(40 + r)) for r in range(world)) / world
        ok_auto = bool(torch.allclose(model.store.grad, want_mean, rtol=1e-6, atol=1e-6))
        p0 = next(iter(model.parameters()))
        ok_auto = ok_auto and p0.grad.data_ptr() == model.store.grad.data_ptr()          # still views of the flat buffer
        model.auto_grad_sync = False
        keep = model.store.grad.clone()
        auto_sync(model)
        ok_auto = ok_auto and bool(torch.equal(keep, model.store.grad))
        ok_mean = ok_mean and ok_auto
        # ---- bucketed exchange in backward-completion order == monolithic; sparse word-embedding rows == dense table --------
        from magic_amd.host.trainer import EMB_TABLE
        ok_bucket = True
        for sparse in (False, True):
            st2 = ParamStore(pretrain_specs(cfg), "cpu", torch.float32, seed=1)
            gen = torch.Generator().manual_seed(900 + rank)
            st2.grad.copy_(torch.randn(st2.total, generator=gen))
            off, n, (R, H) = st2.offsets[EMB_TABLE]
            touched = torch.randperm(R, generator=gen)[:17 + 5 * rank]            # ranks touch different, overlapping row sets
            if sparse:
                tab = st2.grad[off:off + n].view(R, H)
                keep = tab[touched].clone()
                tab.zero_()
                tab[touched] = keep
            mono = st2.grad.clone()
            dist.all_reduce(mono)
            sy = GradSync(st2, chunk_elems=4099, overlap=True, sparse_rows_cap=40 if sparse else None)
            b0, b1, b2 = sy.buckets         # heads + cross-modal | top text blocks + panorama blocks (cut two rounds into that half) | the rest
            covered = sorted(b0 + b1 + b2)
            ok_bucket = ok_bucket and covered[0][0] == 0 and covered[-1][1] == st2.total and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
            ok_bucket = ok_bucket and b0[0][0] == st2.offsets["bert.global_encoder.gmap_pos_embeddings.0.weight"][0]
            # the middle bucket holds exactly the last text block and the last panorama block of this 1 + 1-layer model, in both decay groups
            inb = lambda name, bk: any(lo <= st2.offsets[name][0] and st2.offsets[name][0] + st2.offsets[name][1] <= hi for lo, hi in bk)
            ok_bucket = ok_bucket and all(inb(n_, b1) == (n_.startswith("bert.lang_encoder.layer.0.") or n_.startswith("bert.img_embeddings.pano_encoder.layer.0."))
                                          for n_ in st2.offsets)
            ok_bucket = ok_bucket and inb(EMB_TABLE, b2) and inb("bert.img_embeddings.img_linear.weight", b2)
            sy.reduce_bucket(0)
            first = st2.grad.clone()
            sy.reduce_bucket(1)
            second = st2.grad.clone()
            sy.reduce_bucket(2, touched if sparse else None)
            sy.finish()
            for lo, hi in b1:                                   # the middle bucket was final after its own call
                ok_bucket = ok_bucket and bool(torch.allclose(second[lo:hi], mono[lo:hi], rtol=1e-6, atol=1e-6))
            ok_bucket = ok_bucket and bool(torch.allclose(st2.grad, mono, rtol=1e-6, atol=1e-6))
            lo, hi = b0[0]
            ok_bucket = ok_bucket and bool(torch.allclose(first[lo:hi], mono[lo:hi], rtol=1e-6, atol=1e-6))     # bucket 0 was final after its own call
            # replicas are BITWISE identical after the exchange (an all-reduce guarantees it; the sparse path sums rows in rank order)
            every = [torch.empty_like(st2.grad) for _ in range(world)]
            dist.all_gather(every, st2.grad)
            ok_bucket = ok_bucket and all(bool(torch.equal(every[0], e)) for e in every[1:])
        # round 6: the STATIC padded form a captured graph holds for streamed batches (GradSync._sparse_rows_padded): ids first, -1 pads behind them;
        # row 0 touched on rank 0 only (the pads gather and clear row 0: its own contribution must still come back), an all-pad buffer on the last rank
        st4 = ParamStore(pretrain_specs(cfg), "cpu", torch.float32, seed=1)
        gen = torch.Generator().manual_seed(970 + rank)
        st4.grad.copy_(torch.randn(st4.total, generator=gen))
        off, n, (R, H) = st4.offsets[EMB_TABLE]
        touched = torch.randperm(R - 1, generator=gen)[:11 + 3 * rank] + 1
        if rank == 0:
            touched = torch.cat([torch.zeros(1, dtype=torch.int64), touched])
        if rank == world - 1 and world > 2:
            touched = touched[:0]
        tab = st4.grad[off:off + n].view(R, H)
        keep = tab[touched].clone()
        tab.zero_()
        tab[touched] = keep
        mono = st4.grad.clone()
        dist.all_reduce(mono)
        sy = GradSync(st4, chunk_elems=4099, overlap=True, sparse_rows_cap=40)
        ids_pad = torch.full((40,), -1, dtype=torch.int64)
        ids_pad[:touched.numel()] = touched
        sy.reduce_bucket(0)
        sy.reduce_bucket(1)
        sy.reduce_bucket(2, ids_pad, padded=True)
        sy.finish()
        ok_bucket = ok_bucket and bool(torch.allclose(st4.grad, mono, rtol=1e-6, atol=1e-6))
        every = [torch.empty_like(st4.grad) for _ in range(world)]
        dist.all_gather(every, st4.grad)
        ok_bucket = ok_bucket and all(bool(torch.equal(every[0], e)) for e in every[1:])
        # a bucket-padded plan carries an EMPTY id list (host/plan.py): that must mean "dense table", never "no rows"
        st3 = ParamStore(pretrain_specs(cfg), "cpu", torch.float32, seed=1)
        st3.grad.copy_(torch.randn(st3.total, generator=torch.Generator().manual_seed(950 + rank)))
        mono = st3.grad.clone()
        dist.all_reduce(mono)
        sy = GradSync(st3, chunk_elems=4099, overlap=True, sparse_rows_cap=40)
        sy.reduce_bucket(0)
        sy.reduce_bucket(1)
        sy.reduce_bucket(2, torch.zeros(0, dtype=torch.int64))
        sy.finish()
        ok_bucket = ok_bucket and bool(torch.allclose(st3.grad, mono, rtol=1e-6, atol=1e-6))
        ok_mean = ok_mean and ok_bucket
        task = broadcast_task(2 if rank == 0 else 0, "cpu")   # MetaLoader: rank 0's draw wins (data/loader.py:55-59)
        q.put((rank, same_params, ok_sum, ok_mean, gscale, task, float(mine.abs().sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_allreduce_param_broadcast_and_task_broadcast():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, same_params, ok_sum, ok_mean, gscale, task, _ in res:
        assert same_params, f"rank {rank}: parameters differ after broadcast"
        assert ok_sum and ok_mean, f"rank {rank}: all-reduce result wrong"
        assert gscale == 0.5 and task == 2


def _worker3(rank, world, port, q):
    """three ranks, three 'optimizer steps' of two micro-batches each: the flat buffer accumulates both micro-batches' word-embedding rows,
    the last bucket exchanges the UNION of the window's rows (cap x accum_steps) -- against the dense all-reduce of the same buffers"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from magic_amd.host.config import make_config
        from magic_amd.host.model_pretrain import pretrain_specs
        from magic_amd.host.params import ParamStore
        from magic_amd.host.trainer import EMB_TABLE, GradSync, PretrainStep
        cfg = make_config(128, teacher_hidden_size=256, vocab_size=300, num_l_layers=3, num_x_layers=1, num_pano_layers=2)
        store = ParamStore(pretrain_specs(cfg), "cpu", torch.float32, seed=3)
        off, n, (R, H) = store.offsets[EMB_TABLE]
        sy = GradSync(store, chunk_elems=5003, overlap=True, sparse_rows_cap=24)
        win = PretrainStep.__new__(PretrainStep)              # only the window bookkeeping of the trainer (no device, no optimizer)
        win.accum_steps, win._window_rows = 2, []
        ok, worst = True, 0.0
        for step in range(3):
            gen = torch.Generator().manual_seed(1000 * step + rank)
            store.grad.zero_()
            win._window_rows = []
            for micro in range(2):                            # micro-batches touch different (overlapping) rows, <= cap each
                g = torch.randn(store.total, generator=gen)
                rows = torch.randperm(R, generator=gen)[:10 + 3 * rank + micro]
                tab = g[off:off + n].view(R, H)
                keep = tab[rows].clone()
                tab.zero_()
                tab[rows] = keep
                store.grad.add_(g)
                win._window_rows.append(torch.unique(rows))
            mono = store.grad.clone()
            dist.all_reduce(mono)
            touched = win._window_touched()
            assert touched.numel() > 24 or rank == 0          # the union outgrows ONE micro-batch's cap on ranks 1, 2: cap_scale is needed
            sy.reduce_bucket(0)
            sy.reduce_bucket(1)
            sy.reduce_bucket(2, touched, cap_scale=2)
            sy.finish()
            worst = max(worst, float((store.grad - mono).abs().max()))
            ok = ok and bool(torch.allclose(store.grad, mono, rtol=1e-6, atol=1e-6))
            every = [torch.empty_like(store.grad) for _ in range(world)]
            dist.all_gather(every, store.grad)
            ok = ok and all(bool(torch.equal(every[0], e)) for e in every[1:])          # bitwise-equal replicas
            # what the round-3 code did (only the LAST micro-batch's rows): rows touched by the first micro-batch alone stay unsummed
            if step == 0:
                only_last = win._window_rows[-1]
                missed = set(touched.tolist()) - set(only_last.tolist())
                ok = ok and len(missed) > 0
        win._window_rows = [torch.arange(3), None]            # one dense micro-batch (mlm: tied decoder) makes the window dense
        ok = ok and win._window_touched() is None
        q.put((rank, ok, worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_three_ranks_three_buckets_sparse_rows_of_an_accumulation_window():
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker3, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, ok, worst in res:
        assert ok, f"rank {rank}: window-union sparse exchange != dense all-reduce (max diff {worst:.3e}) or replicas differ"
