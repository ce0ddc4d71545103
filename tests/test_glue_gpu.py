"""The two "glue" kernels of round 3 (csrc/loss.hip): the step prologue (MKRW weights + dropout seed from one launch) and the loss assembly
(supervised mean, ability-weighted MAKD terms, total) -- against plain torch arithmetic (-m gpu)."""
import math

import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_step_rng_statistics_and_counter():
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    seed, rw = torch.zeros(2, dtype=torch.int32, device=DEV), torch.zeros(5, device=DEV)
    T, n = 4.0, 4000
    rws, seeds = [], []
    for _ in range(n):
        O.step_rng(1234, counter, T, seed_out=seed, rw_out=rw)
        rws.append(rw.clone()); seeds.append(seed.clone())
    torch.cuda.synchronize()
    assert int(counter) == n
    rws, seeds = torch.stack(rws).cpu().double(), torch.stack(seeds).cpu()
    assert torch.allclose(rws.sum(1), torch.full((n,), 5.0, dtype=torch.float64), atol=1e-4)          # softmax(...) * 5
    assert (rws > 0).all()
    # rw_i = 5 softmax(z / T)_i with z ~ N(0, 1): log(rw_i / rw_j) = (z_i - z_j) / T ~ N(0, 2 / T^2)
    lr = torch.log(rws[:, 0] / rws[:, 1])
    assert abs(lr.mean().item()) < 4 * math.sqrt(2.0 / T ** 2 / n)
    assert abs(lr.std().item() - math.sqrt(2.0) / T) < 0.03
    # the same five draws are independent of each other and across steps (lag-1 correlation ~ 0)
    z = torch.log(rws) - torch.log(rws).mean(1, keepdim=True)
    assert abs(torch.corrcoef(torch.stack([z[:-1, 0], z[1:, 0]]))[0, 1].item()) < 0.06
    assert (seeds >= 0).all() and len({tuple(s.tolist()) for s in seeds}) == n                           # 31-bit words, all distinct
    # determinism: same base seed + same counter -> same values; another base seed -> other values
    c2 = torch.zeros(1, dtype=torch.int32, device=DEV)
    O.step_rng(1234, c2, T, seed_out=seed, rw_out=rw)
    torch.cuda.synchronize()
    assert torch.equal(seed.cpu(), seeds[0]) and torch.allclose(rw.cpu().double(), rws[0], atol=1e-6)
    c3 = torch.zeros(1, dtype=torch.int32, device=DEV)
    O.step_rng(99, c3, T, seed_out=seed, rw_out=rw)
    torch.cuda.synchronize()
    assert not torch.equal(seed.cpu(), seeds[0])


@pytest.mark.parametrize("has_kd,with_w,with_rows", [(True, False, True), (True, True, False), (False, False, False)])
def test_loss_assemble_matches_torch(has_kd, with_w, with_rows):
    g = torch.Generator().manual_seed(3)
    rows = torch.rand(613, generator=g).to(DEV)
    w = torch.rand(613, generator=g).to(DEV) if with_w else None
    kd_rows = torch.rand(48, generator=g).to(DEV) if with_rows else None
    slots = torch.rand(16, generator=g).to(DEV)
    rw = (torch.rand(5, generator=g) + 0.5).to(DEV)
    out = torch.full((16,), float("nan"), device=DEV)
    s0 = slots.clone()
    O.loss_assemble(rows, w, 0.37, kd_rows, slots, rw, 0.5, has_kd, out)
    torch.cuda.synchronize()
    sup = 0.37 * ((rows * w).sum() if with_w else rows.sum())
    assert torch.allclose(out[0], sup, rtol=1e-5)
    if has_kd:
        want = s0.clone()
        if with_rows:
            want[9] = kd_rows.sum()
            assert torch.allclose(slots[9], kd_rows.sum(), rtol=1e-5)
        terms = want[:10] * rw[torch.tensor([0, 0, 1, 1, 1, 2, 2, 3, 3, 4], device=DEV)]
        assert torch.allclose(out[1:11], terms, rtol=1e-5)
        assert torch.allclose(out[11], terms.sum(), rtol=1e-5)
        assert torch.allclose(out[12], 0.5 * terms.sum() + 0.5 * sup, rtol=1e-5)
    else:
        assert float(out[11]) == 0.0 and torch.allclose(out[12], sup, rtol=1e-5)


def test_fused_schedule_and_gradient_zeroing_follow_the_separate_launches():
    """sumsq + schedule in one launch and AdamW zeroing its gradients == sched_step; sumsq; adamw; zero_grad, over three steps"""
    from magic_amd.host.config import make_config
    from magic_amd.host.model_pretrain import pretrain_specs
    from magic_amd.host.params import ParamStore
    from magic_amd.host.trainer import FusedAdamW
    cfg = make_config(128, teacher_hidden_size=256, vocab_size=300, num_l_layers=1, num_x_layers=1, num_pano_layers=1)
    stores = [ParamStore(pretrain_specs(cfg), DEV, torch.bfloat16, seed=1) for _ in range(2)]
    opts = [FusedAdamW(s, lr=1e-3, schedule=(2, 10), max_grad_norm=0.5) for s in stores]
    g = torch.Generator().manual_seed(0)
    for step in range(3):
        grad = (torch.randn(stores[0].total, generator=g) * 0.01).to(DEV)
        for s in stores:
            s.grad.copy_(grad)
        opts[0].step(gscale=0.5)                                   # separate launches
        opts[1].ss.zero_()                                         # (what the step prologue does)
        opts[1].step(gscale=0.5, ss_zeroed=True, zero_grad=True)   # fused forms
        torch.cuda.synchronize()
        # (the gradient norm is an atomic sum over blocks: its last bits, and with them the clip factor, depend on arrival order)
        assert torch.allclose(stores[0].flat, stores[1].flat, rtol=1e-5, atol=1e-8)
        assert torch.allclose(stores[0].shadow.float(), stores[1].shadow.float(), rtol=1e-2, atol=1e-6)
        assert torch.equal(opts[0].lr_ss, opts[1].lr_ss) and opts[0].step_dev.tolist() == opts[1].step_dev.tolist() == [step + 1, step + 1]
        assert float(stores[1].grad.abs().max()) == 0.0 and float(stores[0].grad.abs().max()) > 0.0


def test_dw_guard_orders_streams_that_share_the_weight_gradient_workspace():
    """ops.dw_guard: the deterministic weight-gradient launch's workspace and counters are one pair per device and gradient lane, baked
    into every captured graph; a user on ANOTHER stream must wait for everything the previous user's stream has queued.  Same stream, or another
    lane (own pair): no wait."""
    import torch
    from magic_amd.host import lanes
    a = lanes.beside([torch.cuda.current_stream()])        # two streams that really run side by side (two pool streams may share a hardware queue:
    b = lanes.beside([torch.cuda.current_stream(), a])     # the "did not wait" half of this test would then see an ordering nobody asked for)
    x = torch.randn(4096, 4096, device=DEV)
    torch.cuda.synchronize()
    O._DW_LAST.clear()
    with torch.cuda.stream(a):
        O.dw_guard()
        for _ in range(40):                       # tens of milliseconds of work on stream a
            x = x @ x * 1e-4
        ev_a = torch.cuda.Event()
        ev_a.record()
    with torch.cuda.stream(b):
        O.dw_guard()                              # another stream, same lane: waits for stream a
        ev_b = torch.cuda.Event()
        ev_b.record()
    ev_b.synchronize()
    assert ev_a.query(), "stream b passed the guard while stream a's launches were still running"
    torch.cuda.synchronize()
    y = torch.randn(4096, 4096, device=DEV)
    with torch.cuda.stream(a):
        O.dw_guard()
        for _ in range(40):
            y = y @ y * 1e-4
        ev_a2 = torch.cuda.Event()
        ev_a2.record()
    with lanes.use(1, b):
        O.dw_guard()                              # lane 1 has its own workspace: no ordering against lane 0's stream
        ev_b2 = torch.cuda.Event()
        ev_b2.record()
    ev_b2.synchronize()
    assert not ev_a2.query(), "a lane with its own workspace was made to wait for another lane's stream"
    torch.cuda.synchronize()
    O._DW_LAST.clear()
    lanes._used.clear()


def test_flat_torch_adamw_equals_torch_optim_adamw_with_clipping():
    """trainer.FlatTorchAdamW (the navigator loop's optimizer on the flat buffers: clip + decay-first AdamW in two launches) against
    torch.optim.AdamW + clip_grad_norm_ over three steps (map_nav_src/r2r/agent_base.py:122-137, :273)."""
    import torch
    from magic_amd.host.params import ParamStore
    from magic_amd.host.trainer import FlatTorchAdamW
    specs = [("a.weight", (37, 64), "normal"), ("a.bias", (37,), "zeros"), ("n.LayerNorm.weight", (64,), "ones"), ("b.weight", (5, 3), "normal")]
    st = ParamStore(specs, DEV, torch.float32, seed=3)
    ref = {n: torch.nn.Parameter(st.master(n).clone()) for n, _, _ in specs}
    topt = torch.optim.AdamW(list(ref.values()), lr=1e-2)
    fopt = FlatTorchAdamW(st, lr=1e-2)
    gen = torch.Generator(DEV).manual_seed(1)
    for step in range(3):
        fopt.zero_grad()
        topt.zero_grad()
        for n, shape, _ in specs:
            g = torch.randn(shape, device=DEV, generator=gen) * (30.0 if step == 1 else 0.5)      # step 1: the norm exceeds 40 -> clipped
            st.g(n).copy_(g)
            ref[n].grad = g.clone()
        torch.nn.utils.clip_grad_norm_(list(ref.values()), 40.0)
        topt.step()
        fopt.step(max_norm=40.0)
        torch.cuda.synchronize()
        for n, _, _ in specs:
            torch.testing.assert_close(st.master(n), ref[n].data, rtol=2e-6, atol=1e-6, msg=f"step {step} {n}")      # (updates are ~1e-2 per step: 1e-6 is 1e-4 of one)
        assert float(st.grad.abs().max()) == 0.0               # consumed: the next backward starts from zero


def test_lockstep_segments_alternate_and_pair_inside_a_capture():
    """lib.lockstep (the two co-attention encoders of a captured pre-training step): the two host threads hand a baton back and forth --
    never both inside the runtime at once, which is what lost a node from a capturing stream's dependency chain (hipErrorStreamCaptureUnjoined
    at capture end, intermittently) -- their groupable launches still pair, and the replayed graph computes what the separate launches do"""
    import threading
    import time

    from magic_amd.host import lib as L
    M, N, K, n_calls = 64, 64, 64, 12
    gen = torch.Generator(DEV).manual_seed(5)
    A = [torch.randn(M, K, device=DEV, generator=gen).to(torch.bfloat16) for _ in range(2)]
    W = [[torch.randn(N, K, device=DEV, generator=gen).to(torch.bfloat16) for _ in range(n_calls)] for _ in range(2)]
    C = [[torch.zeros(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(n_calls)] for _ in range(2)]
    inside, overlaps, order = [0], [0], []
    guard = threading.Lock()

    def segment(i):
        def run():
            for k in range(n_calls):
                with guard:
                    inside[0] += 1
                    overlaps[0] += inside[0] > 1
                    order.append(i)
                time.sleep(0.001)              # a window in which a free-running partner would show up
                with guard:
                    inside[0] -= 1
                O.gemm(0, A[i], W[i][k], C[i][k], M, N, K, K, K, N)      # groupable: offered to the partner
            return i
        return run
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        out = L.lockstep(segment(0), segment(1))
    assert out == (0, 1)
    assert overlaps[0] == 0, overlaps
    assert order.count(0) == order.count(1) == n_calls and set(order[:3]) == {0, 1}, order[:8]      # interleaved call by call, not one after the other
    g.replay()
    torch.cuda.synchronize()
    for i in range(2):
        for k in range(n_calls):
            ref = A[i].float() @ W[i][k].float().t()
            assert torch.allclose(C[i][k].float(), ref, rtol=2e-2, atol=2e-1), (i, k)


def test_side_streams_are_picked_by_measuring_that_they_run_beside_the_main_stream():
    """lanes.beside / csrc/encoder.hip magic_stream_probe: the runtime deals streams onto a few hardware queues and two streams on one queue run
    in order; the teacher's stream, the rollout lanes and the gradient-exchange stream are therefore chosen by a measurement, not by luck"""
    from magic_amd.host import lanes
    from magic_amd.host import lib as L
    main = torch.cuda.current_stream()
    n0 = len(lanes.probe_log)
    s = lanes.beside([main])
    assert lanes.probe_log[n0]["found"] and s.cuda_stream != main.cuda_stream
    assert lanes.runs_beside(s, main) and lanes.runs_beside(main, s)
    t = lanes.beside([main, s])                                   # (the exchange stream: beside the student's AND the teacher's)
    assert len({main.cuda_stream, s.cuda_stream, t.cuda_stream}) == 3 and lanes.runs_beside(t, s) and lanes.runs_beside(t, main)
    # the probe does tell streams that share a queue apart: among torch's 32 pool streams some pair must (there are fewer queues than that)
    pool, seen = [], set()
    for _ in range(40):
        c = torch.cuda.Stream()
        if c.cuda_stream not in seen:
            seen.add(c.cuda_stream)
            pool.append(c)
    in_order = sum(not lanes.runs_beside(pool[0], c) for c in pool[1:])
    print(f"{len(pool)} pool streams: {in_order} of {len(pool) - 1} run in order with the first one")
    assert 0 < in_order < len(pool) - 1
    w = torch.zeros(2, dtype=torch.int32, device=DEV)
    with pytest.raises(L.MagicHipError):
        L.call("magic_stream_probe", L.P(w), 100, s.cuda_stream, s.cuda_stream)
