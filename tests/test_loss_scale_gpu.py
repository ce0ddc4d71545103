"""Dynamic loss scale of the fp16 mode (-m gpu): amp.GradScaler's rule (the reference's fp16 option, pretrain_src/train_r2r_magic.py:370-371)
kept on the device -- csrc/loss.hip step_rng_kernel (the rule), magic_seed_scale (the loss kernels read S), csrc/optim.hip adamw_kernel
(divides by S, skips a non-finite step, reports it).  Kernel level: the rule and the seeds.  Trainer level: an injected overflow is
skipped, halves the scale, leaves weights / moments / the optimizer's step count untouched, and the run then follows the oracle
optimizer (oracle/optim_ref.py) exactly as a run that never saw the bad step; a scale that really overflows recovers within a few steps."""
import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import ops as O
from magic_amd.host import synth
from magic_amd.host.trainer import PretrainStep
from oracle import optim_ref
from tests.test_model_gpu import RW, build

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _state(S, tracker=0.0, pending=0.0):
    return torch.tensor([S, 1.0 / S, tracker, pending], dtype=torch.float32, device=DEV)


def _prologue(st, interval=3):
    c, seed = torch.zeros(1, dtype=torch.int32, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV)
    O.step_rng(7, c, 4.0, seed_out=seed, scale_state=st, growth=2.0, backoff=0.5, interval=interval)
    torch.cuda.synchronize()
    return st.tolist()


def test_scale_rule_is_gradscalers():
    st = _state(1024.0)
    assert _prologue(st) == [1024.0, 1 / 1024.0, 0.0, 0.0]             # no optimizer step since the last prologue (an accumulation micro-step)
    for k in (1.0, 2.0):
        st[3] = 1.0                                                     # a clean update
        assert _prologue(st) == [1024.0, 1 / 1024.0, k, 0.0]
    st[3] = 1.0
    assert _prologue(st) == [2048.0, 1 / 2048.0, 0.0, 0.0]              # `interval` clean updates in a row: doubled
    st[3] = 1.0
    assert _prologue(st)[2] == 1.0
    st[3] = 2.0                                                         # a skipped update: halved, the streak forgotten
    assert _prologue(st) == [1024.0, 1 / 1024.0, 0.0, 0.0]


def test_adamw_divides_by_the_scale_reports_and_takes_the_step_back():
    n, S = 4099, 512.0
    g0 = torch.randn(n, device=DEV, generator=torch.Generator(DEV).manual_seed(1))
    p0 = torch.randn(n, device=DEV, generator=torch.Generator(DEV).manual_seed(2))

    def run(g, st, step):
        p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        ss = torch.zeros(1, device=DEV)
        O.sumsq(g, ss)
        over = torch.zeros(1, dtype=torch.int32, device=DEV)
        O.adamw(n, p, g, m, v, None, 1e-3, 0.9, 0.98, 1e-6, 0.01, 1e-3, ss, 5.0, 1.0, zero_grad=False, overflow=over, scale_state=st, sched_step=step)
        torch.cuda.synchronize()
        return p, int(over.item())
    ref, _ = run(g0.clone(), None, None)
    st, step = _state(S), torch.full((2,), 7, dtype=torch.int32, device=DEV)      # {global_step, optimizer state step} (csrc/optim.hip)
    got, over = run(g0 * S, st, step)
    assert torch.allclose(got, ref, rtol=1e-6, atol=1e-7) and over == 0        # S x gradient, divided by S in the kernel (clip norm included)
    assert st.tolist()[3] == 1.0 and step.tolist() == [7, 7]
    bad = g0 * S
    bad[5] = float("inf")
    got, over = run(bad, st, step)
    assert torch.equal(got, p0) and over == 1 and st.tolist()[3] == 2.0 and step.tolist() == [7, 6]      # skipped: the optimizer's state step taken back, global_step left alone


def test_loss_kernels_multiply_their_seeds_by_the_registered_word():
    M, N, S = 16, 40, 256.0
    gen = torch.Generator(DEV).manual_seed(3)
    logits = torch.randn(M, N, device=DEV, generator=gen).to(torch.float16)
    labels = torch.randint(0, N, (M,), device=DEV, generator=gen, dtype=torch.int32)
    t_log = torch.randn(M, N, device=DEV, generator=gen)
    s_log = torch.randn(M, N, device=DEV, generator=gen)
    targets = torch.softmax(torch.randn(M, N, device=DEV, generator=gen), 1).contiguous()

    def all_seeds():
        out = {}
        lr, d = torch.zeros(M, device=DEV), torch.zeros(M, N, device=DEV, dtype=torch.float16)
        O.ce_rows(logits, M, N, N, labels, coef=0.5, loss_row=lr, dlogits=d, ldd=N)
        out["ce"] = (lr, d.float())
        lr2, d2 = torch.zeros(M, device=DEV), torch.zeros(M, N, device=DEV, dtype=torch.float16)
        O.softkl_rows(logits, M, N, N, targets, coef=0.25, loss_row=lr2, dlogits=d2, ldd=N)
        out["softkl"] = (lr2, d2.float())
        lr3, d3 = torch.zeros(M, device=DEV), torch.zeros(M, N, device=DEV)
        O.kd_rows(s_log, t_log, M, N, N, 2.0, norm=1.0 / (M * N), coef=0.7, loss_row=lr3, ds=d3)
        out["kd"] = (lr3, d3)
        torch.cuda.synchronize()
        return out
    plain = all_seeds()
    st = _state(S)
    O.seed_scale(st)
    try:
        scaled = all_seeds()
    finally:
        O.seed_scale(None)
    again = all_seeds()
    for k in plain:
        assert torch.equal(plain[k][0], scaled[k][0]), k                       # loss VALUES are never scaled
        assert torch.allclose(scaled[k][1], plain[k][1] * S, rtol=2e-3, atol=1e-3 * float(plain[k][1].abs().max()) * S), k
        assert torch.equal(plain[k][1], again[k][1]), k                        # cleared: back to unscaled


def _oracle_run(o_t, o_s, steps, skip=()):
    from magic_amd.host.params import is_no_decay
    names = [n for n, _ in o_s.named_parameters()]
    wds = [0.0 if is_no_decay(n) else 0.01 for n in names]
    state = optim_ref.adamw_init([p.data for p in o_s.parameters()])
    applied = 0
    for step, task in enumerate(steps):
        batch = synth.make_batch(task, batch_size=4, seed=77, step=step, vocab=600, min_len=8, max_len=15, min_steps=2, max_steps=3)
        with torch.no_grad():
            ot = o_t(batch, task, compute_loss=True)["outputs"]
        for p in o_s.parameters():
            p.grad = None
        w = o_s(batch, task, compute_loss=True, teacher_outputs=ot, rw=torch.tensor(RW))
        w["loss"].backward()
        if step in skip:
            continue                                  # GradScaler: optimizer.step() skipped -- no update, the state step not advanced
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in o_s.parameters()]
        optim_ref.clip_grad_norm(grads, 5.0)
        lr = optim_ref.get_lr_sched(step, 1e-3, 2, 10)         # lr follows global_step (skipped steps count), the bias correction the applied updates
        with torch.no_grad():
            optim_ref.adamw_step([p.data for p in o_s.parameters()], grads, state, lr=lr, betas=(0.9, 0.98), eps=1e-6, weight_decay=wds)
        applied += 1


def test_injected_overflow_is_skipped_then_the_run_follows_the_oracle_optimizer():
    steps = ["sap", "mlm", "cfp", "sap", "mlm"]
    bad_step = 1
    o_t, o_s, g_t, g_s = build(torch.float16)
    p0 = {n: p.data.clone() for n, p in o_s.named_parameters()}
    trainer = PretrainStep(g_s, g_t, lr=1e-3, warmup_steps=2, num_train_steps=10, grad_norm=5.0, loss_scale_interval=1000)
    st = trainer.opt.loss_scale
    assert st is not None and st.tolist()[0] == 4096.0 and g_s.loss_scale is st
    inner = trainer._optimize
    at = {"step": 0}

    def poisoned():
        if at["step"] == bad_step:
            g_s.store.grad[1234] = float("inf")          # what an activation gradient past 65504 leaves behind
        inner()
    trainer._optimize = poisoned
    before_bad = None
    for step, task in enumerate(steps):
        at["step"] = step
        if step == bad_step:
            before_bad = (g_s.store.flat.clone(), g_s.store.m.clone(), g_s.store.v.clone(), trainer.opt.step_dev.tolist())
        batch = synth.make_batch(task, batch_size=4, seed=77, step=step, vocab=600, min_len=8, max_len=15, min_steps=2, max_steps=3)
        trainer.step(batch, task, rw=RW)
        if step == bad_step:
            torch.cuda.synchronize()
            assert torch.equal(g_s.store.flat, before_bad[0]) and torch.equal(g_s.store.m, before_bad[1]) and torch.equal(g_s.store.v, before_bad[2])
            # the skipped step: global_step (the lr schedule's word) advanced, the optimizer's state step (bias correction) taken back
            assert trainer.opt.step_dev.tolist() == [before_bad[3][0] + 1, before_bad[3][1]] and st.tolist()[3] == 2.0
            assert float(g_s.store.grad.abs().max()) == 0.0                   # consumed: the next step starts from zero
        if step == bad_step + 1:
            torch.cuda.synchronize()
            assert st.tolist()[0] == 2048.0                                   # halved by the prologue of the step after the skipped one
    torch.cuda.synchronize()
    assert trainer.opt.skipped_steps() == 1 and trainer.opt.step_dev.tolist() == [len(steps), len(steps) - 1]
    assert torch.isfinite(g_s.store.flat).all()
    _oracle_run(o_t, o_s, steps, skip=(bad_step,))
    got = g_s.state_dict()
    num = den = 0.0
    for n, p in o_s.named_parameters():
        du_g, du_o = (got[n].float().cpu() - p0[n]).double(), (p.data - p0[n]).double()
        num += (du_g * du_o).sum().item()
        den += (du_o * du_o).sum().item()
        assert (got[n].float().cpu() - p.data).abs().max().item() < 5e-3, n
    assert 0.97 < num / den < 1.03, num / den


def test_a_scale_that_overflows_recovers_within_a_few_steps():
    o_t, o_s, g_t, g_s = build(torch.float16)
    trainer = PretrainStep(g_s, g_t, lr=1e-3, warmup_steps=2, num_train_steps=40, grad_norm=5.0, loss_scale_init=2.0 ** 24)
    st = trainer.opt.loss_scale
    w0 = g_s.store.flat.clone()
    clean_at = None
    for step in range(16):
        task = ["sap", "mlm", "cfp"][step % 3]
        batch = synth.make_batch(task, batch_size=4, seed=77, step=step, vocab=600, min_len=8, max_len=15, min_steps=2, max_steps=3)
        trainer.step(batch, task, rw=RW)
        torch.cuda.synchronize()
        if st.tolist()[3] == 1.0 and clean_at is None:
            clean_at = step
    assert trainer.opt.skipped_steps() >= 1, "2^24 x the gradient was expected to overflow fp16 activations gradients"
    assert clean_at is not None and clean_at <= 14, (clean_at, st.tolist())
    assert st.tolist()[0] == 2.0 ** 24 / 2 ** trainer.opt.skipped_steps()
    assert torch.isfinite(g_s.store.flat).all() and not torch.equal(g_s.store.flat, w0)
    assert trainer.opt.step_dev.tolist() == [16, 16 - trainer.opt.skipped_steps()]
