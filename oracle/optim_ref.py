"""ORACLE (test infrastructure): the reference's optimizer arithmetic restated on plain tensors.
Follows /root/reference/pretrain_src/optim/adamw.py:53-112 (HF-style AdamW: eps added to sqrt(v),
bias-corrected step size, decoupled weight decay applied AFTER the Adam update with the raw lr,
:109-110) and pretrain_src/optim/sched.py:17-30.  Pinned by tests/golden/adamw.pt."""
import math

import torch


def adamw_init(params):
    return dict(step=0, m=[torch.zeros_like(p) for p in params], v=[torch.zeros_like(p) for p in params])


def adamw_step(params, grads, state, lr, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, correct_bias=True):
    state["step"] += 1
    b1, b2 = betas
    wds = weight_decay if isinstance(weight_decay, (list, tuple)) else [weight_decay] * len(params)
    for p, g, m, v, wd in zip(params, grads, state["m"], state["v"], wds):
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        step_size = lr
        if correct_bias:
            step_size = lr * math.sqrt(1 - b2 ** state["step"]) / (1 - b1 ** state["step"])
        p.addcdiv_(m, v.sqrt().add_(eps), value=-step_size)
        if wd > 0:
            p.add_(p, alpha=-lr * wd)


def warmup_linear(step, warmup, total):
    if step < warmup:
        return step / warmup
    return max(0, (total - step) / (total - warmup))


def get_lr_sched(step, lr, warmup, total):
    v = lr * warmup_linear(step, warmup, total)
    return v if v > 0 else 1e-8


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ semantics (grad_norm 5.0: r2r_magic_pretrain.json:22)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total
