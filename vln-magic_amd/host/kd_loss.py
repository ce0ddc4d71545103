"""Drop-in replacements for the reference's MAKD primitives, same names / argument meaning / error behaviour:

    pretrain flavour  pretrain_src/optim/kd_loss.py:5-54      -> mse_loss_pretrain, kd_loss_pretrain
    nav flavour       map_nav_src/utils/kd_loss.py:6-67       -> mse_loss, kd_loss (loss_type='sum'|'mean')
    both              exponential_decay, invert_normalized_losses

Each loss is a torch.autograd.Function over ONE fused HIP kernel that produces the loss value and the gradient
wrt the student input in the same pass (csrc/loss.hip), so `GMapNavAgent.compute_kd_losses`
(map_nav_src/r2r/agent.py:546-719) runs unchanged on top of them.  CUDA tensors only: there is no CPU fallback.
"""
import torch

from . import lib as L
from . import ops as O


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.MagicHipError("magic_amd.kd_loss runs on the GPU only (no CPU fallback); got a CPU tensor")


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, t, w, norm):
        _need_cuda(s, t, w)
        dt = s.dtype if s.dtype in (torch.float32, torch.bfloat16, torch.float16) else torch.float32
        sc, tc = s.detach().to(dt).contiguous(), t.detach().to(dt).contiguous()
        outer = sc.shape[0]
        inner = sc.numel() // outer
        loss = torch.zeros(1, dtype=torch.float32, device=s.device)
        ds = torch.empty_like(sc) if s.requires_grad else None
        O.mse(sc, tc, outer, inner, inner, inner, w=None if w is None else w.detach().float().contiguous(), rows_per_w=1,
              norm=norm, coef=1.0, loss=loss, ds=ds, g_stride=inner)
        ctx.ds, ctx.in_dtype = ds, s.dtype
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        if ctx.ds is None:
            return None, None, None, None
        return (ctx.ds.float() * g).to(ctx.in_dtype), None, None, None


class _MseMulti(torch.autograd.Function):
    """`vals = _MseMulti.apply(meta, x0, y0, x1, y1, ...)`: up to ten mse_loss terms of one distillation step in ONE launch (csrc/loss.hip
    mse_multi_kernel) -- vals[i] = k_i * norm_i * sum w (x_i - y_i)^2 with k_i = coef_i * coef_dev_i[0], and the gradient wrt x_i from the same
    pass.  meta[i] = dict(norm, coef, coef_dev (0-d / 1-element device tensor or None), w (sample weights [outer] or None)).  The navigator's MAKD
    step (host/makd_nav.compute_kd_losses_fused) calls this once per step and direction instead of nine `_MSE` nodes with their casts,
    multiplications by the ability weights and running sums."""

    @staticmethod
    def forward(ctx, meta, *xy):
        n = len(meta)
        dev = xy[0].device
        slots = torch.zeros(n, dtype=torch.float32, device=dev)
        probs, dss, kv = [], [], []
        for i, m in enumerate(meta):
            s, t = xy[2 * i], xy[2 * i + 1]
            _need_cuda(s, t, m.get("w"))
            dt = s.dtype if s.dtype in (torch.float32, torch.bfloat16, torch.float16) else torch.float32
            sc, tc = s.detach().to(dt).contiguous(), t.detach().to(dt).contiguous()
            outer = sc.shape[0]
            inner = sc.numel() // outer
            ds = torch.empty_like(sc) if s.requires_grad else None
            w = m.get("w")
            cd = m.get("coef_dev")
            probs.append(dict(s=sc, t=tc, outer=outer, inner=inner, s_stride=inner, t_stride=inner, w=None if w is None else w.detach().float().contiguous(),
                              rows_per_w=1, norm=m["norm"], coef=m["coef"], coef_dev=cd, loss=slots[i:i + 1], ds=ds, g_stride=inner))
            dss.append(ds)
            kv.append((float(m["coef"]), cd))
        O.mse_multi(probs)
        # the VALUES carry the same factors as the gradients: host coefficient x device-side ability weight
        host = torch.tensor([k[0] for k in kv], dtype=torch.float32).to(dev, non_blocking=True) if any(k[0] != 1.0 for k in kv) else None
        vals = slots if host is None else slots * host
        if any(k[1] is not None for k in kv):
            vals = vals * torch.stack([(k[1].reshape(()) if k[1] is not None else torch.ones((), device=dev)) for k in kv])
        ctx.dss, ctx.dtypes = dss, [xy[2 * i].dtype for i in range(n)]
        ctx.set_materialize_grads(False)
        return vals

    @staticmethod
    def backward(ctx, g):
        n = len(ctx.dss)
        if g is None:
            return (None,) * (1 + 2 * n)
        live = [i for i, ds in enumerate(ctx.dss) if ds is not None]
        gs = g.unbind(0)                                   # one op: n views
        # ds_i * g_i for every term in one multi-tensor launch per dtype group instead of n select / mul / cast triples
        scaled = torch._foreach_mul([ctx.dss[i] for i in live], [gs[i] for i in live]) if live else []
        out = [None] * (1 + 2 * n)
        for i, t in zip(live, scaled):
            out[1 + 2 * i] = t if t.dtype == ctx.dtypes[i] else t.to(ctx.dtypes[i])
        return tuple(out)


class _CERows(torch.autograd.Function):
    """`F.cross_entropy(logits, targets, ignore_index=..., reduction='none')` of the navigator's step loop (agent_base.py:152 criterion,
    agent.py:1007-1021) as ONE launch: the row losses and the unit gradient softmax(logits) - onehot (zero on ignored rows) come out of the
    same pass (csrc/loss.hip ce_rows_kernel); backward scales the saved gradient by the incoming row gradients."""

    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        _need_cuda(logits, targets)
        x = logits.detach().float().contiguous()          # action logits are fp32 (-inf on masked candidates)
        M, N = x.shape
        lab = targets if targets.dtype == torch.int32 else targets.to(torch.int32)
        rows = torch.empty(M, dtype=torch.float32, device=x.device)
        dl = torch.empty_like(x) if logits.requires_grad else None
        O.ce_rows(x, M, N, N, lab, ignore_index=int(ignore_index), coef=1.0, loss_row=rows, dlogits=dl, ldd=N)
        ctx.dl, ctx.in_dtype = dl, logits.dtype
        return rows

    @staticmethod
    def backward(ctx, g):
        if ctx.dl is None:
            return None, None, None
        return (ctx.dl * g[:, None]).to(ctx.in_dtype), None, None


def ce_rows_loss(logits, targets, ignore_index=-100):
    """per-row cross entropy [M] of fp32 logits [M, N] against integer targets (ignored rows: 0 loss, 0 gradient)"""
    return _CERows.apply(logits, targets, ignore_index)


class _KD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, t, w, temperature, norm):
        _need_cuda(s, t, w)
        sc, tc = s.detach().float().contiguous(), t.detach().float().contiguous()
        M, N = sc.shape
        rows = torch.empty(M, dtype=torch.float32, device=s.device)
        ds = torch.empty_like(sc) if s.requires_grad else None
        O.kd_rows(sc, tc, M, N, N, temperature, w=None if w is None else w.detach().float().contiguous(), norm=norm, coef=1.0,
                  loss_row=rows, ds=ds)
        ctx.ds, ctx.in_dtype = ds, s.dtype
        return rows.sum()

    @staticmethod
    def backward(ctx, g):
        if ctx.ds is None:
            return None, None, None, None, None
        return (ctx.ds * g).to(ctx.in_dtype), None, None, None, None


# ---- nav flavour (map_nav_src/utils/kd_loss.py) --------------------------------------------------------
def mse_loss(s_inputs, t_inputs, t_sample_weights=None, loss_type="sum", **kwargs):
    if loss_type not in ("sum", "mean"):
        raise ValueError("Unsupported loss_type. Choose 'sum' or 'mean'.")
    if t_sample_weights is not None and s_inputs.shape[0] != t_sample_weights.shape[0]:
        raise ValueError("Shape mismatch between sample weights and inputs")
    norm = 1.0 if loss_type == "sum" else 1.0 / s_inputs.numel()
    return _MSE.apply(s_inputs, t_inputs, t_sample_weights, norm)


def kd_loss(student_logits, teacher_logits, temperature=1, epsilon=1e-6, t_sample_weights=None, loss_type="sum", **kwargs):
    M, N = student_logits.shape
    if loss_type == "sum":
        norm = 1.0
    elif loss_type == "mean":
        norm = 1.0 / M if t_sample_weights is not None else 1.0 / (M * N)
    else:
        raise ValueError("Unsupported loss_type. Choose 'sum' or 'mean'.")
    return _KD.apply(student_logits, teacher_logits, t_sample_weights, float(temperature), norm)


# ---- pretrain flavour (pretrain_src/optim/kd_loss.py) ---------------------------------------------------
def mse_loss_pretrain(s_inputs, t_inputs, t_sample_weights=None, **kwargs):
    w = t_sample_weights
    if w is not None and s_inputs.shape[0] != w.shape[0]:
        w = None                       # silent fallback to the unweighted mean (kd_loss.py:15-16)
    return _MSE.apply(s_inputs, t_inputs, w, 1.0 / s_inputs.numel())


def kd_loss_pretrain(student_logits, teacher_logits, temperature=1, epsilon=1e-6, t_sample_weights=None, **kwargs):
    return kd_loss(student_logits, teacher_logits, temperature, epsilon, t_sample_weights, loss_type="mean")


def exponential_decay(t_sample_losses, decay_rate=0.1):
    return torch.exp(-decay_rate * t_sample_losses)


def invert_normalized_losses(t_sample_losses, **kwargs):
    lo, hi = torch.min(t_sample_losses), torch.max(t_sample_losses)
    return 1 - (t_sample_losses - lo) / (hi - lo)
