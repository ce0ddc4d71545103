"""Data-parallel wrapping for UNMODIFIED reference drivers.

The reference wraps its models in `torch.nn.parallel.DistributedDataParallel` (pretrain_src/utils/misc.py:57-71
`wrap_model(model, device, local_rank)`; map_nav_src/r2r/agent_base.py:114-116).  DDP averages gradients from autograd
hooks on the parameters' accumulation nodes.  The models here run an explicit HIP backward that writes the flat gradient
buffer directly, so those hooks never fire: under torch's DDP the exchange would be skipped on every step, silently.

Two things close that hazard:
  * `refuse_torch_ddp(module)` -- called at the top of every model forward: if the call arrives through a torch
    DistributedDataParallel / DataParallel whose `.module` is this model it raises, naming the replacement;
  * `wrap_model` / `DistributedDataParallel` below -- same call shapes as the reference's, same DDP semantics
    (rank-0 parameters broadcast at construction, gradients averaged over ranks by the end of `loss.backward()`,
    `.module` attribute, `module.`-prefixed state_dict keys which the reference's savers strip: utils/save.py:33,
    agent_base.py:298-315), implemented on the flat buffers (`trainer.auto_sync`: one bucketed all-reduce per step).
"""
import sys

import torch
import torch.distributed as dist
import torch.nn as nn

_TORCH_WRAPPERS = (torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel)


class TorchDDPWrapperError(RuntimeError):
    pass


def find_torch_wrapper(module, max_depth=16):
    """the torch DDP / DataParallel instance whose forward is calling `module` right now, or None (walks the Python stack:
    DDP.forward -> _run_ddp_forward -> module.__call__ -> module.forward)"""
    f = sys._getframe(1)
    for _ in range(max_depth):
        if f is None:
            return None
        # only frames of torch's own wrapper code can hold such a `self`: `f_locals` (a dict built per access) is not touched for the
        # engine's own frames, which is every frame of an ordinary step
        if "parallel" in f.f_code.co_filename:
            s = f.f_locals.get("self")
            if isinstance(s, _TORCH_WRAPPERS) and getattr(s, "module", None) is module:
                return s
        f = f.f_back
    return None


def refuse_torch_ddp(module):
    w = find_torch_wrapper(module)
    if w is not None:
        raise TorchDDPWrapperError(
            f"{type(module).__name__} is wrapped in torch.nn.parallel.{type(w).__name__}: its reducer hooks on autograd, and this model's "
            "backward is explicit HIP code that bypasses autograd, so gradients would never be averaged across ranks. Replace the wrap "
            "call by `magic_amd.wrap_model(model, device, local_rank)` (same signature as pretrain_src/utils/misc.py:57) or the class by "
            "`magic_amd.DistributedDataParallel` (same constructor arguments); INTEGRATION.md section 1.")


def _world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _stores(module):
    """every ParamStore under `module` (the pretraining model and VLNBert own one each)"""
    seen, out = set(), []
    for m in module.modules():
        s = getattr(m, "store", None)
        if s is not None and id(s) not in seen:
            seen.add(id(s))
            out.append((m, s))
    return out


class DistributedDataParallel(nn.Module):
    """DDP-shaped wrapper: `.module`, call-through forward, rank-0 parameter broadcast at construction, gradient averaging
    at the end of the explicit backward.  Accepts (and ignores) torch DDP's tuning arguments: every rank runs the same task
    per step (data/loader.py:55-59), so task-unused parameters contribute zeros and no `find_unused_parameters` walk is
    needed; bucketing is `trainer.GradSync`'s."""

    def __init__(self, module, device_ids=None, output_device=None, find_unused_parameters=False, broadcast_buffers=True, **_ignored):
        super().__init__()
        if isinstance(module, _TORCH_WRAPPERS):
            raise TorchDDPWrapperError("wrap the bare model, not a torch DDP/DataParallel wrapper")
        owners = _stores(module)
        if not owners:
            raise TypeError(f"{type(module).__name__} holds no magic_amd ParamStore; use torch's DistributedDataParallel for ordinary modules")
        self.module = module
        self.device_ids, self.find_unused_parameters = device_ids, find_unused_parameters
        if _world() > 1:
            for m, s in owners:
                dist.broadcast(s.flat, src=0)             # DDP ctor: rank 0's state reaches every rank (utils/misc.py:62-66 comment)
                s.shadow_clean = False
        for m, s in owners:
            if s.requires_grad:
                m.auto_grad_sync = True

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def no_sync(self):
        """gradient accumulation: `with ddp.no_sync(): loss.backward()` skips the exchange, as torch's DDP does"""
        owners = [m for m, s in _stores(self.module) if s.requires_grad]

        class _NoSync:
            def __enter__(self_):
                self_.prev = [getattr(m, "auto_grad_sync", True) for m in owners]
                for m in owners:
                    m.auto_grad_sync = False

            def __exit__(self_, *exc):
                for m, p in zip(owners, self_.prev):
                    m.auto_grad_sync = p
                return False
        return _NoSync()


def wrap_model(model, device, local_rank, find_unused_parameters=True):
    """Drop-in for pretrain_src/utils/misc.py:57-71: `.to(device)`, then DDP when `local_rank != -1`.  The reference's
    DataParallel branch (single process, several GPUs) is refused: this engine is one process per GPU."""
    model.to(device)
    if local_rank != -1:
        return DistributedDataParallel(model, device_ids=[local_rank], find_unused_parameters=find_unused_parameters)
    if _world() == 1 and torch.cuda.is_available() and torch.cuda.device_count() > 1:
        import warnings
        warnings.warn("magic_amd.wrap_model: several GPUs visible but local_rank == -1; nn.DataParallel is not supported (one process per "
                      "GPU: launch with torch.distributed.run) -- continuing on one device")
    return model
