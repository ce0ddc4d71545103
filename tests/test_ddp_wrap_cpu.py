"""The reference's DDP call shapes against these models, 2 ranks on CPU with gloo (no kernel is launched: the guard fires before
any launch, and the wrapped model's explicit backward is replaced by one that writes known per-rank gradients).

  * torch's own DistributedDataParallel around the REAL module (what the unmodified `wrap_model` of
    pretrain_src/utils/misc.py:57-71 / agent_base.py:114-116 would build) is refused loudly at the first forward -- its reducer
    would never see the gradients the HIP backward writes;
  * `magic_amd.wrap_model(model, device, local_rank)` (same signature) broadcasts rank 0's parameters, keeps `.module` and the
    `module.`-prefixed checkpoint keys, and `loss.backward()` on the real module's loss leaves the rank-averaged gradient in
    `param.grad`; `no_sync()` skips the exchange.
"""
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
    res = {"rank": rank}
    try:
        from torch.nn.parallel import DistributedDataParallel as TorchDDP
        from magic_amd.host.config import make_config
        from magic_amd.host.ddp import TorchDDPWrapperError
        from magic_amd.host.model_nav import VLNBert
        from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining, _BackwardHook
        cfg = make_config(128, teacher_hidden_size=256, vocab_size=200, num_l_layers=1, num_x_layers=1, num_pano_layers=1)

        # ---- 1. torch DDP around the real modules: refused at forward ------------------------------------------------
        for name, build in (("pretrain", lambda: GlocalTextPathCMTPreTraining(cfg, device="cpu", compute_dtype=torch.float32, seed=3)),
                            ("nav", lambda: VLNBert(None, config=cfg, device="cpu", compute_dtype=torch.float32, seed=3))):
            wrapped = TorchDDP(build(), find_unused_parameters=True)          # misc.py:63 call shape (no device_ids on CPU)
            try:
                if name == "pretrain":
                    wrapped({}, task="sap", compute_loss=True)
                else:
                    wrapped("language", {})
                res[f"refused_{name}"] = False
            except TorchDDPWrapperError as e:
                res[f"refused_{name}"] = "magic_amd.wrap_model" in str(e)

        # ---- 2. the drop-in wrapper ----------------------------------------------------------------------------------
        model = GlocalTextPathCMTPreTraining(cfg, device="cpu", compute_dtype=torch.float32, seed=100 + rank)    # ranks start different
        ddp = magic_amd.wrap_model(model, torch.device("cpu"), local_rank=rank)
        ref = GlocalTextPathCMTPreTraining(cfg, device="cpu", compute_dtype=torch.float32, seed=100)
        res["params_from_rank0"] = bool(torch.equal(model.store.flat, ref.store.flat))
        res["module_attr"] = ddp.module is model
        res["prefixed_keys"] = all(k.startswith("module.") for k in ddp.state_dict())

        def fake_backward(m=model, r=rank):          # stands in for the HIP backward: fills the flat gradient buffer
            m.store.ensure_grads()
            m.store.grad.copy_(torch.randn(m.store.total, generator=torch.Generator().manual_seed(40 + r)))
        model.backward = fake_backward
        want = sum(torch.randn(model.store.total, generator=torch.Generator().manual_seed(40 + r)) for r in range(world)) / world

        def train_step():
            for p in model.parameters():
                p.grad = None                                           # optimizer.zero_grad() of torch >= 2.0
            model._ctx = object()
            loss = _BackwardHook.apply(torch.tensor(1.5), model._anchor, model)   # what forward(compute_loss=True) returns as 'loss'
            loss.backward()                                             # unmodified loop: agent_base.py:260 / the pretrain loop
        train_step()
        p0 = next(iter(model.parameters()))
        res["averaged"] = bool(torch.allclose(model.store.grad, want, rtol=1e-6, atol=1e-6)) and p0.grad.data_ptr() == model.store.grad.data_ptr()
        with ddp.no_sync():
            train_step()
        mine = torch.randn(model.store.total, generator=torch.Generator().manual_seed(40 + rank))
        res["no_sync_local"] = bool(torch.equal(model.store.grad, mine))
        train_step()
        res["averaged_again"] = bool(torch.allclose(model.store.grad, want, rtol=1e-6, atol=1e-6))
        q.put(res)
    except Exception as e:          # surface the failure in the parent instead of a queue timeout
        res["error"] = repr(e)
        q.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_reference_wrap_model_call_shapes_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r in res:
        assert "error" not in r, r
        for k in ("refused_pretrain", "refused_nav", "params_from_rank0", "module_attr", "prefixed_keys", "averaged", "no_sync_local",
                  "averaged_again"):
            assert r[k] is True, (k, r)


def test_single_process_wrap_model_is_transparent():
    from magic_amd.host.config import make_config
    from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining
    cfg = make_config(128, vocab_size=200, num_l_layers=1, num_x_layers=1, num_pano_layers=1)
    model = GlocalTextPathCMTPreTraining(cfg, device="cpu", compute_dtype=torch.float32)
    assert magic_amd.wrap_model(model, torch.device("cpu"), local_rank=-1) is model          # misc.py:57-71 with local_rank == -1
    ddp = magic_amd.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True)
    assert ddp.module is model and sorted(k[len("module."):] for k in ddp.state_dict()) == sorted(model.state_dict())
    with pytest.raises(TypeError):
        magic_amd.DistributedDataParallel(torch.nn.Linear(2, 2))
