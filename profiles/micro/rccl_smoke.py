"""RCCL smoke on ONE GPU (world_size 1 process group, backend nccl = RCCL): the collective calls of trainer.GradSync -- chunked fp32 all-reduce of
the flat gradient, all_gather_into_tensor of int64 row ids + fp32 rows -- on the exchange stream, and one data-parallel-structured step (two
student graphs with the bucket exchange between them, MAGIC_FORCE_SPLIT_GRAPH).  No multi-GPU node is available to this build; this checks that
the library initialises and that the calls are well-formed, nothing about scaling."""
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ["MAGIC_FORCE_SPLIT_GRAPH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

import bench
from magic_amd.host import synth
from magic_amd.host.plan import build_plan

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
_, _, teacher, student, trainer = bench.build_models(torch.bfloat16, dev, 0.1, 1, 48)
sync = trainer.sync
sync.stream = torch.cuda.Stream()
student.store.ensure_grads()
student.store.grad.normal_()
ref = student.store.grad.clone()
sync._on_side(lambda: sync._ranges([(0, student.store.total)]))
ids = torch.unique(torch.randint(0, 50265, (3000,), device=dev))
sync._on_side(lambda: sync._sparse_rows(ids))
sync.join()
torch.cuda.synchronize()
assert torch.equal(student.store.grad, ref), "world_size 1: every collective is the identity"
print("RCCL collectives (all_reduce fp32 chunks, all_gather_into_tensor int64 / fp32) ok on", torch.cuda.get_device_name(0))
pool = []
for i in range(3):
    task = bench.TASKS[i]
    b = synth.make_batch(task, batch_size=48, seed=1234, step=i)
    pool.append((task, synth.batch_to(b, dev), build_plan(b, task, dev)))
for task, b, plan in pool:
    trainer.step(b, task, plan=plan)
torch.cuda.synchronize()
graphs = bench.capture_ring(trainer, pool, "split")
for _ in range(2):
    for cs in graphs:
        trainer.replay_split(cs)
torch.cuda.synchronize()
print("data-parallel launch structure (two student graphs per step, exchange between them) replayed:", len(graphs), "graphs, split =", graphs[0].graph2 is not None)
dist.destroy_process_group()
