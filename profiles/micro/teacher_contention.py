"""How much of the headline step is the frozen teacher's forward competing for CUs?  Times three replays of the SAME captured graphs:
(a) student graph || teacher graph (the product schedule), (b) student graphs only (teacher outputs stale -- an experiment, not a valid
training step), (c) teacher graphs only.  python profiles/micro/teacher_contention.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from magic_amd.host import lib as L, synth
from magic_amd.host.plan import build_plan

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.load()
tcfg, scfg, teacher, student, trainer = bench.build_models(torch.bfloat16, dev, 0.1, 1, 48)
pool = []
for i in range(12):
    task = bench.TASKS[i % 3]
    b = synth.make_batch(task, batch_size=48, seed=1234, step=i)
    pool.append((task, synth.batch_to(b, dev), build_plan(b, task, dev)))
for i in range(3):
    trainer.step(pool[i][1], pool[i][0], plan=pool[i][2])
torch.cuda.synchronize()
graphs = bench.capture_ring(trainer, pool, "split")
torch.cuda.synchronize()


def timeit(fn, n=60):
    for i in range(12):
        fn(graphs[i % 12])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(graphs[i % 12])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def t_only(cs):
    cs.t_graph.replay()


print(f"student || teacher : {timeit(trainer.replay_split):.3f} ms/step")
print(f"student only       : {timeit(lambda cs: cs.graph.replay()):.3f} ms/step")
print(f"teacher only       : {timeit(t_only):.3f} ms/step")
