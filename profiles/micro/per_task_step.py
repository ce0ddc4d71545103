"""graph-replay time of the headline step per task (split schedule: student graph || teacher graph of the next batch)"""
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
_, _, teacher, student, trainer = bench.build_models(torch.bfloat16, dev, 0.1, 1, 48)
for task in bench.TASKS:
    pool = []
    for i in range(6):
        b = synth.make_batch(task, batch_size=48, seed=1234, step=3 * i + bench.TASKS.index(task))
        pool.append((task, synth.batch_to(b, dev), build_plan(b, task, dev)))
    for t, b, p in pool[:2]:
        trainer.step(b, t, plan=p)
    torch.cuda.synchronize()
    graphs = bench.capture_ring(trainer, pool, "split")
    for cs in graphs:
        trainer.replay_split(cs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 60
    for i in range(n):
        trainer.replay_split(graphs[i % len(graphs)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    for i in range(n):
        graphs[i % len(graphs)].graph.replay()
    torch.cuda.synchronize()
    ds = (time.perf_counter() - t0) / n * 1e3
    print(f"{task}: {dt:.3f} ms/step with the teacher alongside, student graph alone {ds:.3f} ms")
