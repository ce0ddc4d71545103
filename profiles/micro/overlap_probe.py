"""What does a concurrent stream cost the student's step?  Replays the captured STUDENT graphs (teacher outputs stale: an experiment) next to
(a) nothing, (b) the real teacher graph, (c) a graph of N tiny one-workgroup launches (dispatch pressure only), (d) a graph of N launches that
stream 64 MB each through HBM/L2 with few workgroups (memory pressure, few CUs).  python profiles/micro/overlap_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from magic_amd.host import lib as L, ops as O, synth
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
side = trainer.side


def side_graph(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
        fn()
    return g


tiny = torch.zeros(64, 128, dtype=torch.bfloat16, device=dev)
big = torch.zeros(32 << 20, dtype=torch.bfloat16, device=dev)        # 64 MB
mid = torch.zeros(3840, 256, dtype=torch.bfloat16, device=dev)


def many(n, x):
    def f():
        for _ in range(n):
            O.dact(x, x, 2, out=x)
    return f


others = {
    "nothing": None,
    "86 tiny launches (1 workgroup)": side_graph(many(86, tiny)),
    "300 tiny launches": side_graph(many(300, tiny)),
    "86 launches over [3840, 256] (teacher-sized elementwise)": side_graph(many(86, mid)),
    "16 launches streaming 64 MB": side_graph(many(16, big)),
}


def timeit(other, n=60):
    def step(cs):
        main = torch.cuda.current_stream()
        if other is not None:
            side.wait_stream(main)
        cs.graph.replay()
        if other is not None:
            with torch.cuda.stream(side):
                if other == "teacher":
                    cs.t_graph.replay()
                else:
                    other.replay()
            main.wait_stream(side)
    for i in range(12):
        step(graphs[i % 12])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(graphs[i % 12])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def alone(g, n=60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(side):
            g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"student || real teacher (product schedule, teacher one step ahead): {timeit('teacher'):.3f} ms/step")
for name, g in others.items():
    extra = f" (alone {alone(g):.3f} ms)" if g is not None else ""
    print(f"student || {name}: {timeit(g):.3f} ms/step{extra}")
