"""Where does the teacher's overlap cost the student?  Three phases of graph replays under `rocprofv3 --kernel-trace` (a: student || teacher,
b: student only, c: teacher only), separated by 60 ms of idle time; profiles/micro/overlap_trace_report.py splits the trace at the pauses and
prints, per phase and queue: kernels, summed kernel time, summed gaps between consecutive kernels of the queue, per-kernel-name mean durations.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovt -- python3 profiles/micro/overlap_trace.py"""
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


def phase(fn, n=36):
    for i in range(12):
        fn(graphs[i % 12])
    torch.cuda.synchronize()
    time.sleep(0.06)
    t0 = time.perf_counter()
    for i in range(n):
        fn(graphs[i % 12])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    time.sleep(0.06)
    return dt


print(f"a student || teacher : {phase(trainer.replay_split):.3f} ms/step")
print(f"b student only       : {phase(lambda cs: cs.graph.replay()):.3f} ms/step")
print(f"c teacher only       : {phase(lambda cs: cs.t_graph.replay()):.3f} ms/step")
