"""Where the HOST time of an eager (non-graph) training step goes: cProfile over resident batches of the bench's own configuration.
Run on the GPU box:  python profiles/micro/host_profile.py [steps]   -> top functions by own time + by cumulative time."""
import cProfile
import pstats
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from magic_amd.host import lib as L, synth
from magic_amd.host.plan import build_plan

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.load()
tcfg, scfg, teacher, student, trainer = bench.build_models(torch.bfloat16, dev, 0.1, 1, 48)
pool = []
for i in range(6):
    task = bench.TASKS[i % 3]
    b = synth.make_batch(task, batch_size=48, seed=1234, step=i)
    pool.append((task, synth.batch_to(b, dev), build_plan(b, task, dev)))
for i in range(6):
    trainer.step(pool[i][1], pool[i][0], plan=pool[i][2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    t, b, p = pool[i % 6]
    trainer.step(b, t, plan=p)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"eager: host issue time {t_host / steps * 1e3:.3f} ms/step, wall {t_all / steps * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    t, b, p = pool[i % 6]
    trainer.step(b, t, plan=p)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    st = pstats.Stats(pr)
    st.sort_stats(key)
    print(f"==== by {key} (per {steps} steps)")
    st.print_stats(45)
