"""StreamStep on pre-made bucketed records (no DataLoader): host time per step vs wall time per step, and the same batches as resident single-graph
replays (the schedule StreamStep captures) -- separates the loader, the host work of a step and the GPU work.  python profiles/micro/stream_graph_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from magic_amd.host import lib as L, synth
from magic_amd.host.loader import pack_bucketed
from magic_amd.host.plan import build_plan
from magic_amd.host.stream_graph import StreamStep

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.load()
_, _, teacher, student, trainer = bench.build_models(torch.bfloat16, dev, 0.1, 1, 48)
from magic_amd.host.feature_table import FeatureTable
n_vp = 4096
ftab = FeatureTable([str(i) for i in range(n_vp)], torch.randn(n_vp, 36, 768, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).to(dev))
recs = []
for task, r in bench._StreamSet(48, 1234, 36, n_vp=n_vp, bucketed=True):       # index-only batches (feature table in HBM), in-process
    r["buf"] = r["buf"].pin_memory()
    recs.append((task, r))
for i in range(3):
    task = bench.TASKS[i]
    b = synth.make_batch(task, batch_size=48, seed=1234, step=i)
    trainer.step(synth.batch_to(b, dev), task, plan=build_plan(b, task, dev))
torch.cuda.synchronize()
ss = StreamStep(trainer, feature_table=ftab)
for task, r in recs:
    ss.step(task, r)
torch.cuda.synchronize()
n = 240
t0 = time.perf_counter()
for i in range(n):
    task, r = recs[i % len(recs)]
    ss.step(task, r)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"StreamStep on resident records: host {t_host / n * 1e3:.3f} ms/step, wall {t_all / n * 1e3:.3f} ms/step, {ss.captures} bucket graphs "
      f"(index-only records: {recs[0][1]['buf'].numel() / 1e3:.0f} KB each)")
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for i in range(120):
    task, r = recs[i % len(recs)]
    ss.step(task, r)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
st.print_stats(14)
