"""The frozen TEACHER's forward (H=256) launch by launch with durations (HIP events behind a device-side gate).  python profiles/micro/teacher_launch_list.py [task]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import magic_amd.host.model_pretrain as MP
from magic_amd.host import lib as L, synth
from magic_amd.host.plan import build_plan

task = (sys.argv[1:] or ["sap"])[0]
dev = torch.device("cuda", 0)
L.load()
tcfg, scfg, teacher, student, trainer = bench.build_models(torch.bfloat16, dev, 0.1, 1, 48)
b = synth.make_batch(task, batch_size=48, seed=1234, step=bench.TASKS.index(task))
bd, plan = synth.batch_to(b, dev), build_plan(b, task, dev)
for _ in range(3):
    trainer.teacher_forward(bd, task, plan)
torch.cuda.synchronize()
MP.LOCKSTEP_EAGER = True
L.PROFILE.update(on=True, events=[])
bench._gate(30.0)
trainer.teacher_forward(bd, task, plan)
torch.cuda.synchronize()
L.PROFILE["on"] = False
ev = L.PROFILE["events"]
tot = 0.0
for i, (name, layout, e0, e1) in enumerate(ev):
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    print(f"{i:4d} {name + ('' if layout < 0 else '[' + ['NT', 'NN', 'TN'][layout] + ']'):44s} {us:7.1f} us")
print(f"{len(ev)} launches, sum {tot:.0f} us, span {ev[0][2].elapsed_time(ev[-1][3]) * 1e3:.0f} us")
