"""The STUDENT's launch sequence of one training step per task, in issue order, with each launch's duration (HIP events, device-side gate so
the host runs ahead).  The teacher's outputs are computed beforehand, so only the student chain + optimizer is listed.
python profiles/micro/launch_list.py [task ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import magic_amd.host.model_pretrain as MP
from magic_amd.host import lib as L, synth
from magic_amd.host.plan import build_plan

tasks = sys.argv[1:] or ["sap", "mlm", "cfp"]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.load()
tcfg, scfg, teacher, student, trainer = bench.build_models(torch.bfloat16, dev, 0.1, 1, 48)
side, trainer.side = trainer.side, None
for task in tasks:
    b = synth.make_batch(task, batch_size=48, seed=1234, step=bench.TASKS.index(task))
    bd, plan = synth.batch_to(b, dev), build_plan(b, task, dev)
    for _ in range(3):
        trainer.step(bd, task, plan=plan)
    torch.cuda.synchronize()
    t_out = trainer.teacher_forward(bd, task, plan)
    torch.cuda.synchronize()
    MP.LOCKSTEP_EAGER = True
    L.PROFILE.update(on=True, events=[])
    bench._gate(30.0)
    student.store.zero_grad()
    out = student(bd, task, compute_loss=True, teacher_outputs=t_out, rw=trainer.mkrw(), plan=plan, inputs=t_out["inputs"])
    n_fwd = len(L.PROFILE["events"])
    student.backward()
    n_bwd = len(L.PROFILE["events"])
    trainer._optimize()
    torch.cuda.synchronize()
    L.PROFILE["on"] = False
    MP.LOCKSTEP_EAGER = False
    ev = L.PROFILE["events"]
    tot = 0.0
    print(f"==== {task}: {len(ev)} profiled launches (forward {n_fwd}, backward {n_bwd - n_fwd}, optimizer {len(ev) - n_bwd}); torch-side launches are not listed")
    for i, (name, layout, e0, e1) in enumerate(ev):
        us = e0.elapsed_time(e1) * 1e3
        tot += us
        gap = ev[i - 1][3].elapsed_time(e0) * 1e3 if i else 0.0
        tag = "F" if i < n_fwd else ("B" if i < n_bwd else "O")
        print(f"{i:4d} {tag} {name + ('' if layout < 0 else '[' + ['NT', 'NN', 'TN'][layout] + ']'):44s} {us:7.1f} us   gap before {gap:6.1f}")
    span = ev[0][2].elapsed_time(ev[-1][3]) * 1e3
    print(f"sum of launches {tot:.0f} us, first start -> last end {span:.0f} us")
trainer.side = side
