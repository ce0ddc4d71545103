"""The launch sequence of ONE training step (paired structure as captured), with per-launch durations from HIP events: where the ~270
launches of the headline step go, in issue order."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import lib as L  # noqa: E402
from magic_amd.host import synth  # noqa: E402
from magic_amd.host.config import make_config  # noqa: E402
from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining  # noqa: E402
from magic_amd.host.plan import build_plan  # noqa: E402
from magic_amd.host.trainer import PretrainStep  # noqa: E402
import magic_amd.host.model_pretrain as MP  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
dk = dict(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
tcfg = make_config(256, role="teacher", **dk)
scfg = make_config(128, role="student", teacher_hidden_size=256, kdl=bench.KDL, **dk)
teacher = GlocalTextPathCMTPreTraining(tcfg, device=dev, compute_dtype=torch.bfloat16, seed=0)
student = GlocalTextPathCMTPreTraining(scfg, device=dev, compute_dtype=torch.bfloat16, seed=1)
tr = PretrainStep(student, teacher, lr=5e-5, betas=(0.9, 0.98), weight_decay=0.01, grad_norm=5.0, warmup_steps=10000, num_train_steps=200000)
task = sys.argv[1] if len(sys.argv) > 1 else "sap"
b = synth.make_batch(task, batch_size=48, seed=1234, step=1)
plan = build_plan(b, task, dev)
bd = synth.batch_to(b, dev)
for _ in range(3):
    tr.step(bd, task, plan=plan)
torch.cuda.synchronize()
MP.LOCKSTEP_EAGER = True
L.PROFILE.update(on=True, events=[])
tr.step(bd, task, plan=plan)
torch.cuda.synchronize()
L.PROFILE["on"] = False
tot = 0.0
for i, (name, layout, e0, e1) in enumerate(L.PROFILE["events"]):
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    print(f"{i:4d} {name}{'' if layout < 0 else ['[NT]', '[NN]', '[TN]'][layout]:6s} {us:8.1f}")
print("total_us", round(tot, 1), "launches", len(L.PROFILE["events"]))
