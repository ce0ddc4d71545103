"""cProfile of the eager training step on the host (the step is host-bound without graph replay: where do the ~15 us per launch go)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
import bench  # noqa: E402
from magic_amd.host import synth  # noqa: E402
from magic_amd.host.config import make_config  # noqa: E402
from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining  # noqa: E402
from magic_amd.host.plan import build_plan  # noqa: E402
from magic_amd.host.trainer import PretrainStep  # noqa: E402

dev = torch.device("cuda", 0)
dk = dict(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
teacher = GlocalTextPathCMTPreTraining(make_config(256, role="teacher", **dk), device=dev, compute_dtype=torch.bfloat16, seed=0)
student = GlocalTextPathCMTPreTraining(make_config(128, role="student", teacher_hidden_size=256, kdl=bench.KDL, **dk), device=dev,
                                       compute_dtype=torch.bfloat16, seed=1)
tr = PretrainStep(student, teacher, lr=5e-5, betas=(0.9, 0.98), weight_decay=0.01, grad_norm=5.0, warmup_steps=10000, num_train_steps=200000)
pool = []
for i, task in enumerate(("mlm", "sap", "cfp")):
    b = synth.make_batch(task, batch_size=48, seed=1234, step=i)
    pool.append((task, synth.batch_to(b, dev), build_plan(b, task, dev)))


def run(n):
    for s in range(n):
        task, b, plan = pool[s % 3]
        tr.step(b, task, plan=plan)


run(6)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(30)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time {(t1 - t0) / 30 * 1e3:.2f} ms/step, wall {(t2 - t0) / 30 * 1e3:.2f} ms/step")
cProfile.run("run(30)", "/tmp/step.prof")
pstats.Stats("/tmp/step.prof").sort_stats("tottime").print_stats(45)
