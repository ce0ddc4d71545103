"""cpu_baseline against the number of torch threads on the GPU box's host (the '16 threads' of bench.py's cpu_baseline
was chosen from a round-1 measurement of another oracle build).  Same step as bench.cpu_baseline (oracle teacher forward + student forward +
MAKD + backward + clip + AdamW, B = 48, dropout live), 1 warm-up + 3 timed steps per setting.
    python profiles/micro/cpu_threads_scan.py [threads ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from magic_amd.host import synth
from magic_amd.host.params import is_no_decay
from oracle import model_ref as R
from oracle import optim_ref

threads = [int(x) for x in sys.argv[1:]] or [4, 8, 16, 32, 64, 128]
avail = len(os.sched_getaffinity(0))
print(f"cores of the box {os.cpu_count()}, usable by this process {avail}")
tcfg, scfg = bench.make_cfgs(0.1) if hasattr(bench, "make_cfgs") else (None, None)
if tcfg is None:
    from magic_amd.host.config import make_config
    kw = dict(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    tcfg = make_config(256, role="teacher", **kw)
    scfg = make_config(128, role="student", teacher_hidden_size=256, kdl=bench.KDL, **kw)
torch.manual_seed(0)
teacher, student = R.RefPretrainModel(tcfg).eval(), R.RefPretrainModel(scfg).eval()
params = list(student.parameters())
wds = [0.0 if is_no_decay(n) else 0.01 for n, _ in student.named_parameters()]
state = optim_ref.adamw_init([p.data for p in params])
rw = torch.ones(5)


def step(i, task):
    batch = synth.make_batch(task, batch_size=48, seed=4321, step=i)
    t0 = time.perf_counter()
    with torch.no_grad():
        t_out = teacher(batch, task)["outputs"]
    for p in params:
        p.grad = None
    R.DROPOUT = lambda site, x: torch.nn.functional.dropout(x, 0.1)
    try:
        out = student(batch, task, teacher_outputs=t_out, rw=rw)
    finally:
        R.DROPOUT = None
    out["loss"].backward()
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    optim_ref.clip_grad_norm(grads, 5.0)
    with torch.no_grad():
        optim_ref.adamw_step([p.data for p in params], grads, state, lr=5e-5, betas=(0.9, 0.98), eps=1e-6, weight_decay=wds)
    return time.perf_counter() - t0, sum(batch["traj_step_lens"])


for n in threads:
    if n > avail:
        print(f"{n:4d} threads: more than this process may use ({avail}), skipped")
        continue
    torch.set_num_threads(n)
    step(0, "sap")
    tt, traj = 0.0, 0
    for i, task in enumerate(("mlm", "sap", "cfp"), start=1):
        dt, k = step(i, task)
        tt += dt
        traj += k
    print(f"{n:4d} threads: {tt / 3 * 1e3:7.0f} ms/step  {traj / tt:7.1f} trajectory-steps/s", flush=True)
