"""which parameter tensors of the headline training step are NOT bitwise reproducible run to run (same weights, same batch, same dropout seed):
the launches that still reduce through order-dependent fp32 atomics.  python profiles/micro/det_probe.py [task ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
import bench  # noqa: E402
from magic_amd.host import synth  # noqa: E402

dev = torch.device("cuda")
tcfg, scfg, teacher, student, trainer = bench.build_models(torch.bfloat16, dev, 0.1, 1, 48)
tasks = sys.argv[1:] or ["sap", "mlm", "cfp"]
for task in tasks:
    from magic_amd.host.plan import build_plan
    hb = synth.make_batch(task, batch_size=48, seed=5, step=0)
    plan = build_plan(hb, task, dev)
    batch = synth.batch_to(hb, dev)
    grads = []
    for rep in range(4):
        student.store.zero_grad()
        trainer._rng_counter.zero_()
        trainer._fwd_bwd(batch, task, None, plan)
        torch.cuda.synchronize()
        grads.append(student.store.grad.clone())
    bad = []
    for name, (off, n, shape) in student.store.offsets.items():
        a = grads[0][off:off + n]
        if any(not torch.equal(a, g[off:off + n]) for g in grads[1:]):
            d = max(float((a - g[off:off + n]).abs().max()) for g in grads[1:])
            bad.append((name, d, float(a.abs().max())))
    print(f"[{task}] {len(bad)} of {len(student.store.offsets)} parameter tensors differ between 4 runs of the same step")
    for b in bad:
        print(f"    {b[0]:70s} max|delta| {b[1]:.2e} (max|g| {b[2]:.2e})")
