"""Round 6: what do the gradients of the benchmarked step look like?  Builds bench.py's models (same init, same batches), runs one eager step per proxy
task and prints the global gradient norm, the clip factor at max_norm 5.0 and the tensors that hold most of the squared norm -- with the round 1-5
workload (zero biases, zero [stop]-node position row: MAGIC_OLD_WORKLOAD=1) and with the fixed one.
Run on the GPU box:  python profiles/micro/r06_grad_norm_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from magic_amd.host import synth  # noqa: E402
from magic_amd.host.plan import build_plan  # noqa: E402


def run(old):
    dev = torch.device("cuda", 0)
    if old:
        keep = B.GlocalTextPathCMTPreTraining
        tcfg, scfg, teacher, student, trainer = B.build_models(torch.bfloat16, dev, 0.1, 1)
        for m in (teacher, student):                 # undo the checkpoint-like small parameters
            for name, shape, kind in m.store.specs:
                v = m.store.master(name)
                if name.endswith("bias"):
                    v.zero_()
                elif kind == "ones":
                    v.fill_(1.0)
            m.store.shadow_clean = False
    else:
        tcfg, scfg, teacher, student, trainer = B.build_models(torch.bfloat16, dev, 0.1, 1)
    for i, task in enumerate(B.TASKS):
        b = synth.make_batch(task, batch_size=48, seed=1234, step=i)
        if old:
            b["gmap_pos_fts"][:, 0] = 0
        plan = build_plan(b, task, dev)
        bd = synth.batch_to(b, dev)
        trainer._zero_grad()
        out = trainer._fwd_bwd(bd, task, None, plan)
        torch.cuda.synchronize()
        g = student.store.grad
        tot = float(g.double().pow(2).sum())
        shares = []
        for name, (off, n, shape) in student.store.offsets.items():
            shares.append((float(g[off:off + n].double().pow(2).sum()) / max(tot, 1e-300), name))
        shares.sort(reverse=True)
        nrm = tot ** 0.5
        print(f"[{'old' if old else 'new'} workload] {task}: loss {float(out['loss']):.4f} grad_norm {nrm:.4e} clip_factor {min(1.0, 5.0 / (nrm + 1e-6)):.3e} "
              f"top: " + ", ".join(f"{n} {s:.3f}" for s, n in shares[:4]), flush=True)
        trainer._optimize()
        print("    after the update:", trainer.opt.grad_norm_report(), flush=True)


if __name__ == "__main__":
    run(old=True)
    run(old=False)
