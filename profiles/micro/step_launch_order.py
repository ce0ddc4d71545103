"""Round 6: the launches of ONE benchmarked step in issue order with their durations (eager, the captured graphs' launch structure, teacher serialised on
the main stream, a device-side gate so the host runs ahead): which launches sit at the two ends of the step, where nothing overlaps them.
  python profiles/micro/step_launch_order.py [task]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from magic_amd.host import lib as L, ops as O, synth  # noqa: E402
from magic_amd.host.plan import build_plan  # noqa: E402
import magic_amd.host.model_pretrain as MP  # noqa: E402


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "sap"
    dev = torch.device("cuda", 0)
    tcfg, scfg, teacher, student, trainer = B.build_models(torch.bfloat16, dev, 0.1, 1)
    pool = []
    for i in range(6):
        task = B.TASKS[i % 3]
        b = synth.make_batch(task, batch_size=48, seed=1234, step=i)
        pool.append((task, synth.batch_to(b, dev), build_plan(b, task, dev)))
    for i in range(6):
        trainer.step(pool[i][1], pool[i][0], plan=pool[i][2])
    torch.cuda.synchronize()
    MP.LOCKSTEP_EAGER = True
    trainer.side = None
    task, b, plan = next(p for p in pool if p[0] == want)
    L.PROFILE.update(on=True, events=[])
    B._gate(30.0)
    trainer.step(b, task, plan=plan)
    torch.cuda.synchronize()
    L.PROFILE["on"] = False
    ev = L.PROFILE["events"]
    t0 = ev[0][2]
    tot = 0.0
    for i, (name, layout, e0, e1) in enumerate(ev):
        d = e0.elapsed_time(e1) * 1e3
        tot += d
        print(f"{i:4d}  +{t0.elapsed_time(e0) * 1e3:8.1f} us  {d:7.1f} us  {name}{'' if layout < 0 else ['[NT]', '[NN]', '[TN]'][layout]}")
    print(f"{len(ev)} launches, {tot:.1f} us of launch time, {t0.elapsed_time(ev[-1][3]) * 1e3:.1f} us first start to last end ({want} step)")


if __name__ == "__main__":
    main()
