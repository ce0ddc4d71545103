import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
import torch, magic_amd
import bench
from magic_amd.host import synth
from magic_amd.host.config import make_config
from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining
from magic_amd.host.plan import build_plan
from magic_amd.host.trainer import PretrainStep
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda")
tcfg = make_config(256, role="teacher"); scfg = make_config(128, role="student", teacher_hidden_size=256, kdl=bench.KDL)
t = GlocalTextPathCMTPreTraining(tcfg, device=dev, seed=0); s = GlocalTextPathCMTPreTraining(scfg, device=dev, seed=1)
tr = PretrainStep(s, t)
for task in ("sap",):
    b0 = synth.make_batch(task, batch_size=48, seed=1)
    plan = build_plan(b0, task, dev)
    b = synth.batch_to(b0, dev)
    for _ in range(3): tr.step(b, task, plan=plan)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.step(b, task, plan=plan)
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
    from collections import Counter
    c = Counter()
    for e in ev:
        if e.name in ("aten::copy_", "aten::zero_", "aten::fill_", "aten::sum", "aten::cat", "aten::index", "aten::mul", "aten::div", "aten::add", "aten::clone", "aten::randint", "aten::randn", "aten::softmax", "aten::_to_copy", "aten::zeros", "aten::select", "aten::add_", "aten::mul_"):
            st = [f for f in (e.stack or []) if "magic" in f or "host/" in f]
            c[(e.name, st[0][-70:] if st else "?")] += 1
    for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:50]:
        print(v, k)
