"""Where the streaming step (bench.py --mode stream) spends host time: loader hand-over vs the eager step itself."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
import bench  # noqa: E402
from magic_amd.host.config import make_config  # noqa: E402
from magic_amd.host.feature_table import FeatureTable  # noqa: E402
from magic_amd.host.loader import DevicePrefetcher  # noqa: E402
from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining  # noqa: E402
from magic_amd.host.trainer import PretrainStep  # noqa: E402

dev = torch.device("cuda", 0)
dk = dict(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
teacher = GlocalTextPathCMTPreTraining(make_config(256, role="teacher", **dk), device=dev, compute_dtype=torch.bfloat16, seed=0)
student = GlocalTextPathCMTPreTraining(make_config(128, role="student", teacher_hidden_size=256, kdl=bench.KDL, **dk), device=dev,
                                       compute_dtype=torch.bfloat16, seed=1)
tr = PretrainStep(student, teacher, lr=5e-5, betas=(0.9, 0.98), weight_decay=0.01, grad_norm=5.0, warmup_steps=10000, num_train_steps=200000)
n_vp = 4096
ftab = FeatureTable([str(i) for i in range(n_vp)], torch.randn(n_vp, 36, 768).to(torch.bfloat16).to(dev))
N = 150
for pin in (True, False):
    ds = bench._StreamSet(48, 1234, N + 2, n_vp=n_vp)
    dl = torch.utils.data.DataLoader(ds, batch_size=None, num_workers=12, pin_memory=pin, prefetch_factor=2)
    feed = iter(DevicePrefetcher(dl, dev))
    t_feed = t_step = 0.0
    for i in range(N):
        a = time.perf_counter()
        task, b, plan = next(feed)
        c = time.perf_counter()
        b["view_table"] = ftab
        tr.step(b, task, plan=plan)
        d = time.perf_counter()
        if i >= 20:
            t_feed += c - a
            t_step += d - c
    torch.cuda.synchronize()
    print(f"pin_memory={pin}: next(feed) {t_feed / (N - 20) * 1e3:.2f} ms, trainer.step (host) {t_step / (N - 20) * 1e3:.2f} ms")
