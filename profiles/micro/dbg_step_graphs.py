"""debugging aid: per-parameter gradient comparison of one teacher-forced rollout, eager vs captured step instances"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa
from magic_amd.host.config import make_config
from magic_amd.host.model_nav import VLNBert
from magic_amd.host.nav_rollout import NavRollout
from magic_amd.host.synth_env import SynthNavEnv

DEV = "cuda"
cfg = make_config(128, role="student", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, num_l_layers=2, num_x_layers=1, num_pano_layers=1)
m = VLNBert(None, role="student", config=cfg, device=DEV, compute_dtype=torch.float32, seed=3)
m.train()
B, T = 4, 6
env = SynthNavEnv(batch_size=B, n_scans=2, nodes_per_scan=30, seed=11, instr_len=(6, 14), vocab=(3, 290), path_hops=(2, 4))
table = torch.from_numpy(env.feature_table).to(DEV)
eager = NavRollout(m, table, max_action_len=T)
graph = NavRollout(m, table, max_action_len=T, graphs=True, Lcap=16)
for it in range(3):
    batch = [env._draw_episode() for _ in range(B)]
    res = []
    for ro in (eager, graph):
        m.store.zero_grad()
        r = ro.run(env, env.reset(batch=batch, features=False), feedback="teacher", train_ml=1.0)
        r["loss"].backward()
        torch.cuda.synchronize()
        res.append((float(r["loss"].detach()), m.store.grad.clone(), r["n_steps"]))
        del r
    (le, ge, n), (lg, gg, _) = res
    print(f"iteration {it}: steps {n} loss eager {le:.6f} graph {lg:.6f}  max|dgrad| {(ge - gg).abs().max().item():.3e} (max|g| {ge.abs().max().item():.3e})", graph.graph_report())
    worst = []
    for name, p in m.named_parameters():
        off, cnt, _ = m.store.offsets[name]
        a, b = ge[off:off + cnt], gg[off:off + cnt]
        worst.append(((a - b).abs().max().item(), a.abs().max().item(), name))
    for d, s, name in sorted(worst, reverse=True)[:12]:
        print(f"   {name:70s} max|diff| {d:.3e}  max|eager| {s:.3e}")
