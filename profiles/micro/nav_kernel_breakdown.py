"""Per-(kernel, shape) kernel time + launch count of ONE navigator fine-tuning iteration (bench_nav.py's default workload), and the
host-side wall time of the same iteration un-instrumented.  HIP events around every launch (lib.PROFILE), shapes = the leading integer
arguments of the C-ABI call.  `python profiles/micro/nav_kernel_breakdown.py [--icod] [--top 60]`"""
import argparse
import collections
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import lib as L  # noqa: E402
from magic_amd.host.config import make_config  # noqa: E402
from magic_amd.host.model_nav import VLNBert  # noqa: E402
from magic_amd.host.nav_rollout import NavRollout  # noqa: E402
from magic_amd.host.synth_env import SynthNavEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--top", type=int, default=70)
ap.add_argument("--hidden", type=int, default=768)
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--graphs", action="store_true")
ap.add_argument("--ahead", action="store_true", help="plan the next batch's teacher-forced rollout right after the optimizer launch (NavRollout.plan_ahead(thread=False))")
ap.add_argument("--icod", action="store_true", help="BASELINE config 3: MAGIC-S student + trainable MAGIC-L teacher, R2R lengths")
a = ap.parse_args()
dev = torch.device("cuda", 0)
teacher = t_opt = kd = None
if a.icod:
    from types import SimpleNamespace
    scfg = make_config(128, role="student", teacher_hidden_size=768, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    model = VLNBert(None, role="student", config=scfg, device=dev, compute_dtype=torch.bfloat16, seed=0)
    tcfg = make_config(768, role="teacher", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    teacher = VLNBert(SimpleNamespace(train_kdl_teacher=True, train_kdl=True), role="teacher", config=tcfg, device=dev, compute_dtype=torch.bfloat16, seed=1)
    teacher.train()
    from magic_amd.host.trainer import FlatTorchAdamW
    t_opt = FlatTorchAdamW(teacher.store, lr=1e-5)
    kd = dict(alpha=0.5, t_alpha=0.5, temperature=2.0, decay=0.7)
    T, LEN, HOPS = 15, (20, 80), (4, 7)
else:
    cfg = make_config(a.hidden, role="teacher", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    model = VLNBert(None, role="student", config=cfg, device=dev, compute_dtype=torch.bfloat16, seed=0)
    T, LEN, HOPS = 28, (100, 512), (8, 15)
model.train()
from magic_amd.host.trainer import FlatTorchAdamW
opt = FlatTorchAdamW(model.store, lr=1e-5)
mk = lambda: SynthNavEnv(batch_size=16, n_scans=6, nodes_per_scan=64, seed=1234, instr_len=LEN, path_hops=HOPS)
env, env2 = mk(), mk()
table = torch.from_numpy(env.feature_table).to(dev).to(torch.bfloat16)
ro = NavRollout(model, table, teacher=teacher, kd=kd, train_teacher=a.icod, max_action_len=T, expert_policy="spl" if a.icod else "ndtw", graphs=a.graphs, Lcap=LEN[1])
rng = np.random.default_rng(0)


EV = {}


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    EV[name] = e


PIPE = {"ahead": None, "obs": None}


def iteration():
    mark("start")
    opt.zero_grad()
    if t_opt is not None:
        t_opt.zero_grad()
    ahead = PIPE["ahead"]
    obs = PIPE["obs"] if ahead is not None else env.reset(features=False)
    batch = env.batch
    rw = (torch.softmax(torch.randn(T, 5, device=dev) / 4.0, -1) * 5) if a.icod else None
    r2, r1 = ro.run_interleaved([((env2, env2.reset(batch=batch, features=False)), dict(feedback="sample", train_ml=1.0, rw_seq=rw, sample_draws=rng.uniform(size=(T, 16)))),
                                 ((env, obs), dict(feedback="teacher", train_ml=0.2, rw_seq=rw, ahead=ahead))])
    t_f = time.perf_counter()
    mark("fwd_end")                  # (the current stream has joined the rollouts' lanes: everything of the forward phase is behind this event)
    (r1["loss"] + r2["loss"]).backward(retain_graph=a.icod)
    if a.icod:
        (r1["t_loss"] + r2["t_loss"]).backward()
        t_opt.step(max_norm=40.0)
    t_b = time.perf_counter()
    mark("bwd_end")
    opt.step(max_norm=40.0)
    mark("end")
    if a.ahead:
        PIPE["obs"] = env.reset(features=False)
        PIPE["ahead"] = ro.plan_ahead(env, PIPE["obs"], thread=False)
    return r1["decisions"] + r2["decisions"], t_f, t_b


for _ in range(5 if a.graphs else 2):
    t0 = time.perf_counter()
    iteration()
    torch.cuda.synchronize()
    print(f"warm-up iteration: {1e3 * (time.perf_counter() - t0):.0f} ms", ro.graph_report() if a.graphs else "")
from magic_amd.host import nav_rollout as _NR
_NR._T["acc"].clear()
if os.environ.get("NAV_GC_FREEZE"):
    import gc
    gc.collect()
    gc.freeze()            # the model, the captured instances and their graphs leave the collector's generations: later collections scan only what an iteration made
import gc as _gc
_gc.callbacks.append(lambda phase, info: print(f"   [gc] generation {info['generation']} collected {info.get('collected', 0)}", flush=True) if phase == "stop" and info["generation"] >= 1 else None)
for _ in range(a.iters):
    t0 = time.perf_counter()
    dec, t_f, t_b = iteration()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"iteration: {1e3 * (t2 - t0):.1f} ms wall ({dec} decisions); host: rollouts {1e3 * (t_f - t0):.1f}  backward {1e3 * (t_b - t_f):.1f}  "
          f"clip+opt (+ plan-ahead) {1e3 * (t1 - t_b):.1f}  drain {1e3 * (t2 - t1):.1f} | GPU timeline: forward done at {EV['start'].elapsed_time(EV['fwd_end']):.1f} ms, "
          f"backward done at {EV['start'].elapsed_time(EV['bwd_end']):.1f}, optimizer at {EV['start'].elapsed_time(EV['end']):.1f}")

if _NR._T["on"]:
    print("host sections of steps() over the timed iterations (ms per iteration):")
    for k, v in sorted(_NR._T["acc"].items(), key=lambda kv: -kv[1]):
        print(f"  {k:38s} {1e3 * v / a.iters:7.2f}")

L.PROFILE.update(on=True, events=[], shapes=[])
dec, _, _ = iteration()
torch.cuda.synchronize()
L.PROFILE["on"] = False
by = collections.defaultdict(lambda: [0.0, 0])
byname = collections.defaultdict(lambda: [0.0, 0])
sh = L.PROFILE["shapes"]
plain = [e for e in L.PROFILE["events"] if "+" not in e[0]]
grouped = [e for e in L.PROFILE["events"] if "+" in e[0]]
assert len(plain) == len(sh), (len(plain), len(sh))
for (name, layout, e0, e1), s in zip(plain, sh):
    t = e0.elapsed_time(e1)
    k = (name, s)
    by[k][0] += t; by[k][1] += 1
    byname[name][0] += t; byname[name][1] += 1
for name, layout, e0, e1 in grouped:
    t = e0.elapsed_time(e1)
    by[(name, ())][0] += t; by[(name, ())][1] += 1
    byname[name][0] += t; byname[name][1] += 1
tot = sum(v[0] for v in byname.values())
n = sum(v[1] for v in byname.values())
print(f"\ninstrumented iteration: {n} launches, {tot:.1f} ms of kernel time, {dec} decisions")
print("\n== by entry point ==")
for k, (t, c) in sorted(byname.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:28s} {t:8.2f} ms {100 * t / tot:5.1f} %  {c:5d} launches  {1e3 * t / c:7.1f} us avg")
print("\n== by entry point and leading integer arguments ==")
for (name, s), (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:a.top]:
    print(f"{name:24s} {str(s):44s} {t:7.2f} ms {100 * t / tot:5.1f} %  {c:4d} x {1e3 * t / c:7.1f} us")
