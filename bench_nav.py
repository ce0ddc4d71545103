#!/usr/bin/env python
"""bench_nav.py -- navigator step loop throughput (BASELINE config 5: RxR-length instructions, MAGIC-L fine-tune).

NOT the headline line (that is bench.py / config 1): the measurement of the next §8 row, to the same bar.
One "step" = one fine-tuning iteration of `Seq2SeqAgent.train` (map_nav_src/r2r/agent_base.py:215-296) on one batch of
synthetic episodes per rank: teacher-forced rollout (train_ml = ml_weight 0.2) + DAgger rollout (feedback 'sample',
train_ml 1) -> `loss.backward()` -> clip 40 -> `torch.optim.AdamW.step()` (the reference's own optimizer, agent_base.py:128-139).
A trajectory-step = one (episode, t) decision of either rollout.  Episodes: instructions U{100..512} tokens,
ground-truth paths of 8..15 hops on synthetic connectivity graphs, 36 x 768 features gathered from the HBM-resident table.

`python bench_nav.py --gpus N --steps K --warmup W` (N > 1 under torch.distributed.run); rank 0 prints ONE JSON line with the
same keys as bench.py.  `roofline`: dense-contraction kernels (MFMA), FLOPs counted from the true ragged sizes, durations from HIP
events around every launch of one instrumented iteration.  `cpu_baseline`: the reference-style per-sample loop + fp32 CPU oracle
model (oracle/rollout_ref.py, oracle/nav_ref.py) on a bounded sample.  `host_loop`: the same HIP model driven by the
reference-style per-sample loop (GraphMap.update_node_embed / get_node_embed on device tensors, .item() per sample) for the
teacher-forced rollout -- what the index plans replace.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import magic_amd  # noqa: E402,F401
from magic_amd.host import lib as L  # noqa: E402
from magic_amd.host import ops as O  # noqa: E402
from magic_amd.host.config import make_config  # noqa: E402
from magic_amd.host.model_nav import VLNBert  # noqa: E402
from magic_amd.host.nav_rollout import NavRollout  # noqa: E402
from magic_amd.host.synth_env import SynthNavEnv  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0


def make_env(a, seed):
    return SynthNavEnv(batch_size=a.batch, n_scans=a.scans, nodes_per_scan=a.nodes, seed=seed, instr_len=(a.instr_min, a.instr_max),
                       path_hops=(a.hops_min, a.hops_max))


def cpu_baseline(a, cfg, budget=25.0):
    from oracle import rollout_ref as R
    from oracle.nav_ref import RefVLNBert
    ncores = max(1, min(16, len(os.sched_getaffinity(0))))
    torch.set_num_threads(ncores)
    torch.manual_seed(0)
    model = RefVLNBert(cfg).eval()
    env = SynthNavEnv(batch_size=min(a.batch, 4), n_scans=2, nodes_per_scan=a.nodes, seed=99, instr_len=(a.instr_min, a.instr_max),
                      path_hops=(a.hops_min, a.hops_max))
    dec, tt, n = 0, 0.0, 0
    while tt < budget and n < 3:
        t0 = time.perf_counter()
        out = R.rollout(env, model, env.reset(), feedback="teacher", train_ml=0.2, max_action_len=a.max_action_len)
        out["loss"].backward()
        tt += time.perf_counter() - t0
        dec += sum(int((s["targets"] != -100).sum()) for s in out["steps"])
        n += 1
    return {"value": round(dec / tt, 2), "unit": "trajectory-steps/sec", "cores": ncores, "kind": "port",
            "sample": f"{n} teacher-forced rollouts + backward, B={env.batch_size}, fp32 torch CPU oracle model under the reference-style host loop "
                      f"(no optimizer step), {tt / n:.1f} s each"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=4)                  # (iteration 1 sees the shapes, 2 captures the step instances, 3-4 top the pools up)
    ap.add_argument("--plan-ahead", choices=("off", "inline", "thread"), default="off",
                    help="build the NEXT batch's teacher-forced step plans (model-independent: the actions are the expert's) ahead of time "
                         "(NavRollout.plan_ahead): inline = on the main thread right after this iteration's backward and optimizer launches, while "
                         "the GPU drains them; thread = on a helper thread (measured round 5: 210 vs 200 ms per iteration -- the helper's Python "
                         "fights the backward's callbacks for the GIL); off = step by step inside the iteration (default: with the rollouts on "
                         "gradient lanes the forward phase is bound by the sampled rollout's step-by-step dependency, not by the host's planning -- inline "
                         "measured 161.1 vs 161.8 ms per iteration on config 5 and slower on the host-bound ICoD iteration)")
    ap.add_argument("--optimizer", choices=("flat", "torch"), default="flat",
                    help="flat: trainer.FlatTorchAdamW -- torch.optim.AdamW's arithmetic + clip_grad_norm_ as one sum-of-squares and one update launch "
                         "over the model's flat buffers; torch: torch.optim.AdamW + torch.nn.utils.clip_grad_norm_ over the parameter list")
    ap.add_argument("--no-graphs", action="store_true",
                    help="launch every kernel of the training steps eagerly instead of replaying captured step instances (host/step_graphs.py)")
    ap.add_argument("--batch", type=int, default=16)                  # run_rxr_kdl_valid.sh:39
    ap.add_argument("--hidden", type=int, default=768)                # MAGIC-L
    ap.add_argument("--instr-min", type=int, default=100)
    ap.add_argument("--instr-max", type=int, default=512)
    ap.add_argument("--hops-min", type=int, default=8)
    ap.add_argument("--hops-max", type=int, default=15)
    ap.add_argument("--max-action-len", type=int, default=28)         # run_rxr_kdl_valid.sh:36
    ap.add_argument("--scans", type=int, default=6)
    ap.add_argument("--nodes", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-host-loop", action="store_true")
    ap.add_argument("--icod", action="store_true",
                    help="BASELINE config 3: MAGIC-L teacher (trainable, --teacher-hidden) + student (--hidden) co-training -- MAKD t2s for the "
                         "student, reverse s2t for the teacher, both losses back-propagated, two optimizers (agent_base.py:260-279)")
    ap.add_argument("--teacher-hidden", type=int, default=768)
    ap.add_argument("--sequential-rollouts", action="store_true",
                    help="run the iteration's two rollouts strictly one after the other (the reference's order) instead of interleaving their steps")
    ap.add_argument("--fuse-rollouts", action="store_true",
                    help="run the iteration's two rollouts as ONE batch of 2B episodes (per-episode feedback / loss weight, text encoded once) "
                         "instead of one after the other as the reference does: 27 %% fewer launches and 11 %% less kernel time, but the same wall "
                         "time -- every step then waits for the 'sample' half's action copy and the 2B-episode plan (measured 329 vs 332 ms)")
    ap.add_argument("--graph", action="store_true",
                    help="--mode eval: the decision step as ONE HIP graph with padded static shapes (host/nav_graph.GreedyNavigator) instead of eager launches")
    ap.add_argument("--kmax", type=int, default=64, help="--graph: map tokens per episode the static shapes provide for")
    ap.add_argument("--mode", default="train", choices=["train", "eval"],
                    help="eval: greedy inference rollouts (feedback 'argmax', no grad, model.eval()) -- decisions/s and ms per step")
    a = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    L.load()
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    cfg = make_config(a.hidden, role="teacher", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    model = VLNBert(None, role="student", config=cfg, device=dev, compute_dtype=dtype, seed=0)
    if world > 1:
        dist.broadcast(model.store.flat, src=0)
    model.train() if a.mode == "train" else model.eval()
    from magic_amd.host.trainer import FlatTorchAdamW
    mk_opt = (lambda m: FlatTorchAdamW(m.store, lr=1e-5)) if (a.optimizer == "flat" and a.mode != "eval") else (lambda m: torch.optim.AdamW(m.parameters(), lr=1e-5))
    opt = mk_opt(model)                                               # agent_base.py:128-129
    env = make_env(a, 1234 + rank)
    env2 = make_env(a, 1234 + rank)            # same scans and features (deterministic in the seed): the second rollout's stepper
    table = torch.from_numpy(env.feature_table).to(dev).to(dtype)
    teacher = t_opt = None
    if a.icod:
        from types import SimpleNamespace
        scfg = make_config(a.hidden, role="student", teacher_hidden_size=a.teacher_hidden, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        model = VLNBert(None, role="student", config=scfg, device=dev, compute_dtype=dtype, seed=0)
        tcfg = make_config(a.teacher_hidden, role="teacher", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        teacher = VLNBert(SimpleNamespace(train_kdl_teacher=True, train_kdl=True), role="teacher", config=tcfg, device=dev, compute_dtype=dtype, seed=1)
        if world > 1:
            dist.broadcast(model.store.flat, src=0)
            dist.broadcast(teacher.store.flat, src=0)
        model.train()
        teacher.train()
        opt = mk_opt(model)
        t_opt = mk_opt(teacher)                                       # agent_base.py:133-139
    kd = dict(alpha=0.5, t_alpha=0.5, temperature=2.0, decay=0.7) if a.icod else None      # run_r2r_kdl_valid.sh:97-104
    ro = NavRollout(model, table, teacher=teacher, kd=kd, train_teacher=a.icod, max_action_len=a.max_action_len,
                    expert_policy="ndtw" if not a.icod else "spl",     # run_rxr_kdl_valid.sh:29 / run_r2r_kdl_valid.sh:29
                    graphs=(a.mode == "train" and not a.no_graphs), Lcap=a.instr_max)
    rng = np.random.default_rng(rank)

    gnav = None
    if a.mode == "eval" and a.graph:
        from magic_amd.host.nav_graph import GreedyNavigator
        gnav = GreedyNavigator(model, table, a.batch, Lmax=a.instr_max, Kmax=a.kmax, Tmax=a.max_action_len)

    def eval_iteration():
        obs = env.reset(features=False)
        with torch.no_grad():
            r = gnav.run(env, obs) if gnav is not None else ro.run(env, obs, feedback="argmax", train_ml=1.0, grad=False)
        counters["rollout_steps"] += r["n_steps"]
        return r["decisions"]
    counters = {"rollout_steps": 0}

    pipe = {"ahead": None, "obs": None}

    def iteration():
        if a.mode == "eval":
            return eval_iteration()
        opt.zero_grad()
        if t_opt is not None:
            t_opt.zero_grad()
        ahead = pipe["ahead"]
        obs = pipe["obs"] if ahead is not None else env.reset(features=False)
        batch = env.batch
        rw = None
        planned = (not a.fuse_rollouts or a.icod) and not a.sequential_rollouts
        if a.icod:      # MKRW: softmax(randn(5) / rw_temp) * 5 per step (agent.py:866-871)
            rw = torch.softmax(torch.randn(a.max_action_len, 5, device=dev) / 4.0, -1) * 5
        if (not a.fuse_rollouts or a.icod) and a.sequential_rollouts:
            r1 = ro.run(env, obs, feedback="teacher", train_ml=0.2, rw_seq=rw)
            obs = env.reset(batch=batch, features=False)
            r2 = ro.run(env, obs, feedback="sample", train_ml=1.0, sample_draws=rng.uniform(size=(a.max_action_len, a.batch)), rw_seq=rw)
        elif not a.fuse_rollouts or a.icod:
            # the two rollouts advance in turn, one step each, on two steppers: while the GPU runs one rollout's step the host plans and
            # launches the other's, so the 'sample' rollout's per-step action copy and planning no longer stall the GPU
            r2, r1 = ro.run_interleaved([
                ((env2, env2.reset(batch=batch, features=False)), dict(feedback="sample", train_ml=1.0, rw_seq=rw,
                                                                       sample_draws=rng.uniform(size=(a.max_action_len, a.batch)))),
                ((env, obs), dict(feedback="teacher", train_ml=0.2, rw_seq=rw, ahead=ahead))])
            if a.plan_ahead == "thread":
                # the NEXT batch's teacher-forced rollout needs nothing from the model: its step plans are built on a helper thread while the GPU
                # runs this iteration's backward (host/nav_rollout.PlanAhead; the reference's PrefetchLoader prepares its next batch the same way)
                pipe["obs"] = env.reset(features=False)
                pipe["ahead"] = ro.plan_ahead(env, pipe["obs"])
        else:
            # the two rollouts are independent per episode: one batch of 2B episodes with per-episode feedback and loss weight, text
            # encoded once (tests/test_rollout_gpu.py: same logits, summed loss and gradients as the separate rollouts)
            B = a.batch
            obs = env.reset(batch=batch + batch, features=False)
            draws = np.concatenate([np.zeros((a.max_action_len, B)), rng.uniform(size=(a.max_action_len, B))], 1)
            r1 = ro.run(env, obs, feedback=["teacher"] * B + ["sample"] * B, train_ml=[0.2] * B + [1.0] * B, sample_draws=draws, text_copies=2)
            r2 = {"loss": 0.0, "decisions": 0}
        (r1["loss"] + r2["loss"]).backward(retain_graph=a.icod)       # agent_base.py:260-263
        if a.icod:
            (r1["t_loss"] + r2["t_loss"]).backward()                  # agent_base.py:268-269
            if a.optimizer == "flat":
                t_opt.step(max_norm=40.0)
            else:
                torch.nn.utils.clip_grad_norm_(teacher.parameters(), 40.0)
                t_opt.step()
        if a.optimizer == "flat":
            opt.step(max_norm=40.0)                                  # clip 40 (agent_base.py:273) + AdamW in two launches
        else:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 40.0)     # agent_base.py:273
            opt.step()
        if a.plan_ahead == "inline" and planned:
            # everything above is launched, nothing has been waited for: the host builds the next batch's teacher-forced plans now, under the
            # GPU's backward + optimizer (the next iteration's first host wait -- the language call's length check -- comes after)
            pipe["obs"] = env.reset(features=False)
            pipe["ahead"] = ro.plan_ahead(env, pipe["obs"], thread=False)
        return r1["decisions"] + r2["decisions"]

    for _ in range(a.warmup):
        iteration()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dec = 0
    counters["rollout_steps"] = 0
    for _ in range(a.steps):
        dec += iteration()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    graphs_rep = ro.graph_report() if ro.graphs else None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        td = torch.tensor([dec], device=dev, dtype=torch.float64)
        dist.all_reduce(td)
        dec = float(td.item())

    roof = None
    if not a.no_profile:
        O.FLOPS.update(total=0.0, gemm=0.0, linear_ln=0.0, attn=0.0, enabled=True)
        L.PROFILE.update(on=True, events=[])
        t1 = time.perf_counter()
        pdec = iteration()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t1
        L.PROFILE["on"] = False
        O.FLOPS["enabled"] = False
        by = {}
        for name, layout, e0, e1 in L.PROFILE["events"]:
            k = name if layout < 0 else f"{name}[{['NT', 'NN', 'TN'][layout]}]"
            t, c = by.get(k, (0.0, 0))
            by[k] = (t + e0.elapsed_time(e1), c + 1)
        gemm_ms = max(sum(t for k, (t, c) in by.items() if k.startswith("magic_gemm")), 1e-9)
        gemm_n = sum(c for k, (t, c) in by.items() if k.startswith("magic_gemm"))
        all_ms = sum(t for t, c in by.values())
        ach = O.FLOPS["gemm"] / (gemm_ms * 1e-3) / 1e12         # GEMM family: its own FLOPs over its own launch durations
        mfma_ms = max(sum(t for k, (t, c) in by.items() if any(x in k for x in ("magic_gemm", "magic_linear_ln", "magic_attn_"))), 1e-9)
        roof = {"bound": "mfma", "kernel": "gemm_kernel <bf16, NT|NN|TN> (every dense contraction of the iteration)", "achieved": round(ach, 2),
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 5), "traffic": None,
                "detail": {"algorithmic_gflop_per_iteration": round(O.FLOPS["gemm"] / 1e9, 1), "gemm_launches": gemm_n,
                           "all_dense_contraction_kernels": {"algorithmic_gflop": round(O.FLOPS["total"] / 1e9, 1), "ms": round(mfma_ms, 2),
                                                             "achieved_tflops": round(O.FLOPS["total"] / (mfma_ms * 1e-3) / 1e12, 2)},
                           "avg_gemm_launch_us": round(gemm_ms / max(gemm_n, 1) * 1e3, 2), "gemm_ms": round(gemm_ms, 2),
                           "all_kernels_ms": round(all_ms, 2), "launches": sum(c for t, c in by.values()),
                           "instrumented_iteration_wall_ms": round(wall * 1e3, 1), "decisions": pdec,
                           "instrumented_pass": "eager per-op launches, a HIP event pair around each: the timed iterations replay captured step graphs in which the "
                                                "two cross-modal encoders' launches are PAIRED (one grouped kernel per twin pair), so they dispatch fewer kernels than "
                                                "`launches` -- profiles/r05_nav_launches.txt holds the profiler's count per steady iteration",
                           "top_kernels_share": {k: round(t / all_ms, 4) for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:8]}}}

    host_loop = None
    if rank == 0 and not a.no_host_loop and a.mode == "train" and not a.icod:
        host_loop = host_loop_rate(a, model, dev)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.mode == "train" and not a.icod:
        cpu = cpu_baseline(a, cfg)
    if rank == 0:
        print(json.dumps({
            "metric": "trajectory-steps/sec (whole node), navigator step loop, MAGIC-L fine-tune", "value": round(dec / dt, 2),
            "unit": "trajectory-steps/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"plan_ahead": (a.plan_ahead if a.mode != "eval" else None),
                       "rollouts": ("one batch of 2B episodes" if (a.fuse_rollouts and not a.icod) else "sequential" if a.sequential_rollouts else "interleaved step by step"),
                       "workload": f"navigator loop (agent_base.py:215-296 iteration = teacher-forced + DAgger 'sample' rollout, backward, clip 40, "
                                   f"{'torch.optim.AdamW arithmetic on the flat buffers' if a.optimizer == 'flat' else 'torch AdamW'}), VLNBert H={a.hidden} 6+2+3 layers, dropout 0.1, expert ndtw, instructions U{{{a.instr_min}..{a.instr_max}}} tokens, "
                                   f"paths {a.hops_min}..{a.hops_max} hops, max_action_len {a.max_action_len}",
                       "per_gpu_batch": a.batch, "global_batch": a.batch * world, "views": 36, "feat_dim": 768, "parallelism": f"dp{world}",
                       "decisions_per_iteration": round(dec / a.steps / world, 1)},
            "mode": a.mode + ("/icod" if a.icod else "") + ("/graph" if (a.graph and a.mode == "eval") else ""), "teacher_hidden": a.teacher_hidden if a.icod else None, "ms_per_rollout_step": (round(dt / max(counters["rollout_steps"], 1) * 1e3, 3) if a.mode == "eval" else None),
            "step_graphs": graphs_rep, "roofline": roof, "cpu_baseline": cpu, "host_loop": host_loop}))
    if world > 1:
        dist.destroy_process_group()


def host_loop_rate(a, model, dev):
    """teacher-forced rollout + backward: index plans vs the reference-style per-sample loop over the compat GraphMap, same model"""
    from magic_amd.host import graph_map as GM
    from oracle import rollout_ref as R                      # the reference-style loop lives with the checker; timed here as a baseline
    out = {}
    env = make_env(a, 777)
    table = torch.from_numpy(env.feature_table).to(dev).to(model.net.dtype)
    ro = NavRollout(model, table, max_action_len=a.max_action_len, expert_policy="ndtw")
    saved = R.RefGraphMap
    R.RefGraphMap = GM.GraphMap

    def call(mode, b):
        return model(mode, {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()})
    orig = R.F.cross_entropy
    R.F.cross_entropy = lambda lg, tg, **kw: orig(lg.float(), tg.to(lg.device), **kw)
    try:
        for name in ("index_plans", "per_sample_loop"):
            times, decs = [], []
            for it in range(6):
                model.store.zero_grad()
                obs = env.reset(features=(name == "per_sample_loop"))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if name == "index_plans":
                    r = ro.run(env, obs, feedback="teacher", train_ml=0.2)
                    d = r["decisions"]
                else:
                    r = R.rollout(env, call, obs, feedback="teacher", train_ml=0.2, max_action_len=a.max_action_len)
                    d = sum(int((s["targets"] != -100).sum()) for s in r["steps"])
                r["loss"].backward()
                torch.cuda.synchronize()
                if it > 0:      # the first one warms the allocator; a new batch shape can still stall on a device allocation: median
                    times.append(time.perf_counter() - t0)
                    decs.append(d)
            mid = sorted(range(len(times)), key=lambda i: times[i] / decs[i])[len(times) // 2]
            out[name] = {"trajectory_steps_per_sec": round(decs[mid] / times[mid], 1), "ms_per_rollout": round(times[mid] * 1e3, 1),
                         "sample": "median of 5 teacher-forced rollouts + backward"}
    finally:
        R.RefGraphMap, R.F.cross_entropy = saved, orig
    return out


if __name__ == "__main__":
    main()
