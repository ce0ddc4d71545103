"""Round 6: host cost of the navigator step planner alone (no GPU, no model): config-5 shaped episodes (B = 16, paths 8..15 hops, 28 steps, ndtw expert),
a teacher-forced and a 'sample' rollout (the sampled action = the expert's here), per planner section and under cProfile.
  python profiles/micro/navplan_host_profile.py [--profile]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import magic_amd  # noqa: E402,F401
from magic_amd.host.nav_plan import NavPlanner  # noqa: E402
from magic_amd.host.synth_env import SynthNavEnv  # noqa: E402


def rollout(env, fb, sec):
    obs = env.reset(features=False)
    t0 = time.perf_counter()
    pl = NavPlanner(env, obs, feedback=fb, max_action_len=28, expert_policy="ndtw", pad_V=37, k_bucket=16)
    sec["init"] += time.perf_counter() - t0
    steps = 0
    for t in range(28):
        t0 = time.perf_counter()
        p = pl.begin_pano()
        t1 = time.perf_counter()
        p.update(pl.begin_nav())
        t2 = time.perf_counter()
        done = pl.end_step(None if fb == "teacher" else p["targets"].clip(min=0))
        t3 = time.perf_counter()
        sec["begin_pano"] += t1 - t0
        sec["begin_nav"] += t2 - t1
        sec["end_step"] += t3 - t2
        steps += 1
        if done:
            break
    return steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--iters", type=int, default=6)
    a = ap.parse_args()
    env = SynthNavEnv(batch_size=16, n_scans=6, nodes_per_scan=64, seed=0, instr_len=(100, 512), path_hops=(8, 15))
    sec = dict(init=0.0, begin_pano=0.0, begin_nav=0.0, end_step=0.0)
    for _ in range(2):
        rollout(env, "teacher", dict(sec)), rollout(env, "sample", dict(sec))
    pr = cProfile.Profile() if a.profile else None
    steps = 0
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    for _ in range(a.iters):
        steps += rollout(env, "teacher", sec) + rollout(env, "sample", sec)
    if pr:
        pr.disable()
    tot = time.perf_counter() - t0
    print(f"{a.iters} iterations (teacher + sample rollout each), {steps} steps: {tot / a.iters * 1e3:.1f} ms per iteration, {tot / steps * 1e3:.3f} ms per step")
    for k, v in sec.items():
        print(f"  {k:12s} {v / a.iters * 1e3:7.2f} ms per iteration   {v / steps * 1e6:7.1f} us per step")
    if pr:
        pstats.Stats(pr).sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
