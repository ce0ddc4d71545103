"""CPU-only: time of the navigator's index planner (host/nav_plan.NavPlanner) per decision step at the bench_nav.py workload
(B = 16, 64-node scans, paths of 8-15 hops, up to 28 steps, random 'sample' actions).  python profiles/micro/planner_cpu_profile.py [--prof]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host.nav_plan import NavPlanner  # noqa: E402
from magic_amd.host.synth_env import SynthNavEnv  # noqa: E402


def episode(env, rng, T=28, times=None):
    obs = env.reset(features=False)
    pl = NavPlanner(env, obs, feedback="sample", max_action_len=T, expert_policy="ndtw", train=True)
    pl.language()
    for t in range(T):
        t0 = time.perf_counter()
        plan = pl.begin_pano()
        t1 = time.perf_counter()
        plan.update(pl.begin_nav())
        t2 = time.perf_counter()
        K = plan["K"]
        a = np.zeros(plan["B"], np.int64)
        for b in range(plan["B"]):                                 # a random admissible map node (what a 'sample' draw would give)
            ok = np.flatnonzero(np.asarray(plan["gmap_masks"][b], bool) & ~np.asarray(plan["gmap_visited_masks"][b], bool))
            a[b] = rng.choice(ok) if len(ok) else 0
        t3 = time.perf_counter()
        done = pl.end_step(a)
        t4 = time.perf_counter()
        if times is not None:
            times.append((t1 - t0, t2 - t1, t4 - t3))
        if done:
            break
    return t + 1


env = SynthNavEnv(batch_size=16, n_scans=6, nodes_per_scan=64, seed=1234, instr_len=(100, 512), path_hops=(8, 15))
rng = np.random.default_rng(0)
episode(env, rng)
if "--prof" in sys.argv:
    cProfile.run("episode(env, rng)", "/tmp/planner.prof")
    pstats.Stats("/tmp/planner.prof").sort_stats("tottime").print_stats(28)
else:
    times = []
    n = sum(episode(env, rng, times=times) for _ in range(5))
    tm = np.array(times) * 1e3
    print(f"{n} steps: begin_pano {tm[:, 0].mean():.3f} ms  begin_nav {tm[:, 1].mean():.3f} ms  end_step {tm[:, 2].mean():.3f} ms per step (B=16)")
