"""cProfile of one navigator fine-tuning iteration on the host (the loop is host-bound): where the Python time goes."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host.config import make_config  # noqa: E402
from magic_amd.host.model_nav import VLNBert  # noqa: E402
from magic_amd.host.nav_rollout import NavRollout  # noqa: E402
from magic_amd.host.synth_env import SynthNavEnv  # noqa: E402

dev = torch.device("cuda", 0)
cfg = make_config(768, role="teacher", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
model = VLNBert(None, role="student", config=cfg, device=dev, compute_dtype=torch.bfloat16, seed=0)
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-5)
mk = lambda: SynthNavEnv(batch_size=16, n_scans=6, nodes_per_scan=64, seed=1234, instr_len=(100, 512), path_hops=(8, 15))
env, env2 = mk(), mk()
table = torch.from_numpy(env.feature_table).to(dev).to(torch.bfloat16)
GRAPHS = "--graphs" in sys.argv
ro = NavRollout(model, table, max_action_len=28, expert_policy="ndtw", graphs=GRAPHS, Lcap=512)
rng = np.random.default_rng(0)


def iteration():
    opt.zero_grad()
    obs = env.reset(features=False)
    batch = env.batch
    r2, r1 = ro.run_interleaved([((env2, env2.reset(batch=batch, features=False)), dict(feedback="sample", train_ml=1.0, sample_draws=rng.uniform(size=(28, 16)))),
                                 ((env, obs), dict(feedback="teacher", train_ml=0.2))])
    (r1["loss"] + r2["loss"]).backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 40.0)
    opt.step()


for _ in range(5 if GRAPHS else 2):
    iteration()
torch.cuda.synchronize()
cProfile.run("iteration(); torch.cuda.synchronize()", "/tmp/nav.prof")
pstats.Stats("/tmp/nav.prof").sort_stats("tottime").print_stats(40)
pstats.Stats("/tmp/nav.prof").sort_stats("cumtime").print_stats(70)
