"""what ONE captured navigator step instance launches (MAGIC-L width, RxR-length instructions): the C-ABI calls issued while a step's forward /
backward graph is being captured, in order, with the pairs lib.lockstep formed.  `python profiles/micro/nav_capture_trace.py`"""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import lib as L  # noqa: E402
from magic_amd.host.config import make_config  # noqa: E402
from magic_amd.host.model_nav import VLNBert  # noqa: E402
from magic_amd.host.nav_rollout import NavRollout  # noqa: E402
from magic_amd.host.synth_env import SynthNavEnv  # noqa: E402

dev = torch.device("cuda", 0)
B, T = 16, 6
cfg = make_config(768, role="teacher", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
model = VLNBert(None, role="student", config=cfg, device=dev, compute_dtype=torch.bfloat16, seed=0)
model.train()
env = SynthNavEnv(batch_size=B, n_scans=2, nodes_per_scan=40, seed=3, instr_len=(300, 480), vocab=(3, 30000), path_hops=(3, 5))
table = torch.from_numpy(env.feature_table).to(dev).to(torch.bfloat16)
ro = NavRollout(model, table, max_action_len=T, expert_policy="ndtw", graphs=True, Lcap=480)
log = []
raw, submit = L._raw_call, L.Lockstep.submit


def traced_raw(name, args):
    if torch.cuda.is_current_stream_capturing():
        log.append(("solo", name))
    return raw(name, args)


def traced_submit(self, idx, name, args):
    if self.pending[1 - idx] is not None:
        log.append(("pair", self.pending[1 - idx][0] + " + " + name))
        L._raw_call = raw                    # (the pair is recorded through _fn directly; a partner finishing alone goes through _raw_call)
        try:
            return submit(self, idx, name, args)
        finally:
            L._raw_call = traced_raw
    return submit(self, idx, name, args)


L._raw_call, L.Lockstep.submit = traced_raw, traced_submit
rng = np.random.default_rng(0)
for it in range(3):
    model.store.zero_grad()
    batch = [env._draw_episode() for _ in range(B)]
    env_t, env_s = env, SynthNavEnv(batch_size=B, n_scans=2, nodes_per_scan=40, seed=3, instr_len=(300, 480), vocab=(3, 30000), path_hops=(3, 5))
    mark = len(log)
    r_s, r_t = ro.run_interleaved([
        ((env_s, env_s.reset(batch=batch, features=False)), dict(feedback="sample", train_ml=1.0, sample_draws=rng.uniform(size=(T, B)))),
        ((env_t, env_t.reset(batch=batch, features=False)), dict(feedback="teacher", train_ml=0.2))])
    fwd_end = len(log)
    (r_t["loss"] + r_s["loss"]).backward()
    torch.cuda.synchronize()
    print(f"iteration {it}: {fwd_end - mark} calls traced under capture in the forward, {len(log) - fwd_end} in the backward; {ro.graph_report()}")
cnt = collections.Counter(log)
print("\n== all captures of the two iterations, by call ==")
for (kind, name), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{n:6d}  {kind:4s}  {name}")
