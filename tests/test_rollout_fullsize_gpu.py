"""Navigator loop at BASELINE config-5 size (MAGIC-L width H = 768, B = 16, RxR-length instructions up to 512 tokens, 8-15 hop paths)
where the fp64 oracle is too slow to be the checker: size-independent properties (-m gpu).
  * the per-episode K/V cache is exact: logits with `VLNBert.text_kv` are bit-identical to per-step projection, gradients agree;
  * batch-order equivariance: permuting the episodes permutes the per-step logits bit-exactly (no cross-sample coupling anywhere
    in the loop -- embedding log, CSR gather, fusion map are all per sample);
  * bf16 (the benchmarked arithmetic) tracks the fp32 engine, which the small-size tests pin to the oracle;
  * an ended episode contributes exactly nothing: its targets are -100 from the step it ends and the loss ignores them.
Plus edge cases of the loop at small size: B = 1, the max_action_len cut-off, a rollout that ends in one step."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import magic_amd  # noqa: F401
from magic_amd.host.config import make_config
from magic_amd.host.model_nav import VLNBert
from magic_amd.host.nav_rollout import NavRollout
from magic_amd.host.synth_env import SynthNavEnv

pytestmark = pytest.mark.gpu
DEV = "cuda"


def big_env(seed, B=16):
    return SynthNavEnv(batch_size=B, n_scans=3, nodes_per_scan=64, seed=seed, instr_len=(100, 512), path_hops=(8, 15))


def model(dtype, H=768, seed=0, drop=0.0):
    cfg = make_config(H, role="teacher", hidden_dropout_prob=drop, attention_probs_dropout_prob=drop)
    m = VLNBert(None, role="student", config=cfg, device=DEV, compute_dtype=dtype, seed=seed)
    m.eval()
    return m


def rollout(m, env, batch=None, cache=True, T=28, dtype=None):
    table = torch.from_numpy(env.feature_table).to(DEV).to(m.net.dtype)
    ro = NavRollout(m, table, max_action_len=T, expert_policy="ndtw", cache_text_kv=cache)
    m.store.zero_grad()
    out = ro.run(env, env.reset(batch=batch, features=False), feedback="teacher", train_ml=0.2, record=True)
    out["loss"].backward()
    torch.cuda.synchronize()
    return out, m.store.grad.clone()


def test_text_kv_cache_is_exact_at_full_size():
    m = model(torch.bfloat16)
    env = big_env(5)
    batch = [env._draw_episode() for _ in range(16)]
    a, ga = rollout(m, env, batch, cache=True)
    b, gb = rollout(m, env, batch, cache=False)
    assert len(a["steps"]) == len(b["steps"]) >= 9
    for x, y in zip(a["steps"], b["steps"]):
        assert torch.equal(x["logits"], y["logits"])              # same kernels on the same K/V values
    assert float(a["loss"].detach()) == float(b["loss"].detach())
    assert F.cosine_similarity(ga, gb, dim=0).item() > 0.9999     # the projection's weight gradient: once on the summed dK/dV vs per step


def test_episode_permutation_equivariance_and_ended_episodes():
    m = model(torch.bfloat16, seed=1)
    env = big_env(7)
    batch = [env._draw_episode() for _ in range(16)]
    perm = np.random.default_rng(0).permutation(16)
    a, _ = rollout(m, env, batch)
    b, _ = rollout(m, env, [batch[i] for i in perm])
    assert len(a["steps"]) == len(b["steps"])
    for x, y in zip(a["steps"], b["steps"]):
        K = min(x["logits"].shape[1], y["logits"].shape[1])
        assert torch.equal(x["logits"][perm][:, :K], y["logits"][:, :K])
        assert torch.equal(x["targets"][perm], y["targets"])
    # an episode that has ended only carries ignore_index targets afterwards
    lens = [len(ep["path"]) for ep in batch]
    for t, st in enumerate(a["steps"]):
        for i in range(16):
            assert (int(st["targets"][i]) == -100) == (t >= lens[i])
    assert a["decisions"] == sum(lens)


def test_bf16_tracks_fp32_at_full_size():
    env = big_env(9, B=8)
    batch = [env._draw_episode() for _ in range(8)]
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        m = model(dt, seed=2)
        res[dt] = rollout(m, env, batch)
        del m
    (o32, g32), (o16, g16) = res[torch.float32], res[torch.bfloat16]
    assert torch.isfinite(g16).all() and torch.isfinite(g32).all()
    assert abs(float(o32["loss"].detach()) - float(o16["loss"].detach())) <= 2e-2 * abs(float(o32["loss"].detach())) + 1e-4
    assert F.cosine_similarity(g32, g16, dim=0).item() > 0.98
    for x, y in zip(o32["steps"], o16["steps"]):
        assert torch.equal(torch.isinf(x["logits"]), torch.isinf(y["logits"]))
        d = (torch.nan_to_num(x["logits"], neginf=0) - torch.nan_to_num(y["logits"], neginf=0)).abs().max()
        assert d < 0.15, float(d)


def test_edge_cases_single_episode_cutoff_and_one_step():
    m = model(torch.float32, H=128, seed=3)
    env = SynthNavEnv(batch_size=1, n_scans=1, nodes_per_scan=20, seed=4, instr_len=(5, 9), path_hops=(3, 5), vocab=(3, 290))
    table = torch.from_numpy(env.feature_table).to(DEV)
    ep = env._draw_episode()
    out = NavRollout(m, table, max_action_len=15).run(env, env.reset(batch=[ep], features=False), record=True)       # B = 1
    assert len(out["steps"]) == len(ep["path"]) and out["decisions"] == len(ep["path"])
    assert sum(out["traj"][0]["path"], []) [:len(ep["path"])] == ep["path"]
    out["loss"].backward()
    # cut-off: fewer steps allowed than the path needs -> the last allowed step ends the episode
    cut = NavRollout(m, table, max_action_len=2).run(env, env.reset(batch=[ep], features=False), record=True, grad=False)
    assert len(cut["steps"]) == 2 and cut["steps"][-1]["actions"] == [None]
    # a path of one viewpoint: the first decision is "stop"
    one = dict(ep, path=ep["path"][:1])
    o1 = NavRollout(m, table, max_action_len=15).run(env, env.reset(batch=[one], features=False), record=True, grad=False)
    assert len(o1["steps"]) == 1 and int(o1["steps"][0]["targets"][0]) == 0 and o1["traj"][0]["path"] == [[one["path"][0]]]
