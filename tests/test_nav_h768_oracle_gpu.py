"""MAGIC-L WIDTH against the fp64 oracle, tensor by tensor (-m gpu): H = 768, 12 heads, I = 3072, instructions long enough
for the key-split attention (128 < L <= 512 keys), at a depth the oracle finishes in seconds.  The HIP side runs the fine-tuning iteration
exactly as bench_nav.py does -- both rollouts interleaved on gradient lanes, every step a captured instance, LayerNorm parameter gradients
through partial rows + column sums, the lean one-row-per-wave LayerNorm backward, the key-split attention forward / backward with in-place
K/V gradient accumulation, the embedding log's in-place gradient rows, all weight gradients in the pass-end concatenated launch -- and is
compared with oracle/rollout_ref.py (the reference-style per-sample loop over oracle/nav_ref.RefVLNBert in fp64) on the same episodes:

  * fp32 storage: every step's action logits (north-star bar: |delta| < 1e-3, argmax identical), the loss, and EVERY parameter gradient
    tensor, each against its own scale;
  * bf16 storage (the benchmarked arithmetic): logits and loss within bf16 tolerance, every gradient tensor's cosine with the oracle's.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import magic_amd  # noqa: F401
from magic_amd.host.config import make_config
from magic_amd.host.model_nav import VLNBert
from magic_amd.host.nav_rollout import NavRollout
from magic_amd.host.synth_env import SynthNavEnv
from oracle import rollout_ref as R
from oracle.nav_ref import RefVLNBert

pytestmark = pytest.mark.gpu
DEV = "cuda"
KW = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=400, num_l_layers=2, num_x_layers=2, num_pano_layers=1)
FULL = dict(KW, num_l_layers=6, num_x_layers=3, num_pano_layers=2)        # MAGIC-L's depth (pretrain_src/config/r2r_magic_model_config.json:10-13): one run, slow
B, T, LCAP = 3, 4, 192


def _env(seed):
    return SynthNavEnv(batch_size=B, n_scans=2, nodes_per_scan=30, seed=seed, instr_len=(140, 185), vocab=(3, 390), path_hops=(2, 3))


def _pair(dtype, seed=0, kw=None):
    cfg = make_config(768, role="teacher", **(kw or KW))
    assert cfg.hidden_size == 768 and cfg.num_attention_heads == 12 and cfg.intermediate_size == 3072
    torch.manual_seed(seed)
    o = RefVLNBert(cfg).double().eval()
    with torch.no_grad():
        for n, p in o.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.02)
            if "LayerNorm.weight" in n or "layer_norm.weight" in n:
                p.add_(torch.randn_like(p) * 0.05)
    g = VLNBert(None, role="student", config=cfg, device=DEV, compute_dtype=dtype)
    g.load_state_dict(o.state_dict())
    g.train()                     # (dropout 0: train() only switches the captured training path on)
    return o, g


def _f64(model):
    def call(mode, b):
        return model(mode, {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()})
    return call


def _oracle_iteration(o, seed, batch, draws):
    env_t, env_s = _env(seed), _env(seed)
    for p in o.parameters():
        p.grad = None
    w_t = R.rollout(env_t, _f64(o), env_t.reset(batch=batch), feedback="teacher", train_ml=0.2, max_action_len=T, expert_policy="ndtw")
    w_s = R.rollout(env_s, _f64(o), env_s.reset(batch=batch), feedback="sample", train_ml=1.0, max_action_len=T, sample_draws=draws,
                    expert_policy="ndtw")              # (run_rxr_kdl_valid.sh: --expert_policy ndtw)
    (w_t["loss"] + w_s["loss"]).backward()                # agent_base.py:243-263: two rollouts, ONE backward
    return w_t, w_s


def _engine_iteration(ro, g, seed, batch, draws):
    env_t, env_s = _env(seed), _env(seed)
    g.store.zero_grad()
    r_s, r_t = ro.run_interleaved([
        ((env_s, env_s.reset(batch=batch, features=False)), dict(feedback="sample", train_ml=1.0, sample_draws=draws, record=True)),
        ((env_t, env_t.reset(batch=batch, features=False)), dict(feedback="teacher", train_ml=0.2, record=True))])
    (r_t["loss"] + r_s["loss"]).backward()
    torch.cuda.synchronize()
    return r_t, r_s


def _logits_of(got, want, tol, exact_argmax):
    assert len(got["steps"]) == len(want["steps"])
    worst = 0.0
    for t, (a, b) in enumerate(zip(got["steps"], want["steps"])):
        K = b["logits"].shape[1]
        x, y = a["logits"][:, :K], b["logits"].float()
        assert torch.equal(a["targets"], b["targets"]), t         # the expert's (nDTW) targets: same planner decisions on both sides
        assert torch.equal(torch.isinf(x), torch.isinf(y)), t
        assert torch.isinf(a["logits"][:, K:]).all(), t           # map tokens padded up to the captured bucket can never be chosen
        d = (torch.nan_to_num(x, neginf=0) - torch.nan_to_num(y, neginf=0)).abs().max().item()
        worst = max(worst, d)
        assert d < tol, (t, d)
        if exact_argmax:
            assert torch.equal(x.argmax(1), y.argmax(1)), t
            assert a["actions"] == b["actions"], t
    return worst


def _robust_draws(w_s, draws):
    """the uniform numbers of the 'sample' rollout moved to the MIDDLE of the CDF interval of the action the oracle sampled with them: the oracle's trajectory
    is unchanged by construction, and the engine follows it unless a probability moves by half the chosen action's mass -- a bf16-level tie on an interval
    edge can no longer fork the two rollouts (the test used to skip itself then)"""
    out = np.array(draws, dtype=np.float64, copy=True)
    for t, st in enumerate(w_s["steps"]):
        p = torch.softmax(st["logits"].double(), 1)
        cdf = p.cumsum(1)
        tot = cdf[:, -1]
        u = torch.as_tensor(draws[t], dtype=torch.float64)
        a = (cdf < (u * tot)[:, None]).sum(1).clamp(max=p.shape[1] - 1)
        hi = cdf.gather(1, a[:, None])[:, 0]
        lo = torch.where(a > 0, cdf.gather(1, (a - 1).clamp(min=0)[:, None])[:, 0], torch.zeros_like(hi))
        out[t] = ((lo + hi) / 2 / tot).numpy()
    return out


@pytest.mark.parametrize("depth", ["reduced", "full"])
def test_fp32_iteration_at_magic_l_width_matches_the_oracle_tensor_by_tensor(depth):
    o, g = _pair(torch.float32, kw=FULL if depth == "full" else KW)
    env = _env(11)
    table = torch.from_numpy(env.feature_table).to(DEV)
    ro = NavRollout(g, table, max_action_len=T, expert_policy="ndtw", graphs=True, Lcap=LCAP)
    rng = np.random.default_rng(0)
    for it in range(3):                       # iteration 0 sees the shapes (eager), 1 captures the instances, 2 replays them
        batch = [env._draw_episode() for _ in range(B)]
        draws = rng.uniform(size=(T, B))
        r_t, r_s = _engine_iteration(ro, g, 11, batch, draws)
    w_t, w_s = _oracle_iteration(o, 11, batch, draws)
    rep = ro.graph_report()["student"]
    assert rep["instances"] >= 4 and rep["captures"] >= 2 * rep["instances"], rep
    for got, want in ((r_t, w_t), (r_s, w_s)):
        _logits_of(got, want, 1e-3, exact_argmax=True)                       # the north star's bar
        assert [x["path"] for x in got["traj"]] == [x["path"] for x in want["traj"]]
        assert abs(float(got["loss"].detach()) - float(want["loss"])) <= 2e-4 * abs(float(want["loss"]))
    params = dict(g.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in o.parameters() if p.grad is not None)
    n = 0
    for name, p in o.named_parameters():
        gg = params[name].grad
        if p.grad is None:
            assert gg is None or gg.abs().max().item() == 0.0, name
            continue
        ref = p.grad.float()
        err = (gg.detach().float().cpu() - ref).abs().max().item()
        assert err <= 2e-3 * ref.abs().max().item() + 2e-6 * gmax, (name, err, ref.abs().max().item())
        n += 1
    assert n > 60, n


def test_bf16_iteration_at_magic_l_width_tracks_the_oracle_tensor_by_tensor():
    o, g = _pair(torch.bfloat16)
    env = _env(13)
    table = torch.from_numpy(env.feature_table).to(DEV).to(torch.bfloat16)
    ro = NavRollout(g, table, max_action_len=T, expert_policy="ndtw", graphs=True, Lcap=LCAP)
    rng = np.random.default_rng(1)
    for it in range(3):
        batch = [env._draw_episode() for _ in range(B)]
        draws = rng.uniform(size=(T, B))
        r_t, r_s = _engine_iteration(ro, g, 13, batch, draws)
    # the compared iteration: its draws sit mid-interval of the oracle's sampled actions (see _robust_draws), so the rollouts cannot fork on a tie
    batch = [env._draw_episode() for _ in range(B)]
    u0 = rng.uniform(size=(T, B))
    draws = _robust_draws(_oracle_iteration(o, 13, batch, u0)[1], u0)
    r_t, r_s = _engine_iteration(ro, g, 13, batch, draws)
    w_t, w_s = _oracle_iteration(o, 13, batch, draws)
    assert ro.graph_report()["student"]["instances"] >= 4
    worst_logit = _logits_of(r_t, w_t, 6e-2, exact_argmax=False)                   # teacher forcing: the trajectory is the expert's whatever the logits
    assert abs(float(r_t["loss"].detach()) - float(w_t["loss"])) <= 2e-2 * abs(float(w_t["loss"])), worst_logit
    assert [x["path"] for x in r_s["traj"]] == [x["path"] for x in w_s["traj"]], "the bf16 sample rollout left the oracle's trajectory although its draws sit mid-interval"
    params = dict(g.named_parameters())
    rms = {name: p.grad.double().pow(2).mean().sqrt().item() for name, p in o.named_parameters() if p.grad is not None}
    rms_max = max(rms.values())
    n, low, worst = 0, [], 1.0
    for name, p in o.named_parameters():
        if p.grad is None or rms[name] < 1e-2 * rms_max:      # analytically ~0 gradients (a key bias in front of a softmax ...): cosine is noise there
            continue
        a, b = params[name].grad.detach().double().cpu().reshape(-1), p.grad.reshape(-1)
        c = F.cosine_similarity(a, b, dim=0).item()
        worst = min(worst, c)
        bar = 0.95 if (name.endswith("bias") or "pos" in name) else 0.99       # row-sum parameters cancel: the 16-bit test's cancellation class
        if c < bar:
            low.append((name, round(c, 4)))
        n += 1
    print(f"[bf16 H=768] {n} sizeable gradient tensors, worst cosine {worst:.5f}, worst logit delta {worst_logit:.2e}")
    assert n > 40 and not low, low


def test_paired_and_forked_cross_modal_encoders_give_the_same_iteration():
    """the two forms a captured navigation step can take -- the map and viewpoint encoders' launches PAIRED into grouped kernels (default) or the two
    encoders FORKED onto two branches of the step graph (MAGIC_NAV_PAIR=0) -- with the panorama backwards on their own stream or in the lane's chain,
    are the same arithmetic: fp32 storage, same episodes, same draws -> same trajectories, logits and parameter gradients to summation-order noise"""
    from magic_amd.host import model_nav

    def run(pair, pano_side):
        saved = model_nav.NAV_PAIR
        model_nav.NAV_PAIR = pair
        try:
            _, g = _pair(torch.float32, seed=3)
            env = _env(17)
            table = torch.from_numpy(env.feature_table).to(DEV)
            ro = NavRollout(g, table, max_action_len=T, expert_policy="ndtw", graphs=True, Lcap=LCAP)
            batch = [env._draw_episode() for _ in range(B)]            # (a fresh environment of the same seed: the same episodes in both runs)
            draws = np.random.default_rng(5).uniform(size=(T, B))
            for it in range(3):                # eager first sight, capture, replay -- the same batch each time (dropout 0: identical iterations)
                for sg in ro._sg.values():
                    sg.pano_side = pano_side
                r_t, r_s = _engine_iteration(ro, g, 17, batch, draws)
            rep = ro.graph_report()["student"]
            assert rep["instances"] >= 4, rep
            return r_t, r_s, {n: p.grad.detach().float().cpu().clone() for n, p in g.named_parameters() if p.grad is not None}
        finally:
            model_nav.NAV_PAIR = saved
    t1, s1, g1 = run(True, True)
    t2, s2, g2 = run(False, False)
    for a, b in ((t1, t2), (s1, s2)):
        assert [x["path"] for x in a["traj"]] == [x["path"] for x in b["traj"]]
        assert abs(float(a["loss"].detach()) - float(b["loss"].detach())) <= 1e-5 * abs(float(b["loss"].detach()))
        for sa, sb in zip(a["steps"], b["steps"]):
            x, y = torch.nan_to_num(sa["logits"], neginf=0), torch.nan_to_num(sb["logits"], neginf=0)
            assert (x - y).abs().max().item() < 1e-4
    gmax = max(v.abs().max().item() for v in g2.values())
    assert set(g1) == set(g2) and len(g2) > 60
    for n in g2:
        assert (g1[n] - g2[n]).abs().max().item() <= 1e-4 * g2[n].abs().max().item() + 1e-6 * gmax, n
