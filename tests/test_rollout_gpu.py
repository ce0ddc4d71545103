"""Navigator step loop parity (-m gpu): the index-plan rollout on the HIP engine (host/nav_rollout.py) against the
reference-style per-sample loop (oracle/rollout_ref.py, builders pinned to the reference's own agent methods) driving the fp64 CPU
oracle model, on the same synthetic episodes: per-step action logits, chosen actions (bit-exact), trajectories, episode loss, MAKD
terms and every parameter gradient -- the gradient reaches earlier steps' panorama encoders only through the embedding-log gather."""
import numpy as np
import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host.config import make_config
from magic_amd.host.model_nav import VLNBert
from magic_amd.host.nav_rollout import NavRollout
from magic_amd.host.synth_env import SynthNavEnv
from oracle import rollout_ref as R
from oracle.nav_ref import RefVLNBert

pytestmark = pytest.mark.gpu
DEV = "cuda"
HEADS = ("txt_emb_w", "kdl_img_w", "kdl_avg_img_w", "global_cross_w", "local_cross_w")
KW = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, num_l_layers=2, num_x_layers=1, num_pano_layers=1)


def _env(seed, B=4):
    return SynthNavEnv(batch_size=B, n_scans=2, nodes_per_scan=30, seed=seed, instr_len=(6, 14), vocab=(3, 290), path_hops=(2, 4))


def _pair(cfg, role, seed, args=None):
    torch.manual_seed(seed)
    o = RefVLNBert(cfg).double().eval()
    with torch.no_grad():
        for n, p in o.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.02)
    g = VLNBert(args, role=role, config=cfg, device=DEV, compute_dtype=torch.float32)
    g.load_state_dict(o.state_dict())
    g.eval()
    return o, g


def _f64(model):
    def call(mode, b):
        return model(mode, {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()})
    return call


def close(a, b, name, rtol=2e-4, atol=2e-5):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{name}: max|err| {(a - b).abs().max().item():.3e} (ref {b.abs().max().item():.3e})"


def _check_steps(got, want):
    assert len(got["steps"]) == len(want["steps"])
    for t, (g, w) in enumerate(zip(got["steps"], want["steps"])):
        a, b = g["logits"], w["logits"].float()
        assert torch.equal(torch.isinf(a), torch.isinf(b)), f"step {t}: -inf pattern"
        close(torch.nan_to_num(a, neginf=0), torch.nan_to_num(b, neginf=0), f"step {t} fused logits", 1e-4, 1e-3)
        assert (torch.nan_to_num(a, neginf=0) - torch.nan_to_num(b, neginf=0)).abs().max() < 1e-3      # north-star bar
        assert torch.equal(a.argmax(1), b.argmax(1)), f"step {t}: action argmax must be bit-exact"
        assert torch.equal(g["targets"], w["targets"]) and g["actions"] == w["actions"] and g["vpids"] == w["vpids"]
    assert [x["path"] for x in got["traj"]] == [x["path"] for x in want["traj"]]


def _check_grads(g_model, o_model, floor=40):
    params = dict(g_model.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in o_model.parameters() if p.grad is not None)
    n = 0
    for name, p in o_model.named_parameters():
        g = params[name].grad
        if p.grad is None:
            assert g is None or g.abs().max().item() == 0.0, name
            continue
        close(g, p.grad, f"grad {name}", 2e-3, 1e-3 * p.grad.abs().max().item() + 2e-6 * gmax)
        n += 1
    assert n > floor


@pytest.mark.parametrize("seed,cache_kv", [(3, True), (7, False)])
def test_teacher_forced_rollout_matches_oracle(seed, cache_kv):
    """cache_kv: the instruction's cross-attention K/V projected once per episode (VLNBert.text_kv) vs at every step"""
    cfg = make_config(128, role="student", **KW)
    o_s, g_s = _pair(cfg, "student", 0)
    env_a, env_b = _env(seed), _env(seed)
    want = R.rollout(env_a, _f64(o_s), env_a.reset(), feedback="teacher", train_ml=0.2, max_action_len=6)
    want["loss"].backward()
    table = torch.from_numpy(env_b.feature_table).to(DEV)
    ro = NavRollout(g_s, table, max_action_len=6, cache_text_kv=cache_kv)
    g_s.store.zero_grad()
    got = ro.run(env_b, env_b.reset(features=False), feedback="teacher", train_ml=0.2, record=True)
    _check_steps(got, want)
    close(got["loss"], want["loss"], "episode loss", 2e-4, 1e-6)
    got["loss"].backward()
    torch.cuda.synchronize()
    _check_grads(g_s, o_s)
    assert got["decisions"] == sum(int((s["targets"] != -100).sum()) for s in want["steps"])


def test_argmax_and_sample_rollouts_match_oracle():
    cfg = make_config(128, role="student", **KW)
    o_s, g_s = _pair(cfg, "student", 1)
    table = None
    for feedback, seed in (("argmax", 5), ("sample", 9)):
        env_a, env_b = _env(seed, B=5), _env(seed, B=5)
        draws = np.random.default_rng(seed).uniform(size=(7, 5))
        with torch.no_grad():
            want = R.rollout(env_a, _f64(o_s), env_a.reset(), feedback=feedback, train_ml=1.0, max_action_len=7, sample_draws=draws)
        table = torch.from_numpy(env_b.feature_table).to(DEV)
        ro = NavRollout(g_s, table, max_action_len=7)
        got = ro.run(env_b, env_b.reset(features=False), feedback=feedback, train_ml=1.0, sample_draws=draws, grad=False, record=True)
        _check_steps(got, want)
        close(got["loss"], want["loss"], f"{feedback} loss", 2e-4, 1e-6)


def test_makd_rollout_matches_oracle():
    """teacher (H=256) + student (H=128): MAKD t2s with MKRW weights and MKTD sample weights inside the loop (agent.py:1007-1024)"""
    tcfg, scfg = make_config(256, role="teacher", **KW), make_config(128, role="student", teacher_hidden_size=256, **KW)
    o_t, g_t = _pair(tcfg, "teacher", 0)
    o_s, g_s = _pair(scfg, "student", 1)
    T = 5
    rw = (torch.softmax(torch.randn(T, 5, generator=torch.Generator().manual_seed(4)) / 4, -1) * 5)
    kd = dict(alpha=0.5, temperature=2.0, decay=0.7)
    env_a, env_b = _env(13), _env(13)
    heads = {n: getattr(o_s.vln_bert, n) for n in HEADS}
    want = R.rollout(env_a, _f64(o_s), env_a.reset(), feedback="teacher", train_ml=0.2, max_action_len=T, teacher=_f64(o_t),
                     kd=dict(kd, heads=heads), rw_seq=rw.double())
    want["loss"].backward()
    table = torch.from_numpy(env_b.feature_table).to(DEV)
    ro = NavRollout(g_s, table, teacher=g_t, kd=kd, max_action_len=T)
    g_s.store.zero_grad()
    got = ro.run(env_b, env_b.reset(features=False), feedback="teacher", train_ml=0.2, rw_seq=rw.to(DEV), record=True)
    _check_steps(got, want)
    for k, v in want["kdl_terms"].items():
        close(got["kdl_terms"][k], v, f"kd {k}", 3e-4, 1e-6)
    close(got["loss"], want["loss"], "episode loss", 2e-4, 1e-6)
    got["loss"].backward()
    torch.cuda.synchronize()
    _check_grads(g_s, o_s)


def test_icod_cotraining_rollout_matches_oracle():
    """args.train_kdl_teacher: the teacher runs with gradients and is distilled FROM the student in the reverse ('s2t', mean-reduced)
    direction while the student is distilled from it (agent.py:1013-1026,1138-1149; agent_base.py:260-269 backpropagates both losses)"""
    from types import SimpleNamespace
    tcfg, scfg = make_config(256, role="teacher", **KW), make_config(128, role="student", teacher_hidden_size=256, **KW)
    o_t, g_t = _pair(tcfg, "teacher", 3, args=SimpleNamespace(train_kdl_teacher=True, teacher_hidden_size=256))
    o_s, g_s = _pair(scfg, "student", 4)
    T = 4
    rw = (torch.softmax(torch.randn(T, 5, generator=torch.Generator().manual_seed(8)) / 4, -1) * 5)
    kd = dict(alpha=0.5, t_alpha=0.3, temperature=2.0, decay=0.7)
    env_a, env_b = _env(19), _env(19)
    heads = {n: getattr(o_s.vln_bert, n) for n in HEADS}
    want = R.rollout(env_a, _f64(o_s), env_a.reset(), feedback="teacher", train_ml=0.2, max_action_len=T, teacher=_f64(o_t),
                     kd=dict(kd, heads=heads), rw_seq=rw.double(), train_teacher=True)
    want["loss"].backward(retain_graph=True)
    want["t_loss"].backward()
    table = torch.from_numpy(env_b.feature_table).to(DEV)
    ro = NavRollout(g_s, table, teacher=g_t, kd=kd, max_action_len=T, train_teacher=True)
    g_s.store.zero_grad()
    g_t.store.zero_grad()
    got = ro.run(env_b, env_b.reset(features=False), feedback="teacher", train_ml=0.2, rw_seq=rw.to(DEV), record=True)
    _check_steps(got, want)
    for k, v in want["t_kdl_terms"].items():
        close(got["t_kdl_terms"][k], v, f"s2t {k}", 3e-4, 1e-7)
    close(got["loss"], want["loss"], "student loss", 2e-4, 1e-6)
    close(got["t_loss"], want["t_loss"], "teacher loss", 2e-4, 1e-6)
    got["loss"].backward(retain_graph=True)
    got["t_loss"].backward()
    torch.cuda.synchronize()
    _check_grads(g_s, o_s)
    _check_grads(g_t, o_t)


def test_two_rollouts_as_one_batch_equal_two_separate_rollouts():
    """An iteration's teacher-forced (ml_weight 0.2) and DAgger 'sample' (weight 1) rollouts on the same episodes, run as ONE batch of
    2B episodes with per-episode feedback and the text encoder / K-V projections computed once: same per-step logits, same summed loss,
    same gradients as the two separate rollouts (agent_base.py:243-263 adds the two losses before backward)."""
    cfg = make_config(128, role="student", **KW)
    _, g_s = _pair(cfg, "student", 5)
    B, T = 4, 6
    env = _env(23, B=B)
    batch = [env._draw_episode() for _ in range(B)]
    draws = np.random.default_rng(1).uniform(size=(T, B))
    table = torch.from_numpy(env.feature_table).to(DEV)
    ro = NavRollout(g_s, table, max_action_len=T)
    g_s.store.zero_grad()
    r1 = ro.run(env, env.reset(batch=batch, features=False), feedback="teacher", train_ml=0.2, record=True)
    r2 = ro.run(env, env.reset(batch=batch, features=False), feedback="sample", train_ml=1.0, sample_draws=draws, record=True)
    (r1["loss"] + r2["loss"]).backward()
    torch.cuda.synchronize()
    want_loss, want_grad = float((r1["loss"] + r2["loss"]).detach()), g_s.store.grad.clone()
    env2 = _env(23, B=2 * B)
    g_s.store.zero_grad()
    draws2 = np.concatenate([np.zeros((T, B)), draws], 1)
    rc = ro.run(env2, env2.reset(batch=batch + batch, features=False), feedback=["teacher"] * B + ["sample"] * B,
                train_ml=[0.2] * B + [1.0] * B, sample_draws=draws2, record=True, text_copies=2)
    rc["loss"].backward()
    torch.cuda.synchronize()
    close(rc["loss"], want_loss, "combined loss", 2e-5, 1e-6)
    assert rc["decisions"] == r1["decisions"] + r2["decisions"]
    for t, st in enumerate(rc["steps"]):
        for half, rr in ((0, r1), (1, r2)):
            if t < len(rr["steps"]):
                a, b = st["logits"][half * B:(half + 1) * B], rr["steps"][t]["logits"]
                K = min(a.shape[1], b.shape[1])
                live = rr["steps"][t]["targets"] != -100
                close(torch.nan_to_num(a[live][:, :K], neginf=0), torch.nan_to_num(b[live][:, :K], neginf=0), f"step {t} half {half}", 1e-5, 1e-5)
                assert st["actions"][half * B:(half + 1) * B] == rr["steps"][t]["actions"]
    gmax = want_grad.abs().max().item()
    assert (g_s.store.grad - want_grad).abs().max().item() <= 2e-4 * gmax
    assert [x["path"] for x in rc["traj"]] == [x["path"] for x in r1["traj"]] + [x["path"] for x in r2["traj"]]


def test_interleaved_rollouts_equal_sequential_rollouts():
    """run_interleaved advances the teacher-forced and the 'sample' rollout step by step in turn (each on its own stepper): identical
    per-step logits, actions, losses and gradients to running them one after the other."""
    cfg = make_config(128, role="student", **KW)
    _, g_s = _pair(cfg, "student", 6)
    B, T = 4, 6
    env_a, env_b = _env(29, B=B), _env(29, B=B)
    batch = [env_a._draw_episode() for _ in range(B)]
    draws = np.random.default_rng(2).uniform(size=(T, B))
    table = torch.from_numpy(env_a.feature_table).to(DEV)
    ro = NavRollout(g_s, table, max_action_len=T)
    g_s.store.zero_grad()
    r1 = ro.run(env_a, env_a.reset(batch=batch, features=False), feedback="teacher", train_ml=0.2, record=True)
    r2 = ro.run(env_a, env_a.reset(batch=batch, features=False), feedback="sample", train_ml=1.0, sample_draws=draws, record=True)
    (r1["loss"] + r2["loss"]).backward()
    torch.cuda.synchronize()
    want = g_s.store.grad.clone()
    g_s.store.zero_grad()
    q1, q2 = ro.run_interleaved([
        ((env_a, env_a.reset(batch=batch, features=False)), dict(feedback="teacher", train_ml=0.2, record=True)),
        ((env_b, env_b.reset(batch=batch, features=False)), dict(feedback="sample", train_ml=1.0, sample_draws=draws, record=True))])
    (q1["loss"] + q2["loss"]).backward()
    torch.cuda.synchronize()
    for a, b in ((q1, r1), (q2, r2)):
        assert float(a["loss"].detach()) == float(b["loss"].detach())
        assert len(a["steps"]) == len(b["steps"])
        for x, y in zip(a["steps"], b["steps"]):
            assert torch.equal(x["logits"], y["logits"]) and x["actions"] == y["actions"]
        assert [x["path"] for x in a["traj"]] == [x["path"] for x in b["traj"]]
    assert (g_s.store.grad - want).abs().max().item() <= 1e-5 * want.abs().max().item()      # fp32 atomics: summation order only


@pytest.mark.parametrize("B", [1, 4])
def test_greedy_navigation_as_one_graph_per_step_matches_the_eager_loop(B):
    """host/nav_graph.GreedyNavigator: the decision step captured once with padded static shapes (37 views, Kmax map tokens, Lmax
    instruction tokens) and replayed -- same actions, trajectories and (on the valid tokens) logits as the eager index-plan loop; the
    graph is captured on the first episode batch and REUSED for the following ones."""
    from magic_amd.host.nav_graph import GreedyNavigator
    cfg = make_config(128, role="student", **KW)
    _, g_s = _pair(cfg, "student", 7)
    env_a, env_b = _env(31, B=B), _env(31, B=B)
    table = torch.from_numpy(env_a.feature_table).to(DEV)
    nav = GreedyNavigator(g_s, table, B, Lmax=16, Kmax=40, Tmax=7)
    ro = NavRollout(g_s, table, max_action_len=7)
    for rep in range(3):
        batch = [env_a._draw_episode() for _ in range(B)]
        want = ro.run(env_a, env_a.reset(batch=batch, features=False), feedback="argmax", grad=False, record=True)
        got = nav.run(env_b, env_b.reset(batch=batch, features=False), record=True)
        assert got["n_steps"] == want["n_steps"] and got["decisions"] == want["decisions"]
        for t, (g, w) in enumerate(zip(got["steps"], want["steps"])):
            K = w["logits"].shape[1]
            a, b = g["logits"][:, :K], w["logits"]
            assert torch.equal(torch.isinf(a), torch.isinf(b)), (rep, t)
            close(torch.nan_to_num(a, neginf=0), torch.nan_to_num(b, neginf=0), f"episode batch {rep} step {t}", 1e-5, 1e-5)
            assert g["actions"] == w["actions"], (rep, t)
        assert [x["path"] for x in got["traj"]] == [x["path"] for x in want["traj"]]
    assert nav.graph is not None


def test_compat_graphmap_drives_the_same_numbers():
    """The reference's unmodified loop shape (per-sample GraphMap.update_node_embed / get_node_embed on device tensors +
    pad_tensors_wgrad) over the product GraphMap gives the same logits as the index-plan path."""
    from magic_amd.host import graph_map as GM
    cfg = make_config(128, role="student", **KW)
    _, g_s = _pair(cfg, "student", 2)
    env_a, env_b = _env(17), _env(17)
    table = torch.from_numpy(env_b.feature_table).to(DEV)
    with torch.no_grad():
        got = NavRollout(g_s, table, max_action_len=5).run(env_b, env_b.reset(features=False), grad=False, record=True)
    saved = R.RefGraphMap, R.pad_rows
    R.RefGraphMap = GM.GraphMap
    try:
        def call(mode, b):
            b = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}
            out = g_s(mode, b)
            return out
        with torch.no_grad():
            want = _rollout_on_device(env_a, call)
    finally:
        R.RefGraphMap, R.pad_rows = saved
    for t, (g, w) in enumerate(zip(got["steps"], want)):
        close(torch.nan_to_num(g["logits"], neginf=0), torch.nan_to_num(w, neginf=0), f"step {t}", 1e-5, 1e-5)


def _rollout_on_device(env, call):
    """thin adapter: the oracle loop with device tensors (targets / masks moved where the loop compares them on the host)"""
    logs = []
    orig_ce = torch.nn.functional.cross_entropy

    def ce(logits, targets, **kw):
        logs.append(logits.detach().float().cpu())
        return orig_ce(logits.float().cpu(), targets, **kw)
    R.F.cross_entropy = ce
    try:
        R.rollout(env, call, env.reset(), feedback="teacher", train_ml=1.0, max_action_len=5)
    finally:
        R.F.cross_entropy = orig_ce
    return logs


def test_plans_built_ahead_of_time_drive_the_same_teacher_forced_rollout():
    """NavRollout.plan_ahead: the step plans of a teacher-forced rollout built on a helper thread before the rollout starts (nothing in them
    depends on the model) -- same per-step logits, targets, actions, trajectories, loss and gradients as planning step by step."""
    cfg = make_config(128, role="student", **KW)
    _, g_s = _pair(cfg, "student", 8)
    B, T = 4, 6
    env_a, env_b = _env(37, B=B), _env(37, B=B)
    batch = [env_a._draw_episode() for _ in range(B)]
    table = torch.from_numpy(env_a.feature_table).to(DEV)
    ro = NavRollout(g_s, table, max_action_len=T, expert_policy="ndtw")
    g_s.store.zero_grad()
    want = ro.run(env_a, env_a.reset(batch=batch, features=False), feedback="teacher", train_ml=0.2, record=True)
    want["loss"].backward()
    torch.cuda.synchronize()
    gw = g_s.store.grad.clone()
    obs = env_b.reset(batch=batch, features=False)
    ahead = ro.plan_ahead(env_b, obs)
    g_s.store.zero_grad()
    got = ro.run(env_b, obs, feedback="teacher", train_ml=0.2, record=True, ahead=ahead)
    got["loss"].backward()
    torch.cuda.synchronize()
    assert float(got["loss"].detach()) == float(want["loss"].detach()) and got["decisions"] == want["decisions"]
    assert len(got["steps"]) == len(want["steps"])
    for x, y in zip(got["steps"], want["steps"]):
        assert torch.equal(x["logits"], y["logits"]) and torch.equal(x["targets"], y["targets"]) and x["actions"] == y["actions"]
    assert [p["path"] for p in got["traj"]] == [p["path"] for p in want["traj"]]
    assert (g_s.store.grad - gw).abs().max().item() <= 1e-5 * gw.abs().max().item()
    with pytest.raises(ValueError):
        ro.run(env_b, env_b.reset(batch=batch, features=False), feedback="sample", sample_draws=np.zeros((T, B)), ahead=ro.plan_ahead(env_b, obs))
