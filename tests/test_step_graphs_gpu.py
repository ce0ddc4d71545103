"""Captured step instances of the navigator TRAINING loop (host/step_graphs.py; -m gpu): the panorama / navigation segments of every
step replayed as HIP graphs on per-step instances must reproduce the eager index-plan loop, which tests/test_rollout_gpu.py pins to the
fp64 oracle under the reference-style loop -- per-step logits and actions, trajectories, losses, every parameter gradient; with dropout
on, the backward graphs must regenerate the masks of their OWN forward replays; instances are reused across iterations."""
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
KW = dict(vocab_size=300, num_l_layers=2, num_x_layers=1, num_pano_layers=1)


def _env(seed, B=4):
    return SynthNavEnv(batch_size=B, n_scans=2, nodes_per_scan=30, seed=seed, instr_len=(6, 14), vocab=(3, 290), path_hops=(2, 4))


def _model(H, role="student", seed=0, drop=0.0, dtype=torch.float32, args=None, **kw):
    cfg = make_config(H, role=role, hidden_dropout_prob=drop, attention_probs_dropout_prob=drop, **dict(KW, **kw))
    m = VLNBert(args, role=role, config=cfg, device=DEV, compute_dtype=dtype, seed=seed)
    m.train()
    return m


def _iteration(ro, model, env, env2, batch, draws, teacher=None, rw=None):
    model.store.zero_grad()
    if teacher is not None:
        teacher.store.zero_grad()
    r2, r1 = ro.run_interleaved([
        ((env2, env2.reset(batch=batch, features=False)), dict(feedback="sample", train_ml=1.0, sample_draws=draws, record=True, rw_seq=rw)),
        ((env, env.reset(batch=batch, features=False)), dict(feedback="teacher", train_ml=0.2, record=True, rw_seq=rw))])
    (r1["loss"] + r2["loss"]).backward(retain_graph=teacher is not None and ro.train_teacher)
    if teacher is not None and ro.train_teacher:
        (r1["t_loss"] + r2["t_loss"]).backward()
    torch.cuda.synchronize()
    return r1, r2, model.store.grad.clone(), (teacher.store.grad.clone() if teacher is not None and ro.train_teacher else None)


def _same_steps(a, b, tol):
    assert len(a["steps"]) == len(b["steps"])
    for t, (x, y) in enumerate(zip(a["steps"], b["steps"])):
        K = y["logits"].shape[1]                                   # the graph path pads the map tokens to a bucket: compare the eager extent
        la, lb = x["logits"][:, :K], y["logits"]
        assert torch.equal(torch.isinf(la), torch.isinf(lb)), t
        assert torch.isinf(x["logits"][:, K:]).all(), t            # padded map slots can never be chosen
        assert (torch.nan_to_num(la, neginf=0) - torch.nan_to_num(lb, neginf=0)).abs().max().item() <= tol, t
        assert x["actions"] == y["actions"], t
    assert [p["path"] for p in a["traj"]] == [p["path"] for p in b["traj"]]


def test_graph_instanced_iteration_equals_the_eager_iteration_fp32():
    m = _model(128, seed=3)
    B, T = 4, 6
    env_a, env_b = _env(11, B), _env(11, B)
    table = torch.from_numpy(env_a.feature_table).to(DEV)
    eager = NavRollout(m, table, max_action_len=T)
    graph = NavRollout(m, table, max_action_len=T, graphs=True, Lcap=16)
    rng = np.random.default_rng(0)
    for it in range(4):                                            # iteration 0 of a key runs eagerly (host-side lazy init), then instances are captured, then reused
        batch = [env_a._draw_episode() for _ in range(B)]
        draws = rng.uniform(size=(T, B))
        e1, e2, ge, _ = _iteration(eager, m, env_a, env_b, batch, draws)
        g1, g2, gg, _ = _iteration(graph, m, env_a, env_b, batch, draws)
        _same_steps(g1, e1, 1e-5)
        _same_steps(g2, e2, 1e-5)
        for a, b in ((g1, e1), (g2, e2)):
            assert abs(float(a["loss"].detach()) - float(b["loss"].detach())) <= 1e-5 * abs(float(b["loss"].detach()))
        assert (gg - ge).abs().max().item() <= 2e-5 * ge.abs().max().item(), it
        del e1, e2, g1, g2
    rep = graph.graph_report()["student"]
    assert rep["instances"] >= 4 and rep["captures"] >= 2 * rep["instances"]
    n0 = rep["captures"]
    batch = [env_a._draw_episode() for _ in range(B)]
    _iteration(graph, m, env_a, env_b, batch, rng.uniform(size=(T, B)))
    assert graph.graph_report()["student"]["captures"] <= n0 + 8           # steady state: (almost) everything replays instances captured before


def test_backward_graphs_regenerate_the_masks_of_their_own_forward():
    """dropout 0.1: the gradient of a graph-instanced iteration must equal the eager gradient computed with the SAME masks.  The seeds are
    drawn on the device per instance, so the check is done through linearity instead: with dropout on, loss.backward() of the graph path
    must satisfy the directional-derivative identity  dL(w + eps d) ~ eps <grad, d>  only if forward and backward used the same masks --
    replay the forward graphs at w + eps d with the instances' seed words held fixed."""
    m = _model(128, seed=5, drop=0.1)
    B, T = 4, 5
    env_a, env_b = _env(17, B), _env(17, B)
    table = torch.from_numpy(env_a.feature_table).to(DEV)
    ro = NavRollout(m, table, max_action_len=T, graphs=True, Lcap=16)
    batch = [env_a._draw_episode() for _ in range(B)]
    for _ in range(2):                                             # warm the keys, capture the instances
        m.store.zero_grad()
        r = ro.run(env_a, env_a.reset(batch=batch, features=False), feedback="teacher", train_ml=1.0)
        r["loss"].backward()
        del r
    sg = ro.step_graphs(m, B)
    insts = [i for lst in sg.pools.values() for i in lst]
    assert insts
    # freeze the seed draw: the forward graphs redraw their seed from a device counter -- hold the counter fixed between the two evaluations
    m.store.zero_grad()
    cnt0 = sg.rng_counter.clone()
    torch.manual_seed(1234)                                        # (the eager language call draws its seed from torch's generator)
    r = ro.run(env_a, env_a.reset(batch=batch, features=False), feedback="teacher", train_ml=1.0)
    l0 = float(r["loss"].detach())
    r["loss"].backward()
    torch.cuda.synchronize()
    g = m.store.grad.clone()
    del r
    # direction: random, plus a component along the gradient so the derivative is well above the finite difference's noise floor (the fp32 loss
    # moves by whole ulps: one ulp of 9.0 over 2 eps is ~0.05 here -- a direction nearly orthogonal to the gradient made this check flaky)
    d = torch.randn(g.shape, device=g.device, generator=torch.Generator(g.device).manual_seed(7))
    d *= (g.abs() > 0)                                             # (text encoder parameters: their masks come from the eager language call)
    d += g * (d.norm() / g.norm())
    names = [n for n, _ in m.named_parameters() if "lang_encoder" in n or n.startswith("vln_bert.embeddings")]
    for n in names:
        off, cnt, _ = m.store.offsets[n]
        d[off:off + cnt] = 0
    eps = 1e-2 / d.norm().item()
    vals = []
    for sgn in (+1, -1):
        with torch.no_grad():
            m.store.flat.add_(d, alpha=sgn * eps)
        m.store.shadow_clean = False
        sg.rng_counter.copy_(cnt0)
        torch.manual_seed(1234)
        with torch.enable_grad():
            r = ro.run(env_a, env_a.reset(batch=batch, features=False), feedback="teacher", train_ml=1.0)
        vals.append(float(r["loss"].detach()))
        del r
        with torch.no_grad():
            m.store.flat.add_(d, alpha=-sgn * eps)
    fd = (vals[0] - vals[1]) / (2 * eps)
    an = float((g * d).sum())
    assert abs(fd - an) <= 0.05 * max(abs(an), 1e-3) + 1e-3, (fd, an, l0)


def test_icod_cotraining_on_instances_equals_eager():
    from types import SimpleNamespace
    t = _model(256, role="teacher", seed=1, args=SimpleNamespace(train_kdl_teacher=True, teacher_hidden_size=256))
    s = _model(128, seed=2, teacher_hidden_size=256)
    B, T = 4, 5
    env_a, env_b = _env(23, B), _env(23, B)
    table = torch.from_numpy(env_a.feature_table).to(DEV)
    kd = dict(alpha=0.5, t_alpha=0.3, temperature=2.0, decay=0.7)
    eager = NavRollout(s, table, teacher=t, kd=kd, max_action_len=T, train_teacher=True)
    graph = NavRollout(s, table, teacher=t, kd=kd, max_action_len=T, train_teacher=True, graphs=True, Lcap=16)
    rw = (torch.softmax(torch.randn(T, 5, generator=torch.Generator().manual_seed(4)) / 4, -1) * 5).to(DEV)
    rng = np.random.default_rng(1)
    for it in range(3):
        batch = [env_a._draw_episode() for _ in range(B)]
        draws = rng.uniform(size=(T, B))
        e1, e2, ges, get = _iteration(eager, s, env_a, env_b, batch, draws, teacher=t, rw=rw)
        g1, g2, ggs, ggt = _iteration(graph, s, env_a, env_b, batch, draws, teacher=t, rw=rw)
        _same_steps(g1, e1, 1e-5)
        _same_steps(g2, e2, 1e-5)
        for k in ("loss", "t_loss"):
            assert abs(float(g1[k].detach()) - float(e1[k].detach())) <= 2e-5 * abs(float(e1[k].detach())), (it, k)
        assert (ggs - ges).abs().max().item() <= 5e-5 * ges.abs().max().item(), it
        assert (ggt - get).abs().max().item() <= 5e-5 * get.abs().max().item(), it
        del e1, e2, g1, g2
    rep = graph.graph_report()
    assert rep["student"]["instances"] > 0 and rep["teacher"]["instances"] > 0


def test_full_width_bf16_iteration_on_instances_tracks_eager():
    """MAGIC-L width, RxR-length instructions, bf16 (BASELINE config 5's arithmetic): same trajectories, loss within bf16 summation noise,
    gradient cosine -- the graphs replay the eager launches on K-padded shapes."""
    cfg = make_config(768, role="teacher", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = VLNBert(None, role="student", config=cfg, device=DEV, compute_dtype=torch.bfloat16, seed=0)
    m.train()
    B, T = 8, 10
    mk = lambda: SynthNavEnv(batch_size=B, n_scans=3, nodes_per_scan=64, seed=5, instr_len=(100, 300), path_hops=(4, 7))
    env_a, env_b = mk(), mk()
    table = torch.from_numpy(env_a.feature_table).to(DEV).to(torch.bfloat16)
    eager = NavRollout(m, table, max_action_len=T, expert_policy="ndtw")
    graph = NavRollout(m, table, max_action_len=T, expert_policy="ndtw", graphs=True, Lcap=304)
    rng = np.random.default_rng(2)
    for it in range(3):
        batch = [env_a._draw_episode() for _ in range(B)]
        draws = rng.uniform(size=(T, B))
        e1, e2, ge, _ = _iteration(eager, m, env_a, env_b, batch, draws)
        g1, g2, gg, _ = _iteration(graph, m, env_a, env_b, batch, draws)
        # teacher-forced rollout: the trajectory is the expert's whatever the logits; compare logits loosely, losses and gradients
        assert len(g1["steps"]) == len(e1["steps"])
        for x, y in zip(g1["steps"], e1["steps"]):
            K = y["logits"].shape[1]
            assert (torch.nan_to_num(x["logits"][:, :K], neginf=0) - torch.nan_to_num(y["logits"], neginf=0)).abs().max().item() < 5e-2
        assert abs(float(g1["loss"].detach()) - float(e1["loss"].detach())) <= 2e-2 * abs(float(e1["loss"].detach()))
        if [p["path"] for p in g2["traj"]] == [p["path"] for p in e2["traj"]]:      # (a sampled action can flip on a bf16-level tie: then the rollouts differ)
            assert F.cosine_similarity(gg, ge, dim=0).item() > 0.995, it
        del e1, e2, g1, g2
    assert graph.graph_report()["student"]["instances"] > 0
