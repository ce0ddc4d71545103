"""Navigator-loop host logic on CPU: the index plans of host/nav_plan.py against the reference-style per-sample
loops of oracle/rollout_ref.py (pinned to the reference's own methods by tests/golden/nav_loop.pt, see test_oracle_golden.py),
and the array-backed GraphMap / FloydGraph against the dict-of-dict restatement."""
import numpy as np
import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host.config import make_config
from magic_amd.host.graph_map import FloydGraph, GraphMap, pad_tensors_wgrad
from magic_amd.host.nav_plan import NavPlanner
from magic_amd.host.synth_env import SynthNavEnv
from oracle import rollout_ref as R
from oracle.nav_ref import RefVLNBert


def _env(seed, B=5, **kw):
    return SynthNavEnv(batch_size=B, n_scans=2, nodes_per_scan=30, seed=seed, instr_len=(6, 14), **kw)


def test_floyd_matches_dict_version():
    rng = np.random.default_rng(0)
    a, b = FloydGraph(cap=2), R.RefFloyd()
    names = [f"n{i}" for i in range(14)]
    for step in range(40):
        x, y = rng.choice(len(names), 2, replace=False)
        d = float(rng.uniform(0.5, 6))
        a.add_edge(names[x], names[y], d)
        b.add_edge(names[x], names[y], d)
        if step % 3 == 2:
            k = names[int(rng.integers(len(names)))]
            if k in b.dis:
                a.update(k)
                b.update(k)
        for u in names:
            for v in names:
                if u in b.dis and v in b.dis:
                    assert a.distance(u, v) == b.distance(u, v)
                    if a.distance(u, v) < 9e7:
                        assert a.path(u, v) == b.path(u, v)
            assert a.visited(u) == b.visited(u)


def test_graphmap_matches_dict_version():
    env = _env(1)
    obs = env.reset()
    for i, ob in enumerate(obs):
        g, r = GraphMap(ob["viewpoint"]), R.RefGraphMap(ob["viewpoint"])
        cur = ob
        for hop in range(3):
            g.update_graph(cur)
            r.update_graph(cur)
            ids = [None] + list(g.node_positions.keys())
            assert list(g.node_positions.keys()) == list(r.node_positions.keys())
            np.testing.assert_allclose(g.get_pos_fts(cur["viewpoint"], ids, cur["heading"], cur["elevation"]),
                                       r.get_pos_fts(cur["viewpoint"], ids, cur["heading"], cur["elevation"]), rtol=0, atol=1e-6)
            nxt = cur["candidate"][hop % len(cur["candidate"])]["viewpointId"]
            env.step([nxt if j == i else None for j in range(len(obs))], [cur["viewpoint"] if j == i else None for j in range(len(obs))])
            cur = env._get_obs()[i]
        e1, e2 = torch.randn(4), torch.randn(4)
        g.update_node_embed("x", e1)
        g.update_node_embed("x", e2)
        torch.testing.assert_close(g.get_node_embed("x"), (e1 + e2) / 2)
        g.update_node_embed("x", e2, rewrite=True)
        torch.testing.assert_close(g.get_node_embed("x"), e2)
        g.update_node_embed("x", e1, teacher=True)
        torch.testing.assert_close(g.get_node_embed("x", teacher=True), e1)


def test_pad_tensors_wgrad():
    a, b = torch.randn(2, 3, requires_grad=True), torch.randn(4, 3, requires_grad=True)
    out = pad_tensors_wgrad([a, b])
    assert out.shape == (2, 4, 3) and float(out[0, 2:].abs().sum()) == 0
    out.sum().backward()
    assert a.grad is not None and b.grad is not None


def _apply_csr(csr, log, n_out):
    ptr, idx, w = csr
    out = np.zeros((n_out, log.shape[1]))
    for n in range(n_out):
        for e in range(ptr[n], ptr[n + 1]):
            out[n] += w[e] * log[idx[e]]
    return out


@pytest.mark.parametrize("feedback,seed", [("teacher", 2), ("argmax", 5), ("sample", 7), ("teacher", 11)])
def test_plans_match_reference_loops(feedback, seed):
    torch.manual_seed(0)
    cfg = make_config(64, role="student", num_l_layers=1, num_x_layers=1, num_pano_layers=1)
    model = RefVLNBert(cfg).double()

    def call(mode, b):
        with torch.no_grad():
            return model(mode, {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()})
    T = 6
    draws = np.random.default_rng(seed).uniform(size=(T, 5))
    env_a, env_b = _env(seed), _env(seed)
    ref = R.rollout(env_a, call, env_a.reset(), feedback=feedback, max_action_len=T, sample_draws=draws, record="nav",
                    expert_policy="ndtw" if seed == 7 else "spl")
    obs = env_b.reset(features=False)
    pl = NavPlanner(env_b, obs, feedback=feedback, max_action_len=T, expert_policy="ndtw" if seed == 7 else "spl")
    lang = pl.language()
    want = R.language_variable(env_b._get_obs(False))
    assert (torch.from_numpy(lang["txt_ids"]) == want["txt_ids"]).all()
    H = 64
    log = np.zeros((0, H))
    for t, rs in enumerate(ref["steps"]):
        p = pl.begin_step()
        nav, pano = rs["nav"], rs["pano"]
        B, V, K, Vp = p["B"], p["V"], p["K"], p["Vp"]
        # panorama inputs: the gather indices reproduce the host-stacked features
        feats = env_b.feature_table[p["vp_rows"][:, None], np.maximum(p["view_order"], 0)] * (p["view_order"] >= 0)[..., None]
        np.testing.assert_array_equal(feats, pano["view_img_fts"].numpy())          # incl. the zero rows of padded slots
        np.testing.assert_allclose(p["loc_fts"], pano["loc_fts"].numpy(), atol=1e-6)
        assert (p["nav_types"] == pano["nav_types"].numpy()).all() and (p["view_lens"] == pano["view_lens"].numpy()).all()
        assert p["cand_vpids"] == pano["cand_vpids"]
        # map / local tokens
        assert p["gmap_vpids"] == nav["gmap_vpids"] and p["vp_cand_vpids"] == nav["vp_cand_vpids"]
        assert (p["gmap_step_ids"] == nav["gmap_step_ids"].numpy()).all()
        np.testing.assert_allclose(p["gmap_pos_fts"], nav["gmap_pos_fts"].numpy(), atol=1e-6)
        np.testing.assert_allclose(p["gmap_pair_dists"], nav["gmap_pair_dists"].numpy(), rtol=1e-6)
        assert (p["gmap_visited_masks"] == nav["gmap_visited_masks"].numpy()).all()
        assert (p["gmap_masks"] == nav["gmap_masks"].numpy()).all()
        assert list(p["no_vp_left"]) == nav["no_vp_left"]
        np.testing.assert_allclose(p["vp_pos_fts"], nav["vp_pos_fts"].numpy(), atol=1e-6)
        assert (p["vp_nav_masks"] == nav["vp_nav_masks"].numpy()).all() and (p["vp_masks"] == nav["vp_masks"].numpy()).all()
        assert (p["targets"] == rs["targets"].numpy()).all()
        # embeddings: CSR over the log == stack(get_node_embed) / cat([stop, mem, views])
        pe, pf, cls = (x.numpy() for x in rs["embeds"])
        assert p["log_base"] == log.shape[0]
        log = np.concatenate([log, pe.reshape(B * V, H), pf])
        got = _apply_csr(p["csr"], log, p["n_out"])
        np.testing.assert_allclose(got[:B * K].reshape(B, K, H), nav["gmap_img_embeds"].numpy(), atol=1e-12)
        np.testing.assert_allclose(got[B * K:].reshape(B, Vp, H), nav["vp_img_embeds"].numpy(), atol=1e-12)
        # transposed CSR is the exact transpose
        dense = np.zeros((p["n_out"], p["log_cls"]))
        ptr, idx, w = p["csr"]
        for n in range(p["n_out"]):
            for e in range(ptr[n], ptr[n + 1]):
                dense[n, idx[e]] += w[e]
        dt = np.zeros((p["log_cls"], p["n_out"]))
        ptr, idx, w = p["csr_t"]
        for n in range(p["log_cls"]):
            for e in range(ptr[n], ptr[n + 1]):
                dt[n, idx[e]] += w[e]
        np.testing.assert_allclose(dt, dense.T)
        log = np.concatenate([log, cls])
        done = pl.end_step(None if feedback == "teacher" else rs["a_t"].numpy())
        assert pl.actions == rs["actions"]
        assert done == (t == len(ref["steps"]) - 1)
    stop = [torch.softmax(rs["logits"], 1)[:, 0].numpy() for rs in ref["steps"]]
    traj = pl.finish(stop)
    assert [x["path"] for x in traj] == [x["path"] for x in ref["traj"]]


def test_native_planner_helpers_equal_the_python_forms():
    """csrc/hostplan.c (hop counts of FloydGraph.path, the nDTW expert's table rows) against the Python loops they replace: identical plans --
    position features (hop counts), expert targets under 'sample' feedback with the ndtw expert -- step by step over whole rollouts."""
    from magic_amd.host import hostplan
    if hostplan.lib() is None:
        pytest.skip("_magic_hostplan.so not built")
    saved = hostplan._lib

    def plans(native):
        hostplan._lib = saved if native else None
        env = _env(21, B=6, path_hops=(3, 5))
        rng = np.random.default_rng(3)
        pl = NavPlanner(env, env.reset(features=False), feedback="sample", max_action_len=8, expert_policy="ndtw")
        out = []
        for t in range(8):
            p = pl.begin_step()
            out.append((p["targets"].copy(), p["gmap_pos_fts"].copy(), p["vp_pos_fts"].copy()))
            acts = np.array([int(rng.integers(0, int(n))) for n in p["gmap_lens"]])          # arbitrary (valid-token) actions: off-path walks
            acts = np.where(np.asarray(p["gmap_visited_masks"])[np.arange(len(acts)), acts] | (acts == 1), p["targets"].clip(min=0), acts)
            if pl.end_step(acts):
                break
        return out
    try:
        a, b = plans(True), plans(False)
    finally:
        hostplan._lib = saved
    assert len(a) == len(b) >= 3
    for (ta, ga, va), (tb, gb, vb) in zip(a, b):
        assert np.array_equal(ta, tb) and np.array_equal(ga, gb) and np.array_equal(va, vb)


def test_native_planner_state_grows_past_its_initial_128_nodes_per_episode():
    """the C planner's by-dense-id arrays started at 128 nodes per episode and a larger map raised mid-rollout, while the Python planner (the
    reference's GraphMap, no bound) carried on.  An exploring rollout over a 260-node scan passes 128 map nodes: the native state doubles, re-registers,
    and every step's plan stays identical to the Python planner's."""
    from magic_amd.host import hostplan
    if hostplan.lib() is None:
        pytest.skip("_magic_hostplan.so not built")
    saved = hostplan._lib
    T = 90

    def plans(native):
        hostplan._lib = saved if native else None
        env = SynthNavEnv(batch_size=2, n_scans=1, nodes_per_scan=260, seed=5, instr_len=(6, 10), path_hops=(3, 4))
        pl = NavPlanner(env, env.reset(features=False), feedback="sample", max_action_len=T, expert_policy="spl")
        out = []
        for t in range(T):
            p = pl.begin_step()
            out.append((p["targets"].copy(), p["gmap_pos_fts"].copy(), p["gmap_pair_dists"].copy(), np.asarray(p["gmap_visited_masks"]).copy(), p["gmap_lens"].copy()))
            vis = np.asarray(p["gmap_visited_masks"])
            acts = []
            for i, n in enumerate(p["gmap_lens"]):          # explore: the LAST unvisited node of the map (the newest frontier), never stop / [mem]
                free = [k for k in range(2, int(n)) if not vis[i, k]]
                acts.append(free[-1] if free else 0)
            if pl.end_step(np.array(acts)):
                break
        return out, max(len(g.graph.names) for g in pl.gmaps), (pl.native.step.shape[1] if pl.native is not None else None)
    try:
        (a, na, cap), (b, nb_, _) = plans(True), plans(False)
    finally:
        hostplan._lib = saved
    assert na == nb_ and na > 128, (na, nb_)
    assert cap >= na and cap > hostplan.CAP
    assert len(a) == len(b) > 20
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)
