"""The oracle's distillation / optimizer / collate restatements vs vectors minted from the reference
(tests/golden/mint_golden.py).  CPU only."""
import os
from collections import defaultdict

import pytest
import torch
import torch.nn as nn

from oracle import makd_ref as M
from oracle import optim_ref


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


def test_kd_primitives_match_reference(golden_dir):
    fx = _load(golden_dir, "makd_primitives.pt")
    s, t, w, c = fx["s"], fx["t"], fx["w"], fx["cases"]
    for T in (1, 2):
        torch.testing.assert_close(M.kd_loss(s, t, T, flavour="pretrain"), c[f"pre_kd_T{T}"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(M.kd_loss(s, t, T, w, flavour="pretrain"), c[f"pre_kd_T{T}_w"], rtol=1e-5, atol=1e-6)
        for lt in ("sum", "mean"):
            torch.testing.assert_close(M.kd_loss(s, t, T, None, lt), c[f"nav_kd_T{T}_{lt}"], rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(M.kd_loss(s, t, T, w, lt), c[f"nav_kd_T{T}_{lt}_w"], rtol=1e-5, atol=1e-6)
    fs, ft, fa, fb = fx["fs"], fx["ft"], fx["fa"], fx["fb"]
    torch.testing.assert_close(M.mse_loss(fs, ft, flavour="pretrain"), c["pre_mse"])
    torch.testing.assert_close(M.mse_loss(fs, ft, w, flavour="pretrain"), c["pre_mse_w"])
    torch.testing.assert_close(M.mse_loss(fs, ft, fx["wbad"], flavour="pretrain"), c["pre_mse_wbad"])
    torch.testing.assert_close(M.mse_loss(fa, fb, w, flavour="pretrain"), c["pre_mse4_w"])
    for lt in ("sum", "mean"):
        torch.testing.assert_close(M.mse_loss(fs, ft, None, lt), c[f"nav_mse_{lt}"])
        torch.testing.assert_close(M.mse_loss(fs, ft, w, lt), c[f"nav_mse_{lt}_w"])
        torch.testing.assert_close(M.mse_loss(fa, fb, w, lt), c[f"nav_mse4_{lt}_w"])
    assert int(c["nav_mse_wbad_raises"]) == 1
    with pytest.raises(ValueError):
        M.mse_loss(fs, ft, fx["wbad"], "sum")
    torch.testing.assert_close(M.exponential_decay(fx["losses"], 0.7), c["exp_decay_0.7"])
    torch.testing.assert_close(M.exponential_decay(fx["losses"], 0.7), c["exp_decay_pre_0.7"])
    torch.testing.assert_close(M.invert_normalized_losses(fx["losses"]), c["invert_norm"])


def _heads(fx):
    heads = {}
    for n, (w, b) in fx["heads"].items():
        lin = nn.Linear(w.shape[1], w.shape[0])
        lin.weight.data.copy_(w)
        lin.bias.data.copy_(b)
        heads[n] = lin
    return heads


def test_nav_makd_matches_compute_kd_losses(golden_dir):
    fx = _load(golden_dir, "makd_agent.pt")
    heads = _heads(fx)
    s_out, t_out, rw = fx["s_out"], fx["t_out"], fx["rw"]
    for name, want in fx["cases"].items():
        parts = name.split("_")
        role, t = parts[0], int(parts[1][1:])
        acc = defaultdict(float)
        learned = "learned" in name             # 'learned_weight': softplus of the LEARNER's five kdl_*_weight scalars (student: t2s, teacher: s2t)
        if role == "t2s":
            mode = "learned_weight" if learned else (None if parts[2] == "None" else "RW")
            got = M.nav_makd(t, s_out, t_out, heads, acc, role="t2s", loss_type=parts[-1], weights=rw, weight_mode=mode, learned=fx["learned_student"])
        else:
            got = M.nav_makd(t, t_out, s_out, heads, acc, role="s2t", weights=rw, weight_mode="learned_weight" if learned else "RW",
                             learned=fx["learned_teacher"])
        assert set(got) == set(want), name
        for k in want:
            torch.testing.assert_close(torch.as_tensor(float(got[k])), want[k], rtol=2e-5, atol=1e-6, msg=f"{name}:{k}")


def test_episode_loss_formula():
    # agent.py:1110-1123
    assert abs(M.episode_loss(12.0, 8.0, 4, 0.2, 0.5) - (0.5 * 3.0 + 0.5 * 0.4)) < 1e-12


def test_adamw_and_schedule_match_reference(golden_dir):
    fx = _load(golden_dir, "adamw.pt")
    params = [p.clone() for p in fx["p0"]]
    state = optim_ref.adamw_init(params)
    for st in range(3):
        optim_ref.adamw_step(params, fx["grads"][st], state, lr=fx["lrs"][st], betas=(0.9, 0.98), eps=1e-6,
                             weight_decay=[0.01, 0.0])
        for p, want in zip(params, fx["hist"][st]):
            torch.testing.assert_close(p, want, rtol=1e-6, atol=1e-7)
    for s, want in zip(fx["sched_steps"], fx["sched"]):
        assert abs(optim_ref.get_lr_sched(s, 5e-5, 10000, 200000) - want) < 1e-15


def test_synthetic_collate_matches_reference_collate(golden_dir):
    import magic_amd  # noqa: F401
    from magic_amd.host import synth
    fx = _load(golden_dir, "collate.pt")
    for task in ("sap", "cfp"):
        got = synth.collate(fx["samples"], task)
        want = fx[task]
        for k, v in want.items():
            if k == "extra_heads":
                continue
            if torch.is_tensor(v):
                assert got[k].dtype == v.dtype and got[k].shape == v.shape, (task, k)
                assert torch.equal(got[k], v), (task, k)
            else:
                assert got[k] == v, (task, k)
    # mlm: same masked items
    import numpy as np
    mrng = np.random.default_rng(5)
    got = synth.collate(fx["samples"], "mlm", rng=mrng)
    for k, v in fx["mlm"].items():
        if torch.is_tensor(v):
            assert torch.equal(got[k], v), k
        else:
            assert got[k] == v, k


def test_seq_masks_and_padding(golden_dir):
    from oracle.model_ref import seq_mask
    fx = _load(golden_dir, "ops.pt")
    assert torch.equal(seq_mask(fx["lens"], 5), fx["masks"])
    assert torch.equal(seq_mask(fx["lens"], 8), fx["masks8"])
    import magic_amd  # noqa: F401
    from magic_amd.host.synth import _pad_stack
    assert torch.equal(_pad_stack(fx["ts"]), fx["padded"])


def test_mrc_collate_and_validate_arithmetic_match_reference(golden_dir):
    """synth.collate('mrc') vs the reference's MrcDataset masking + mrc_collate; oracle/engine target extraction order and
    validate_mrc's KL-sum / soft-target accuracy vs numbers computed with the reference's primitives."""
    import numpy as np
    import torch.nn.functional as F
    import magic_amd  # noqa: F401
    from magic_amd.host import synth
    from magic_amd.host.plan import build_plan
    fx = _load(golden_dir, "collate_mrc.pt")
    samples = _load(golden_dir, "collate.pt")["samples"]
    got = synth.collate(samples, "mrc", rng=np.random.default_rng(fx["seed"]))
    for k, v in fx["mrc"].items():
        if k == "vp_angles":
            continue
        if torch.is_tensor(v):
            assert got[k].dtype == v.dtype and got[k].shape == v.shape, k
            assert torch.equal(got[k], v), k
        else:
            assert got[k] == v, k
    # the masked rows really are zeroed in the last panorama of each sample, and nowhere else
    last = torch.tensor(got["traj_step_lens"]).cumsum(0) - 1
    z = (got["traj_view_img_fts"][last].abs().sum(-1) == 0)
    assert torch.equal(z, got["vp_view_mrc_masks"])
    # target order = boolean-mask order (tasks.py:183-187); the plan gathers rows b*Vp + 1 + v in the same order
    plan = build_plan(got, "mrc", torch.device("cpu"))
    assert torch.equal(plan["mrc_targets"], fx["targets"])
    rows = plan["mrc_rows"][1][: plan["n_mrc"]]
    bv = torch.nonzero(got["vp_view_mrc_masks"])
    assert torch.equal(rows.long(), bv[:, 0] * plan["Vp"] + 1 + bv[:, 1])
    # validate_mrc arithmetic on the fixture logits
    kl = F.kl_div(F.log_softmax(fx["logits"], -1), plan["mrc_targets"], reduction="sum")
    assert abs(float(kl) - fx["kl_sum"]) < 1e-4
    assert int((fx["logits"].argmax(-1) == plan["mrc_targets"].argmax(-1)).sum()) == fx["n_correct"]


def test_ingest_token_order_matches_reference_get_traj_pano_fts(golden_dir):
    """oracle/ingest_ref.py and the host-side index builder (host/feature_table.py) vs the reference's own
    get_traj_pano_fts output (dataset.py:729-772) on synthetic candidate tables, incl. a panorama where two candidates share
    a view (37 tokens) -- features, loc_fts, nav_types, candidate ids, and the padded batch layout."""
    import numpy as np
    import magic_amd  # noqa: F401
    from magic_amd.host import feature_table as FT
    from oracle import ingest_ref as IR
    fx = _load(golden_dir, "ingest.pt")
    D, scan = fx["D"], fx["scan"]
    for i, base in enumerate((0, 12, 23)):
        assert np.array_equal(IR.get_view_rel_angles(base), fx["rel"][i].numpy())
        assert np.array_equal(FT.get_view_rel_angles(base), fx["rel"][i].numpy())
    a = fx["ang_in"].numpy()
    assert np.array_equal(IR.get_angle_fts(a[:, 0], a[:, 1], 4), fx["ang_fts4"].numpy())
    assert np.array_equal(IR.get_angle_fts(a[:, 0], a[:, 1], 8), fx["ang_fts8"].numpy())
    store = {k: v.numpy() for k, v in fx["store"].items()}
    cands = fx["cands"]
    keys = sorted(store)
    table = np.stack([store[k][:, :D] for k in keys])
    index = {k: i for i, k in enumerate(keys)}
    for path, want in zip(fx["paths"], fx["outs"]):
        f, loc, nav, cv, last = IR.traj_pano_tokens(lambda vp: store[f"{scan}_{vp}"][:, :D], path, lambda vp: cands[f"{scan}_{vp}"])
        assert cv == want["cand"] and [list(x) for x in nav] == want["nav"]
        for t in range(len(path)):
            assert np.array_equal(f[t], want["fts"][t].numpy())
            assert np.array_equal(loc[t], want["loc"][t].numpy())
        assert np.array_equal(last, want["last"].numpy())
        # index-only description + gather restatement reproduce the same tensors
        for t, vp in enumerate(path):
            order, loc2, nav2, cv2 = FT.pano_view_order(cands[f"{scan}_{vp}"])
            got = IR.view_gather(table, np.array([index[f"{scan}_{vp}"]]), order[None])[0]
            assert np.array_equal(got, want["fts"][t].numpy())
            assert np.allclose(loc2, want["loc"][t].numpy(), atol=1e-6) and list(nav2) == want["nav"][t] and cv2 == want["cand"][t]
    # padded batch layout = what pad_tensors / pad_sequence of the collates produce from the per-step lists
    ft = FT.FeatureTable(keys, torch.from_numpy(table))
    b = ft.batch_indices([scan] * 3, fx["paths"], lambda sc, vp: cands[f"{sc}_{vp}"])
    assert b["traj_step_lens"] == [3, 5, 1] and b["order"].shape == (9, 37)
    flat_f = sum([o["fts"] for o in fx["outs"]], [])
    assert b["traj_vp_view_lens"].tolist() == [x.shape[0] for x in flat_f]
    got = IR.view_gather(table, b["vp_row"].numpy(), b["order"].numpy())
    for p, x in enumerate(flat_f):
        assert np.array_equal(got[p, :x.shape[0]], x.numpy()) and not got[p, x.shape[0]:].any()
    flat_l = sum([o["loc"] for o in fx["outs"]], [])
    for p, x in enumerate(flat_l):
        assert np.allclose(b["traj_loc_fts"][p, :x.shape[0]].numpy(), x.numpy(), atol=1e-6) and not b["traj_loc_fts"][p, x.shape[0]:].any()


def test_ragged_view_collate_matches_reference(golden_dir):
    """panoramas with 36 or 37 view tokens in one batch: synth.collate pads exactly like the reference's sap_collate"""
    import random
    import numpy as np
    import magic_amd  # noqa: F401
    from magic_amd.host import synth
    from magic_amd.host.plan import build_plan
    fx = _load(golden_dir, "collate_ragged.pt")
    rng = np.random.default_rng([fx["seed"], 0])
    pyrng = random.Random(fx["seed"])
    samples = [synth.make_sample(rng, pyrng, uid=i, min_len=5, max_len=9, min_steps=2, max_steps=4, dup_view_prob=0.5, img_dim=16)
               for i in range(4)]
    got = synth.collate(samples, "sap")
    for k, v in fx["sap"].items():
        if torch.is_tensor(v):
            assert got[k].dtype == v.dtype and got[k].shape == v.shape, k
            assert torch.equal(got[k], v), k
        else:
            assert got[k] == v, k
    plan = build_plan(got, "sap", torch.device("cpu"))
    assert plan["V"] == got["traj_view_img_fts"].shape[1] == 37 and plan["Vp"] == got["vp_pos_fts"].shape[1]
    assert plan["vp_mask"].sum(1).tolist() == [int(got["traj_vp_view_lens"][r]) + 1 for r in plan["last_rows"]]


# ---- navigator loop: FloydGraph and the agent's input builders ------------------------------------------
class _GoldenEnv:
    """shortest_distances / shortest_paths of the stepper the fixture was minted on (dense tables stored in the fixture)"""

    def __init__(self, e):
        self.e = e
        self.shortest_distances = {n: _Rows(e, n, False) for n in e["vps"]}
        self.shortest_paths = {n: _Rows(e, n, True) for n in e["vps"]}


class _Rows:
    def __init__(self, e, n, paths):
        self.vps, self.sd, self.nxt, self.paths = e["vps"][n], e["shortest"][n], e["nxt"][n], paths

    def __getitem__(self, a):
        ia, me = self.vps.index(a), self

        class Row:
            def __getitem__(_, b):
                ib = me.vps.index(b)
                if not me.paths:
                    return float(me.sd[ia, ib])
                out, x = [a], ia
                while x != ib:
                    x = int(me.nxt[x, ib])
                    out.append(me.vps[x])
                return out
        return Row()


def test_floyd_matches_reference(golden_dir):
    from magic_amd.host.graph_map import FloydGraph
    from oracle.rollout_ref import RefFloyd
    fx = _load(golden_dir, "nav_loop.pt")["floyd"]
    names = fx["names"]
    for G in (RefFloyd, FloydGraph):
        # the fixture records the state after each edge and the update that may follow it
        g, k, it = G(), 0, iter(fx["answers"])
        script = fx["script"]
        while k < len(script):
            op, x, y, d = script[k]
            g.add_edge(names[x], names[y], d)
            k += 1
            if k < len(script) and script[k][0] == "update":
                g.update(names[script[k][1]])
                k += 1
            for u, v, dist, path, seen in next(it):
                assert g.distance(names[u], names[v]) == dist
                assert g.visited(names[u]) == seen
                if path is not None:
                    assert [names.index(q) for q in g.path(names[u], names[v])] == path


def test_nav_builders_match_reference(golden_dir):
    """oracle/rollout_ref.py's builders, replayed on the fixture's observations, against what the reference's own
    GMapNavAgent methods returned for them (agent.py:63-373)"""
    from oracle import rollout_ref as R
    fx = _load(golden_dir, "nav_loop.pt")
    env = _GoldenEnv(fx["env"])
    lv = R.language_variable(fx["lang"]["obs"])
    assert (lv["txt_ids"] == fx["lang"]["txt_ids"]).all() and (lv["txt_masks"] == fx["lang"]["txt_masks"]).all()
    obs0 = fx["steps"][0]["obs"]
    gmaps = [R.RefGraphMap(ob["viewpoint"]) for ob in obs0]
    for g, ob in zip(gmaps, obs0):
        g.update_graph(ob)
    for t, st in enumerate(fx["steps"]):
        obs, ended = st["obs"], st["ended"]
        if t > 0:
            prev_ended = fx["steps"][t - 1]["ended"]
            for i, ob in enumerate(obs):
                if not prev_ended[i]:
                    gmaps[i].update_graph(ob)
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.node_step_ids[obs[i]["viewpoint"]] = t + 1
        pano = R.panorama_variable(obs, feat=16)
        want = st["pano"]
        for k in ("view_img_fts", "loc_fts", "nav_types", "view_lens"):
            torch.testing.assert_close(pano[k], want[k], rtol=0, atol=0)
        assert pano["cand_vpids"] == want["cand_vpids"]
        pe, pf = st["pe"], st["pf"]
        for i, g in enumerate(gmaps):
            if ended[i]:
                continue
            g.update_node_embed(obs[i]["viewpoint"], pf[i], rewrite=True)
            for j, cv in enumerate(pano["cand_vpids"][i]):
                if not g.graph.visited(cv):
                    g.update_node_embed(cv, pe[i, j])
        nav = R.nav_gmap_variable(obs, gmaps, st["last"], teacher=False)
        nav.update(R.nav_vp_variable_mem(obs, gmaps, pe, pano["cand_vpids"], pano["view_lens"], pano["nav_types"], st["last"]))
        ref = st["nav"]
        assert nav["gmap_vpids"] == ref["gmap_vpids"] and nav["vp_cand_vpids"] == ref["vp_cand_vpids"]
        assert nav["no_vp_left"] == ref["no_vp_left"]
        for k in ("gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_visited_masks", "gmap_pair_dists", "gmap_masks",
                  "vp_img_embeds", "vp_pos_fts", "vp_masks", "vp_nav_masks"):
            torch.testing.assert_close(nav[k], ref[k], rtol=0, atol=0, msg=f"step {t}: {k}")
        tr = st["traj"]
        for pol, key in ((None, "tgt_il"), ("spl", "tgt_spl"), ("ndtw", "tgt_ndtw")):
            got = R.teacher_action(env, obs, nav["gmap_vpids"], ended, nav["gmap_visited_masks"], pol is None, t, tr, pol or "spl")
            assert (got == st[key]).all(), (t, key)


def test_map_position_features_and_the_stop_node_row_match_reference_get_gmap_pos_fts(golden_dir):
    """oracle/ingest_ref.gmap_pos_fts vs the reference's own get_gmap_pos_fts (pretrain_src/data/dataset.py:553-575) on a synthetic scan graph, and the
    synthetic generator's [stop]-node row vs the row the reference builds for vpid None -- [sin 0, cos 0, sin 0, cos 0, 0, 0, 0], not zeros (rounds 1-5
    of host/synth.py wrote zeros there, which made the benchmarked step's `gmap_pos_embeddings` a LayerNorm of the zero vector)."""
    import random

    import numpy as np
    import magic_amd  # noqa: F401
    from magic_amd.host import synth
    from oracle import ingest_ref as IR
    fx = _load(golden_dir, "gmap_pos.pt")
    pos = {k: v.numpy() for k, v in fx["pos"].items()}
    for c in fx["cases"]:
        got = IR.gmap_pos_fts(lambda v: pos[v], lambda a, b: fx["dist"][a][b], lambda a, b: fx["path_len"][a][b], c["cur"], c["ids"],
                              c["heading"], c["elevation"], max_dist=fx["max_dist"], max_step=fx["max_step"])
        assert got.dtype == np.float32 and np.array_equal(got, c["out"].numpy()), c["cur"]
        assert c["ids"][0] is None
        assert c["out"][0].tolist() == list(synth.STOP_NODE_POS_FTS)
    s = synth.make_sample(np.random.default_rng(3), random.Random(3), uid=0)
    assert s["gmap_vpids"][0] is None and s["gmap_pos_fts"][0].tolist() == list(synth.STOP_NODE_POS_FTS)
    b = synth.make_batch("sap", batch_size=4, seed=9, step=0)
    assert torch.equal(b["gmap_pos_fts"][:, 0], torch.tensor(synth.STOP_NODE_POS_FTS).expand(4, 7))


def test_map_inputs_and_viewpoint_position_features_match_reference_get_gmap_inputs(golden_dir):
    """oracle/ingest_ref.gmap_inputs / vp_pos_fts vs the reference's own get_gmap_inputs (pretrain_src/data/dataset.py:520-552) and get_vp_pos_fts
    (:555-565) on a synthetic scan whose walk visits a node it first saw as a candidate: token order ([stop] | visited in visiting order | frontier in
    first-sighting order), step ids, both visited-mask rules, position features, pair distances (row / column 0 and diagonal zero), and the
    [vp_ft_len + 1, 14] viewpoint table (start viewpoint in columns 0..6 of EVERY row, candidates in columns 7..13 of rows 1..n) -- bit-exact.  The
    synthetic generator (host/synth.py) is held to the same layout."""
    import random

    import numpy as np
    import magic_amd  # noqa: F401
    from magic_amd.host import synth
    from oracle import ingest_ref as IR
    fx = _load(golden_dir, "gmap_inputs.pt")
    pos = {k: v.numpy() for k, v in fx["pos"].items()}
    pos_of, dist_of, len_of = (lambda v: pos[v]), (lambda a, b: fx["dist"][a][b]), (lambda a, b: fx["path_len"][a][b])
    kw = dict(max_dist=fx["max_dist"], max_step=fx["max_step"])
    assert len(fx["cases"]) == 6
    for c in fx["cases"]:
        ids, steps, vis, p, pair = IR.gmap_inputs(lambda v: fx["cand_lists"][v], pos_of, dist_of, len_of, c["path"], c["heading"], c["elevation"],
                                                  act_visited_node=c["act_visited_node"], **kw)
        assert ids == c["ids"] and steps == c["steps"] and vis == c["vis"], c["path"]
        assert np.array_equal(p, c["pos"].numpy()) and np.array_equal(pair, c["pair"].numpy()) and pair.dtype == np.float32
        assert not pair[0].any() and not pair[:, 0].any() and not np.diag(pair).any()
        vp = IR.vp_pos_fts(pos_of, dist_of, len_of, c["path"][0], c["path"][-1], c["cand"], c["heading"], c["elevation"], 36, **kw)
        assert vp.shape == (37, 14) and np.array_equal(vp, c["vp_pos"].numpy())
        assert (vp[:, :7] == vp[0, :7]).all() and not vp[0, 7:].any() and not vp[len(c["cand"]) + 1:, 7:].any()
    long = [c for c in fx["cases"] if len(c["path"]) == 5 and not c["act_visited_node"]][0]
    assert long["ids"][:6] == [None] + long["path"] and "vp4" not in long["ids"][6:]           # vp4: first a candidate (of vp1), later visited
    assert long["steps"][:6] == [0, 1, 2, 3, 4, 5] and set(long["steps"][6:]) == {0} and long["vis"] == [0] + [1] * 5 + [0] * (len(long["ids"]) - 6)
    s = synth.make_sample(np.random.default_rng(4), random.Random(4), uid=0)
    T = len(s["traj_vpids"])
    assert s["gmap_vpids"][0] is None and s["gmap_vpids"][1:1 + T] == s["traj_vpids"]
    assert s["gmap_step_ids"].tolist() == [0] + list(range(1, T + 1)) + [0] * (len(s["gmap_vpids"]) - 1 - T)
    assert s["gmap_visited_masks"].tolist() == [False] + [True] * T + [False] * (len(s["gmap_vpids"]) - 1 - T)
    d = s["gmap_pair_dists"]
    assert not d[0].any() and not d[:, 0].any() and not d.diagonal().any() and torch.equal(d, d.t())
    vp = s["vp_pos_fts"]
    nc = len(s["traj_cand_vpids"][-1])
    assert vp.shape[1] == 14 and (vp[:, :7] == vp[0, :7]).all() and not vp[0, 7:].any() and not vp[nc + 1:, 7:].any() and vp[1:nc + 1, 7:].abs().sum() > 0

