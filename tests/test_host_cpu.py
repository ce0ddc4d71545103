"""Host-side logic that needs no GPU: parameter naming / flat layout, plan building, schedule, config surgery."""
import json
import os

import numpy as np
import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import synth
from magic_amd.host.config import make_config, teacher_student_from_json
from magic_amd.host.model_pretrain import pretrain_specs
from magic_amd.host.params import ParamStore, is_no_decay
from magic_amd.host.plan import build_plan
from magic_amd.host.trainer import get_lr_sched
from oracle import model_ref as R
from oracle import optim_ref

KDL = dict(knowledge_distillation=True, kd_alpha=0.5, kd_temperature=2, teacher_sample_hard_mining=True,
           t_sample_preprocess_exp_decay=0.7, rw_temp=4,
           kdl_tasks=["txt", "img", "local", "global", "predict"], kdl_task_types=["emb", "attn"])


def test_param_names_match_oracle_state_dict():
    cfg = make_config(128, teacher_hidden_size=256, vocab_size=300, num_l_layers=2, num_x_layers=1, num_pano_layers=1, kdl=KDL)
    oracle = R.RefPretrainModel(cfg)
    want = {k: tuple(v.shape) for k, v in oracle.state_dict().items()}
    got = {n: tuple(s) for n, s, _ in pretrain_specs(cfg)}
    assert got == want


def test_flat_layout_groups_and_fused_qkv_contiguity():
    cfg = make_config(128, teacher_hidden_size=256, vocab_size=300, num_l_layers=1, num_x_layers=1, num_pano_layers=1)
    st = ParamStore(pretrain_specs(cfg), "cpu", torch.float32)
    p = "bert.lang_encoder.layer.0.attention.self."
    assert st.contiguous([p + "query.weight", p + "key.weight", p + "value.weight"])
    assert st.contiguous([p + "query.bias", p + "key.bias", p + "value.bias"])
    x = "bert.global_encoder.encoder.crossattention.0.crossattention.self."
    assert st.contiguous([x + "key.weight", x + "value.weight"]) and st.contiguous([x + "key.bias", x + "value.bias"])
    for name, (off, n, shape) in st.offsets.items():
        assert off % 64 == 0
        assert (off >= st.n_decay) == is_no_decay(name), name
    # reference rule (optim/misc.py:14): these LayerNorms are NOT matched by 'LayerNorm.weight'
    assert not is_no_decay("bert.img_embeddings.img_layer_norm.weight") and is_no_decay("bert.embeddings.LayerNorm.weight")
    # views alias the flat buffer
    st.master(p + "query.weight").fill_(3.0)
    off = st.offsets[p + "query.weight"][0]
    assert st.flat[off] == 3.0 and st.w_span(p + "query.weight", 384, 128)[0, 0] == 3.0


def test_module_state_dict_roundtrip_on_cpu_store():
    from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining
    cfg = make_config(128, teacher_hidden_size=256, vocab_size=300, num_l_layers=1, num_x_layers=1, num_pano_layers=1, kdl=KDL)
    oracle = R.RefPretrainModel(cfg)
    sd = oracle.state_dict()
    sd["some.unrelated.key"] = torch.zeros(3)                  # HF from_pretrained ignores extras
    m = GlocalTextPathCMTPreTraining.from_pretrained(None, config=cfg, state_dict=sd, device="cpu", compute_dtype=torch.float32)
    for k, v in m.state_dict().items():
        assert torch.equal(v, oracle.state_dict()[k]), k
    names = dict(m.named_parameters())
    assert "bert.txt_emb_w.weight" in names and names["bert.txt_emb_w.weight"].grad is not None
    assert not m.store.shadow_clean


def test_plan_matches_oracle_aggregation_and_fusion():
    batch = synth.make_batch("sap", batch_size=5, seed=11, min_len=5, max_len=9, min_steps=2, max_steps=5)
    plan = build_plan(batch, "sap", "cpu")
    Np, V, H = plan["Np"], 36, 8
    pe, pf = torch.randn(Np, V, H), torch.randn(Np, H)
    want = R.aggregate_gmap(pe, pf, batch)
    B, K = batch["gmap_step_ids"].shape

    def apply(csr, src, n_out):
        ptr, idx, w = [t.numpy() for t in csr]
        out = torch.zeros(n_out, H)
        for n in range(n_out):
            for e in range(ptr[n], ptr[n + 1]):
                out[n] += float(w[e]) * src[idx[e]]
        return out
    got = apply(plan["gmap_from_embed"], pe.view(-1, H), B * K) + apply(plan["gmap_from_fused"], pf, B * K)
    torch.testing.assert_close(got.view(B, K, H), want)
    # transpose CSR is the adjoint
    g = torch.randn(B * K, H)
    lhs = (apply(plan["gmap_from_embed"], pe.view(-1, H), B * K) * g).sum()
    rhs = (apply(plan["gmap_from_embed_T"], g, Np * V) * pe.view(-1, H)).sum()
    torch.testing.assert_close(lhs, rhs)
    # fusion map vs the oracle's python loops
    gl = torch.randn(B, K).masked_fill(~plan["gmask"].bool(), float("-inf"))
    ll = torch.randn(B, 37).masked_fill(~plan["lmask"].bool(), float("-inf"))
    want = R.fuse_logits(gl, ll, batch)
    fl = gl.clone()
    for b in range(B):
        bw = sum(ll[b, j] for j in range(37) if plan["bwmask"][b, j] and plan["lmask"][b, j])
        for k in range(K):
            s = int(plan["fsrc"][b, k])
            fl[b, k] = gl[b, k] + (ll[b, s] if s >= 0 else (bw if s == -2 else 0.0))
    assert torch.equal(torch.isinf(fl), torch.isinf(want))
    torch.testing.assert_close(torch.nan_to_num(fl, neginf=0), torch.nan_to_num(want, neginf=0))


def test_schedule_matches_oracle():
    for s in (0, 1, 9999, 10000, 150000, 200000, 300000):
        assert get_lr_sched(s, 5e-5, 10000, 200000) == optim_ref.get_lr_sched(s, 5e-5, 10000, 200000)


def test_config_surgery_matches_reference_rules(tmp_path):
    base = dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, teacher_hidden_size=256, teacher_num_l_layers=6,
                teacher_mlp_ratio=4, student_hidden_size=128, student_num_l_layers=6, student_mlp_ratio=4, vocab_size=50265)
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(base))
    t, s = teacher_student_from_json(str(p), KDL)
    assert (t.hidden_size, t.num_attention_heads, t.intermediate_size, t.role) == (256, 4, 1024, "teacher")
    assert (s.hidden_size, s.num_attention_heads, s.intermediate_size, s.role) == (128, 2, 512, "student")
    assert s.teacher_hidden_size == 256 and s.kd and s.kdl["kd_alpha"] == 0.5


def test_synthetic_batches_are_deterministic_and_r2r_shaped():
    a = synth.make_batch("sap", batch_size=48, seed=1234, step=3)
    b = synth.make_batch("sap", batch_size=48, seed=1234, step=3)
    assert all(torch.equal(a[k], b[k]) for k in a if torch.is_tensor(a[k]))
    assert a["traj_view_img_fts"].shape[1:] == (36, 768) and a["txt_ids"].shape[1] <= 80
    assert 48 * 4 <= a["traj_view_img_fts"].shape[0] <= 48 * 7
    assert a["vp_pos_fts"].shape == (48, 37, 14) and (a["txt_ids"][:, 0] == 0).all()


def test_out_of_range_batch_values_are_rejected_on_the_host():
    """ids the kernels would use as table rows are range-checked against the config before any launch"""
    from types import SimpleNamespace
    from magic_amd.host import synth
    from magic_amd.host.plan import build_plan, check_plan
    cfg = make_config(128, vocab_size=300)
    batch = synth.make_batch("sap", batch_size=2, seed=1, vocab=300, min_len=5, max_len=7, min_steps=1, max_steps=2)
    plan = build_plan(batch, "sap", torch.device("cpu"))
    check_plan(plan, cfg)                                            # fits
    for key, val, field in (("txt_ids", 300, None), ("gmap_step_ids", 100, None), ("traj_nav_types", 3, None), ("txt_ids", -1, None)):
        b = dict(batch)
        t = batch[key].clone()
        t.view(-1)[0] = val
        b[key] = t
        with pytest.raises(ValueError):
            check_plan(build_plan(b, "sap", torch.device("cpu")), cfg)
    long_cfg = make_config(128, vocab_size=300, max_position_embeddings=6)
    with pytest.raises(ValueError):
        check_plan(build_plan(batch, "sap", torch.device("cpu")), long_cfg)


def test_feature_table_envedit_mixing_follows_the_reference_coin_stream():
    """get_scanvp_feature (dataset.py:606-610): one `np.random.rand() > 0.5` per viewpoint visit, in path order, picks the
    augmented feature file; with the packed table that is an offset of n rows on the index."""
    import numpy as np
    from magic_amd.host.feature_table import FeatureTable
    rng = np.random.default_rng(0)
    keys = [f"s_{i}" for i in range(5)]
    base = [rng.standard_normal((36, 8)).astype(np.float32) for _ in keys]
    aug = [b + 100 for b in base]
    ft = FeatureTable.from_arrays(keys, base, device="cpu", dtype=torch.float32, image_feat_size=8, aug_arrays=aug)
    assert ft.has_aug and ft.table.shape == (10, 36, 8)
    cands = lambda scan, vp: {f"c{vp}": [3, 1.0, 0.1, 0.0]}
    paths = [["0", "1", "2"], ["3", "4"]]
    st = np.random.RandomState(5)
    b = ft.batch_indices(["s", "s"], paths, cands, aug_coin=st.rand)
    st2 = np.random.RandomState(5)
    want = [int(vp) + (5 if st2.rand() > 0.5 else 0) for path in paths for vp in path]      # the reference's draw order
    assert b["vp_row"].tolist() == want and any(r >= 5 for r in want) and any(r < 5 for r in want)
    plain = ft.batch_indices(["s", "s"], paths, cands)
    assert plain["vp_row"].tolist() == [0, 1, 2, 3, 4]


def test_packed_loader_record_round_trips():
    """host/loader.pack (DataLoader worker side) -> unpack (training process) gives the same batch entries and index plan as the
    direct path, for every task, incl. an index-only (feature-table) batch"""
    import numpy as np
    from magic_amd.host import synth
    from magic_amd.host.loader import MODEL_KEYS, PlanCollate, pack, unpack
    from magic_amd.host.plan import build_plan, build_plan_host
    for task in ("mlm", "sap", "cfp", "mrc"):
        b = synth.make_batch(task, batch_size=5, seed=3, step=1, vocab=300, min_len=6, max_len=12, min_steps=2, max_steps=4)
        if task == "cfp":
            Np, V = b.pop("traj_view_img_fts").shape[:2]
            b["traj_vp_row"] = torch.arange(Np, dtype=torch.int32)
            b["traj_view_order"] = torch.arange(V, dtype=torch.int32)[None].repeat(Np, 1)
        want = build_plan(b, task, "cpu")
        rec = pack(b, build_plan_host(b, task))
        assert rec["buf"].dtype == torch.uint8 and rec["buf"].dim() == 1
        gb, gp = unpack(rec, "cpu")
        for k in MODEL_KEYS:
            if torch.is_tensor(b.get(k)):
                assert torch.equal(gb[k], b[k]), k
        for k, v in want.items():
            if k == "_stage":
                continue
            if torch.is_tensor(v):
                assert torch.equal(gp[k], v) and gp[k].dtype == v.dtype, k
            elif isinstance(v, tuple):
                assert all(torch.equal(x, y) for x, y in zip(gp[k], v)), k
            elif isinstance(v, np.ndarray):
                assert (gp[k] == v).all(), k
            else:
                assert gp[k] == v, k
    samples = [dict(x=i) for i in range(3)]
    pc = PlanCollate(lambda inp: synth.make_batch("sap", batch_size=len(inp), seed=1, vocab=300, min_len=6, max_len=9, min_steps=2, max_steps=3), "sap")
    assert set(pc(samples)) == {"buf", "blob"}


def test_direct_rccl_binding_stays_out_of_the_way_without_an_nccl_group():
    """host/rccl.py: no process group (or a gloo one) -> make() returns None and GradSync keeps torch.distributed; the module itself imports on a CPU box"""
    import torch
    from magic_amd.host import rccl
    assert rccl.make(torch.device("cpu")) is None
    assert rccl.NCCL_FLOAT32 == 7 and rccl.NCCL_INT64 == 4 and rccl.NCCL_SUM == 0          # rccl.h ncclDataType_t / ncclRedOp_t
    import ctypes
    assert ctypes.sizeof(rccl._UniqueId) == 128
