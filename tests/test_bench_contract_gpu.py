"""The driver's bench contract (-m gpu): `python bench.py --gpus 1 --steps K --warmup W` prints exactly ONE JSON line on stdout with the
required keys, a roofline object and (at N = 1) a cpu_baseline object."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--pool", "3", "--no-secondary"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "bf16" and "workload" in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    # parity leg: both arithmetic modes against the fp64 oracle at full depth / vocabulary, and both modes' step times
    par, modes = d["parity"], d["modes"]
    assert par["fp32"]["max_abs_logit_delta"] < 1e-3 and par["fp32"]["argmax_agreement"] == 1.0
    assert par["bf16"]["max_abs_logit_delta"] < 1e-2 and d["bf16_max_logit_delta"] == par["bf16"]["max_abs_logit_delta"]
    assert modes["fp32"]["ms_per_step"] > 0 and d["fp32_mode_ms_per_step"] == modes["fp32"]["ms_per_step"]
    assert par["bf16x3"]["max_abs_logit_delta"] < 1e-3 and par["bf16x3"]["argmax_agreement"] == 1.0 and modes["bf16x3"]["ms_per_step"] > 0
    assert par["bf16"]["argmax_agreement"] >= 0.9 and par["bf16"]["worst_oracle_gap_between_flipped_picks"] <= 2 * par["bf16"]["max_abs_logit_delta"] + 1e-9
    # round 4: the tolerance status of the headline arithmetic at the top level, the parity-clean mode beside it, a 150-step figure, what the
    # teacher stream's start gate did, the id of the binary, and the RCCL calls of the exchange loaded in a world-1 group
    assert d["meets_north_star_tolerance"] == (par[d["dtype"]]["max_abs_logit_delta"] < 1e-3 and par[d["dtype"]]["argmax_agreement"] == 1.0)
    clean = d["parity_clean_mode"]
    assert clean is not None and clean["max_abs_logit_delta"] < 1e-3 and clean["argmax_agreement"] == 1.0 and clean["ms_per_step"] > 0
    assert d["steady"]["steps"] == 150 and d["ms_per_step_steady"] == d["steady"]["ms_per_step"] > 0
    assert set(d["steady"]["ms_per_step_by_task"]) == {"mlm", "sap", "cfp"} and min(d["steady"]["ms_per_step_by_task"].values()) > 0
    g = d["teacher_gate"]
    assert g["calls"] >= 4 + 2 + 150 and g["calls"] == g["opened"] + g["already_resident"] + g["timeouts"] + g["skipped"]
    assert g["disabled"] or g["timeouts"] <= 3 + g["opened"] + g["already_resident"], g       # timeouts never run unbounded: the gate turns itself off
    from magic_amd.host import lib as L
    assert d["build_id"] == L.source_build_id()
    assert d["rccl_smoke"]["ok"] is True and d["rccl_smoke"]["identity_at_world_1"] is True, d["rccl_smoke"]
    # round 6: what the gradients of the timed steps looked like (device words read after the timed region) -- a clip factor of 1e-5 (rounds 1-5: a
    # LayerNorm of the zero vector under zero-initialised biases) must not hide behind a throughput number again
    h = d["health"]
    assert 0.05 < h["grad_norm"] < 500 and 0.01 <= h["clip_factor"] <= 1.0 and h["max_grad_norm"] == 5.0, h
    for task, gn in d["steady"]["grad_norm_by_task"].items():
        assert 0.05 < gn["grad_norm"] < 500 and 0.01 <= gn["clip_factor"] <= 1.0, (task, gn)
    cpu = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] in ("reference", "port") and cpu["value"] > 0


def test_ctypes_binding_path_still_drives_a_full_step(monkeypatch):
    """the hot entry points normally go through the generated fast-call wrappers; the plain ctypes binding (the one INTEGRATION.md
    documents) must keep working: one oracle-checked training step with every entry point forced onto ctypes"""
    from magic_amd.host import lib as L
    from magic_amd.host import smoke as S
    lib = L.load()
    for name in L.SIGNATURES:
        monkeypatch.setitem(L._FN, name, getattr(lib, name))
    assert type(L._FN["magic_gemm"]).__name__ != "builtin_function_or_method"
    S.run_smoke()


def test_benchmarked_models_have_healthy_gradients_at_initialisation():
    """bench.py's models and batches, one eager step per proxy task -- the global gradient norm is O(1-100) (clip_grad_norm_ at 5.0
    scales by 0.01-1, not by 1e-5), no single parameter tensor holds more than 99 % of the squared norm, and the [stop] node's position row is the
    reference's [0, 1, 0, 1, 0, 0, 0] (pretrain_src/data/dataset.py:557-560), not zeros."""
    import torch
    sys.path.insert(0, ROOT)
    import bench as B
    from magic_amd.host import synth
    from magic_amd.host.plan import build_plan
    dev = torch.device("cuda", 0)
    _, _, teacher, student, trainer = B.build_models(torch.bfloat16, dev, 0.1, 1)
    bias = [student.store.master(n) for n, _, _ in student.store.specs if n.endswith("bias")]
    assert all(float(b.abs().max()) > 0 for b in bias), "checkpoint-like init: no all-zero bias vector"
    for i, task in enumerate(B.TASKS):
        b = synth.make_batch(task, batch_size=48, seed=1234, step=i)
        assert b["gmap_pos_fts"][:, 0].tolist() == [list(synth.STOP_NODE_POS_FTS)] * 48
        plan = build_plan(b, task, dev)
        trainer._zero_grad()
        trainer._fwd_bwd(synth.batch_to(b, dev), task, None, plan)
        torch.cuda.synchronize()
        g = student.store.grad.double()
        tot = float(g.pow(2).sum())
        share = max(float(g[off:off + n].pow(2).sum()) for off, n, _ in student.store.offsets.values()) / tot
        nrm = tot ** 0.5
        assert 0.05 < nrm < 500, (task, nrm)
        assert share < 0.99, (task, share)
        trainer._optimize()
        rep = trainer.opt.grad_norm_report()
        assert abs(rep["grad_norm"] - nrm) < 2e-3 * nrm and 0.01 <= rep["clip_factor"] <= 1.0, (task, rep, nrm)
