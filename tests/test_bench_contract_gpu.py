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
