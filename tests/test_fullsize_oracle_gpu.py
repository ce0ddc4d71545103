"""Oracle parity at the TRUE MAGIC-S pretraining configuration (-m gpu): 6 text / 3 cross-modal / 2 panorama layers, vocabulary
50265, student H=128 vs frozen H=256 teacher, B=8, <= 80 tokens, 36 views x 768 -- the network bench.py times, only the batch is
smaller (the fp64 oracle is the checker; its cost per B=8 step is seconds).

North-star bar (BASELINE.json): action argmax bit-exact, action-logit |delta| < 1e-3.
  * fp32 engine: every forward tensor, the three logit tensors (tolerance written below), argmax, all loss terms, every
    parameter gradient against the fp64 oracle's autograd.
  * bf16 engine (the arithmetic bench.py's headline runs in): the same quantities, to the tolerances STATED below (bf16 cannot
    meet 1e-3: see BF16_LOGIT_TOL); the achieved numbers are printed (pytest -s) and are what bench.py reports as `parity`.
"""
import json

import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import synth
from oracle import parity_probe as PP
from tests.test_model_gpu import close, view_outputs

pytestmark = pytest.mark.gpu

# ---- stated tolerances ------------------------------------------------------------------------------------------------
FP32_LOGIT_TOL = 2e-4          # measured ~1e-5; north-star bar 1e-3
# bf16: measured 4.8e-3 worst over 3 x 8 trajectories (logits ~0.2-0.4, nearest oracle tie 2e-4).  The 1e-3 bar is out of reach
# for single-bf16 operands by construction: rounding ONLY the weights to bf16 inside the fp64 oracle already moves its own logits
# by 1.3e-3 -- the bar is met by the fp32-MFMA mode above, whose throughput bench.py prints next to bf16's.
BF16_LOGIT_TOL = 1e-2
# action selection in bf16: identical wherever the oracle separates its two best candidates by more than twice the logit error; on
# random-init weights some rows are near-ties (smallest top-2 gap of these batches: 2.3e-4, far inside the 4.8e-3 error), and a pick
# may flip THERE -- measured 23-24 of 24 rows identical.  The assertion: >= 90 % identical AND every flip inside the tie band.
BF16_ARGMAX_MIN = 0.9


@pytest.fixture(scope="module")
def models():
    return PP.oracle_models()


@pytest.mark.parametrize("task", ["sap", "mlm", "cfp"])
def test_fp32_engine_matches_fp64_oracle_at_full_depth_and_vocab(models, task):
    tcfg, scfg, o_t, o_s = models
    assert (scfg.num_l_layers, scfg.num_x_layers, scfg.num_pano_layers, scfg.vocab_size) == (6, 3, 2, 50265)
    g_t, g_s = PP.engine_models(tcfg, scfg, o_t, o_s, torch.float32)
    batch = synth.make_batch(task, batch_size=8, seed=1234, step=0)
    ot, want = PP.oracle_step(o_t, o_s, batch, task, backward=True)
    gt, got = PP.engine_step(g_t, g_s, batch, task, backward=True)
    plan = gt["plan"]
    assert plan["L"] <= 80 and plan["V"] >= 36
    for k, v in view_outputs(gt, plan, 256).items():
        close(v, ot[k], f"teacher {k}", 5e-4, 5e-5)
    for k, v in view_outputs(got["outputs"], plan, 128).items():
        close(v, want["outputs"][k], f"student {k}", 5e-4, 5e-5)
    if task == "sap":
        st = PP.logit_stats(got["outputs"], want["outputs"])
        print("fp32 full-size:", json.dumps(st))
        for k in ("global_logits", "local_logits", "fused_logits"):
            assert st[k]["same_inf_mask"], k
            assert st[k]["max_abs_delta"] < FP32_LOGIT_TOL, (k, st[k])
            assert st[k]["argmax_agreement"] == 1.0, (k, st[k])
    elif task == "mlm":
        a, b = got["outputs"]["predict"].float().cpu(), want["outputs"]["predict"].float()
        assert a.shape[1] == 50265
        assert (a - b).abs().max().item() < FP32_LOGIT_TOL
        assert torch.equal(a.argmax(1), b.argmax(1))             # token indices bit-exact
    else:
        for a, b in zip(got["outputs"]["cfp"], want["outputs"]["cfp"]):
            close(a, b, "cfp outputs", 2e-4, 5e-5)
    close(got["supervised_loss"], want["supervised_loss"], "supervised loss", 2e-4, 1e-6)
    for k, v in want["kdl_terms"].items():
        close(got["kdl_terms"][k], v, f"kd term {k}", 5e-4, 1e-7)
    close(got["loss"], want["loss"], "total loss", 2e-4, 1e-6)
    params = dict(g_s.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in o_s.parameters() if p.grad is not None)
    n = 0
    for name, p in o_s.named_parameters():
        g = params[name].grad
        if p.grad is None:
            assert g.abs().max().item() == 0.0, name
            continue
        scale = p.grad.abs().max().item()
        close(g, p.grad, f"grad {name}", 3e-3, 2e-3 * scale + 4e-6 * gmax)
        n += 1
    assert n > 150


def test_bf16_engine_action_logits_within_stated_tolerance_of_oracle(models):
    """three full-size SAP batches: worst |delta logit| and argmax agreement of the benchmarked arithmetic vs the fp64 oracle"""
    st = PP.sap_parity(torch.bfloat16, batch_size=8, seeds=(1234, 77, 5), models=models)
    print("bf16 full-size:", json.dumps(st))
    assert st["same_inf_mask"]
    assert st["max_abs_logit_delta"] < BF16_LOGIT_TOL, st
    assert st["argmax_agreement"] >= BF16_ARGMAX_MIN, st
    assert st["worst_flip_gap"] <= 2.0 * st["max_abs_logit_delta"] + 1e-9, st          # flips only between candidates the oracle itself rates within 2 delta
    assert st["loss_rel_delta"] < 2e-2 and st["kdl_rel_delta"] < 3e-2, st


@pytest.mark.parametrize("task", ["sap", "mlm", "cfp"])
def test_bf16_engine_gradients_track_oracle_at_full_size(models, task):
    tcfg, scfg, o_t, o_s = models
    g_t, g_s = PP.engine_models(tcfg, scfg, o_t, o_s, torch.bfloat16)
    batch = synth.make_batch(task, batch_size=8, seed=99, step=1)
    _, want = PP.oracle_step(o_t, o_s, batch, task, backward=True)
    _, got = PP.engine_step(g_t, g_s, batch, task, backward=True)
    close(got["loss"], want["loss"], "bf16 total loss", 2e-2, 1e-3)
    params = dict(g_s.named_parameters())
    num = da = db = 0.0
    for name, p in o_s.named_parameters():
        if p.grad is None:
            continue
        g = params[name].grad.double().cpu()
        num += (g * p.grad).sum().item()
        da += (g * g).sum().item()
        db += (p.grad * p.grad).sum().item()
    cos = num / (da ** 0.5 * db ** 0.5)
    print(f"bf16 full-size gradient cosine vs oracle ({task}): {cos:.5f}")
    assert cos > 0.99, cos
    if task == "mlm":
        a, b = got["outputs"]["predict"].float().cpu(), want["outputs"]["predict"].float()
        agree = (a.argmax(1) == b.argmax(1)).float().mean().item()
        print(f"bf16 mlm token argmax agreement {agree:.4f}, max |delta| {(a - b).abs().max().item():.3e}")


def test_fp32_storage_with_split_bf16_contraction_meets_the_north_star_bar(models):
    """"bf16x3": fp32 activations / weights / epilogues, every GEMM contraction as three bf16 MFMAs on hi + lo halves of the fp32
    operands (lib.set_f32_mfma('bf16x3')).  Same full-size check as the exact-fp32 engine, to the NORTH-STAR bar itself (argmax
    identical, |delta logit| < 1e-3) -- at roughly half the step time of the exact fp32 MFMA (bench.py `modes`)."""
    from magic_amd.host import lib as L
    prev = L.set_f32_mfma("bf16x3")
    try:
        st = PP.sap_parity(torch.float32, batch_size=8, seeds=(1234, 77, 5), models=models)
    finally:
        L.set_f32_mfma(prev)
    print("bf16x3 full-size:", json.dumps(st))
    assert st["same_inf_mask"] and st["argmax_agreement"] == 1.0
    assert st["max_abs_logit_delta"] < 1e-3, st
    assert st["loss_rel_delta"] < 1e-3 and st["kdl_rel_delta"] < 1e-3, st
