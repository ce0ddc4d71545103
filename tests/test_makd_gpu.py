"""`GMapNavAgent.compute_kd_losses` on the HIP primitives (host/makd_nav.py over csrc/loss.hip) against the numbers the REFERENCE'S OWN METHOD
produced (tests/golden/makd_agent.pt, minted by tests/golden/mint_golden.py::mint_agent running map_nav_src/r2r/agent.py:546-719): every
weighting mode -- 'RW', none, and 'learned_weight' (softplus of the learner model's five kdl_*_weight scalars, the two image-embedding terms
halved on top) --, both directions (t2s 'sum' / 'mean', ICoD's s2t), t = 0 and t > 0, heads 2 vs 4 (-m gpu)."""
import os
from collections import defaultdict

import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import ops as O
from magic_amd.host.makd_nav import LEARNED_NAMES, LOSS_KEYS, compute_kd_losses
from oracle import makd_ref as M

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _Head:
    """nn.Linear(H_s, H_t) of the fixture on the HIP GEMM (fp32 storage, exact-fp32 MFMA), autograd through torch for the input side"""

    def __init__(self, w, b):
        self.w, self.b = w.to(DEV).contiguous(), b.to(DEV).contiguous()

    def __call__(self, x):
        shp = x.shape
        M_ = x.numel() // shp[-1]
        y = O.linear_fwd(x.detach().reshape(M_, shp[-1]).contiguous(), self.w, self.b, M_)
        return y.view(*shp[:-1], self.w.shape[0])


def _dev(o):
    out = {}
    for k, v in o.items():
        out[k] = {kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else (v.to(DEV) if torch.is_tensor(v) else v)
    return out


def _fixture(golden_dir):
    fx = torch.load(os.path.join(golden_dir, "makd_agent.pt"), weights_only=False)
    heads = {n: _Head(w, b) for n, (w, b) in fx["heads"].items()}
    return fx, heads, _dev(fx["s_out"]), _dev(fx["t_out"])


def test_compute_kd_losses_on_hip_matches_the_reference_method_in_every_weighting_mode(golden_dir):
    fx, heads, s_out, t_out = _fixture(golden_dir)
    rw = fx["rw"].to(DEV)
    lw_s = {n: v.to(DEV) for n, v in fx["learned_student"].items()}
    lw_t = {n: v.to(DEV) for n, v in fx["learned_teacher"].items()}
    seen = set()
    for name, want in fx["cases"].items():
        parts = name.split("_")
        role, t = parts[0], int(parts[1][1:])
        mode = "learned_weight" if "learned" in name else (None if parts[2] == "None" else "RW")
        seen.add((role, mode))
        acc = defaultdict(float)
        kw = dict(temperature=2.0, weights=rw if mode == "RW" else None)
        if role == "t2s":
            got = compute_kd_losses(t, s_out, t_out, heads, acc, role="t2s", loss_type=parts[-1], learned=lw_s if mode == "learned_weight" else None, **kw)
        else:       # ICoD reverse: the teacher is the learner (its kdl_*_weight scale the terms), the student's heads project the target side
            got = compute_kd_losses(t, t_out, s_out, heads, acc, role="s2t", learned=lw_t if mode == "learned_weight" else None, **kw)
        assert set(got) == set(want) and set(got) <= set(LOSS_KEYS), name
        for k in want:
            torch.testing.assert_close(torch.as_tensor(float(got[k])), want[k], rtol=3e-5, atol=2e-6, msg=f"{name}:{k}")
    assert seen == {("t2s", "RW"), ("t2s", None), ("t2s", "learned_weight"), ("s2t", "RW"), ("s2t", "learned_weight")}


def test_learned_ability_weights_receive_the_gradient_the_oracle_autograd_gives(golden_dir):
    """d(sum of the ten terms) / d(kdl_*_weight): the softplus runs in torch on the model's own scalars, the loss values come from the fused
    HIP loss kernels -- against plain autograd through the oracle restatement (pinned to the same fixture by tests/test_oracle_golden.py)"""
    fx, heads, s_out, t_out = _fixture(golden_dir)
    lw = {n: fx["learned_student"][n].clone().to(DEV).requires_grad_(True) for n in LEARNED_NAMES}
    acc = defaultdict(float)
    for t in (0, 1):
        acc = compute_kd_losses(t, s_out, t_out, heads, acc, role="t2s", learned=lw)
    total = sum(acc.values())
    total.sum().backward()
    ref_heads = {n: (lambda x, w=w, b=b: torch.nn.functional.linear(x, w, b)) for n, (w, b) in fx["heads"].items()}
    lw_ref = {n: fx["learned_student"][n].clone().double().requires_grad_(True) for n in LEARNED_NAMES}
    dbl = lambda o: {k: ({kk: vv.double() for kk, vv in v.items()} if isinstance(v, dict) else (v.double() if torch.is_tensor(v) else v)) for k, v in o.items()}
    ref_heads = {n: (lambda x, w=w.double(), b=b.double(): torch.nn.functional.linear(x, w, b)) for n, (w, b) in fx["heads"].items()}
    acc_r = defaultdict(float)
    for t in (0, 1):
        acc_r = M.nav_makd(t, dbl(fx["s_out"]), dbl(fx["t_out"]), ref_heads, acc_r, role="t2s", weight_mode="learned_weight", learned=lw_ref)
    sum(acc_r.values()).sum().backward()
    for n in LEARNED_NAMES:
        torch.testing.assert_close(lw[n].grad.cpu().double(), lw_ref[n].grad, rtol=5e-5, atol=1e-6, msg=n)
        assert lw_ref[n].grad.abs().item() > 0


def test_fused_step_form_matches_the_reference_method_and_the_per_term_gradients(golden_dir):
    """makd_nav.compute_kd_losses_fused (the step's nine mse terms in ONE launch / ONE autograd node, running sums as a vector -- what the rollout
    loop calls) against the same golden numbers of the reference's own method ('RW' and unweighted, both directions, t = 0 and t > 0), and its
    input gradients against the per-term form's."""
    from magic_amd.host.makd_nav import compute_kd_losses_fused, kd_terms
    fx, heads, s_out, t_out = _fixture(golden_dir)
    rw = fx["rw"].to(DEV)
    n = 0
    for name, want in fx["cases"].items():
        parts = name.split("_")
        role, t = parts[0], int(parts[1][1:])
        if "learned" in name or (role == "t2s" and parts[-1] != "sum"):
            continue                       # (the fused form serves the rollout loop's configuration: MKRW or no ability weights, t2s 'sum' / s2t 'mean')
        weights = None if parts[2] == "None" else rw
        a, b = (s_out, t_out) if role == "t2s" else (t_out, s_out)
        got = kd_terms(compute_kd_losses_fused(t, a, b, heads, {}, role=role, temperature=2.0, weights=weights))
        for k in want:
            torch.testing.assert_close(torch.as_tensor(float(got[k])), want[k], rtol=3e-5, atol=2e-6, msg=f"{name}:{k}")
        for k in set(LOSS_KEYS) - set(want):
            assert float(got[k]) == 0.0, (name, k)
        n += 1
    assert n >= 4

    def grads(fn):
        s2 = {k: ({kk: vv.clone().requires_grad_(vv.is_floating_point()) for kk, vv in v.items()} if isinstance(v, dict)
                  else (v.clone().requires_grad_(True) if torch.is_tensor(v) and v.is_floating_point() else v)) for k, v in s_out.items()}
        acc = fn(s2)
        sum(kd_terms(acc).values()).backward()
        out = {}
        for k, v in s2.items():
            for kk, vv in (v.items() if isinstance(v, dict) else [(None, v)]):
                if torch.is_tensor(vv) and vv.grad is not None:
                    out[(k, kk)] = vv.grad
        return out
    id_heads = {n_: (lambda x: x) for n_ in heads}              # (identity heads: the fixture's head wrapper detaches its input)
    same_w = {k: v for k, v in t_out.items()}
    ga = grads(lambda s2: compute_kd_losses(1, s2, same_w, id_heads, defaultdict(float), role="s2t", temperature=2.0, weights=rw))
    gb = grads(lambda s2: compute_kd_losses_fused(1, s2, same_w, id_heads, {}, role="s2t", temperature=2.0, weights=rw))
    assert set(ga) == set(gb) and len(ga) >= 6
    for k in ga:
        torch.testing.assert_close(gb[k], ga[k], rtol=1e-5, atol=1e-8, msg=str(k))
