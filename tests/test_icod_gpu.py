"""ICoD co-training direction (BASELINE config 3; agent.py:1026 role='s2t', :1136-1149): the TEACHER learns from the
student -- reverse MAKD with 'mean' reductions, the student's projection heads applied to the target side -- on the
HIP engine with a trainable teacher VLNBert, vs the oracle (nav_makd role='s2t', pinned to the reference by golden vectors)."""
from collections import defaultdict

import pytest
import torch
import torch.nn.functional as F

import magic_amd  # noqa: F401
from magic_amd.host import kd_loss as K
from magic_amd.host.config import make_config
from magic_amd.host.model_nav import VLNBert
from oracle import makd_ref as M
from oracle.nav_ref import RefVLNBert
from tests.test_nav_gpu import HEADS, nav_inputs, one_step, to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda"


def s2t_loss(teacher_res, student_out, heads, rw, inp, mse_fn=None, kd_fn=None):
    """t_total = a_t * (sum t_kdl * train_ml) + (1 - a_t) * t_ml  (agent.py:1138-1147), train_ml = 1, a_t = 0.5"""
    B = inp["txt_ids"].shape[0]
    s_detached = defaultdict(lambda: None)
    for k, v in student_out.items():
        s_detached[k] = {kk: vv.detach() for kk, vv in v.items()} if isinstance(v, dict) else (v.detach() if torch.is_tensor(v) else v)
    s_detached["sample_weights"] = M.exponential_decay(
        F.cross_entropy(student_out["nav_logits"].float().detach(), inp["targets"], reduction="none", ignore_index=-100), 0.7).detach()
    acc = defaultdict(float)
    M.nav_makd(0, teacher_res["out"], s_detached, heads, acc, role="s2t", temperature=2.0, weights=rw, weight_mode="RW", mse_fn=mse_fn, kd_fn=kd_fn)
    t_ml = teacher_res["ce"].sum() / B
    return 0.5 * sum(acc.values()) + 0.5 * t_ml, acc


def test_icod_reverse_distillation_trains_the_teacher():
    kw = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, num_l_layers=1, num_x_layers=1, num_pano_layers=1)
    tcfg, scfg = make_config(256, role="teacher", **kw), make_config(128, role="student", teacher_hidden_size=256, **kw)
    torch.manual_seed(0)
    o_t, o_s = RefVLNBert(tcfg).double().eval(), RefVLNBert(scfg).double().eval()
    args = type("A", (), dict(train_kdl_teacher=True, train_kdl=True))()
    g_t = VLNBert(args, role="teacher", config=tcfg, device=DEV, compute_dtype=torch.float32)
    g_s = VLNBert(None, role="student", config=scfg, device=DEV, compute_dtype=torch.float32)
    assert g_t.store.requires_grad
    g_t.load_state_dict(o_t.state_dict())
    g_s.load_state_dict(o_s.state_dict())
    inp = nav_inputs(B=3, L=10, seed=3)
    rw = [0.9, 1.1, 1.0, 1.2, 0.8]
    i64 = to_dev(inp, "cpu", f64=True)
    with torch.no_grad():
        so = one_step(o_s, i64)["out"]
    tr = one_step(o_t, i64)
    heads = {n: getattr(o_s.vln_bert, n) for n in HEADS}
    want, want_acc = s2t_loss(tr, so, heads, rw, i64)
    want.backward()
    idev = to_dev(inp, DEV)
    with torch.no_grad():
        gso = one_step(g_s, idev)["out"]
    g_t.store.zero_grad()
    gtr = one_step(g_t, idev)
    gheads = {n: getattr(g_s.vln_bert, n) for n in HEADS}
    got, got_acc = s2t_loss(gtr, gso, gheads, rw, idev, mse_fn=lambda a, b, w, lt: K.mse_loss(a, b, w, lt),
                            kd_fn=lambda s, t, T, w, lt: K.kd_loss(s, t, T, t_sample_weights=w, loss_type=lt))
    for k, v in want_acc.items():
        assert abs(float(got_acc[k]) - float(v)) <= 3e-4 * abs(float(v)) + 1e-7, k
    assert abs(float(got) - float(want)) <= 2e-4 * abs(float(want)) + 1e-6
    got.backward()
    torch.cuda.synchronize()
    params = dict(g_t.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in o_t.parameters() if p.grad is not None)
    n = 0
    for name, p in o_t.named_parameters():
        if p.grad is None:
            continue
        g = params[name].grad.float().cpu()
        assert torch.allclose(g, p.grad.float(), rtol=3e-3, atol=2e-3 * p.grad.abs().max().item() + 3e-6 * gmax), name
        n += 1
    assert n > 30
    # the student's heads are only applied to detached targets in this direction: no gradient reaches the student
    assert all(float(p.grad.abs().max()) == 0.0 for p in g_s.parameters() if p.grad is not None)
