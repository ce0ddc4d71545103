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


def test_icod_at_stated_size_magic_l_teacher_and_magic_s_student():
    """BASELINE config 3 at its stated size: MAGIC-L teacher (H = 768, 12 heads, FFN 3072) and MAGIC-S student (H = 128, 2 heads), both
    at full depth (6 text / 3 cross-modal / 2 panorama layers), co-trained for one navigator step in BOTH directions -- t2s MAKD for
    the student (projection heads 128 -> 768, attention maps sliced to min(2, 12) heads, agent.py:560-579) and reverse s2t for the
    trainable teacher.  fp32 engine vs fp64 oracle: the ten MAKD terms of each direction, both losses, every parameter gradient of
    both models; then the same step in bf16 (the benchmarked arithmetic of bench_nav.py --icod) must track it."""
    kw = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300)
    tcfg, scfg = make_config(768, role="teacher", **kw), make_config(128, role="student", teacher_hidden_size=768, **kw)
    assert (tcfg.num_attention_heads, scfg.num_attention_heads, tcfg.num_l_layers, tcfg.num_x_layers, tcfg.num_pano_layers) == (12, 2, 6, 3, 2)
    torch.manual_seed(0)
    o_t, o_s = RefVLNBert(tcfg).double().eval(), RefVLNBert(scfg).double().eval()
    args = type("A", (), dict(train_kdl_teacher=True, train_kdl=True))()
    inp = nav_inputs(B=3, L=14, seed=5)
    rw = [0.9, 1.1, 1.0, 1.2, 0.8]
    i64 = to_dev(inp, "cpu", f64=True)
    heads = {n: getattr(o_s.vln_bert, n) for n in HEADS}
    # oracle: student step against the (detached) teacher, teacher step against the (detached) student
    tr = one_step(o_t, i64)
    t_det = defaultdict(lambda: None)
    for k, v in tr["out"].items():
        t_det[k] = {kk: vv.detach() for kk, vv in v.items()} if isinstance(v, dict) else (v.detach() if torch.is_tensor(v) else v)
    sr = one_step(o_s, i64, heads=heads, rw=rw, teacher_out=t_det)
    want_t, want_t_acc = s2t_loss(tr, sr["out"], heads, rw, i64)
    sr["loss"].backward()
    want_t.backward()
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        g_t = VLNBert(args, role="teacher", config=tcfg, device=DEV, compute_dtype=dtype)
        g_s = VLNBert(None, role="student", config=scfg, device=DEV, compute_dtype=dtype)
        g_t.load_state_dict(o_t.state_dict())
        g_s.load_state_dict(o_s.state_dict())
        g_t.eval(); g_s.eval()
        idev = to_dev(inp, DEV)
        gheads = {n: getattr(g_s.vln_bert, n) for n in HEADS}
        mse_fn = lambda a, b, w, lt: K.mse_loss(a, b, w, lt)
        kd_fn = lambda s_, t_, T, w, lt: K.kd_loss(s_, t_, T, t_sample_weights=w, loss_type=lt)
        g_t.store.zero_grad(); g_s.store.zero_grad()
        gtr = one_step(g_t, idev)
        gt_det = defaultdict(lambda: None)
        for k, v in gtr["out"].items():
            gt_det[k] = {kk: vv.detach() for kk, vv in v.items()} if isinstance(v, dict) else (v.detach() if torch.is_tensor(v) else v)
        gsr = one_step(g_s, idev, heads=gheads, rw=rw, teacher_out=gt_det, mse_fn=mse_fn, kd_fn=kd_fn)
        got_t, got_t_acc = s2t_loss(gtr, gsr["out"], gheads, rw, idev, mse_fn=mse_fn, kd_fn=kd_fn)
        gsr["loss"].backward()
        got_t.backward()
        torch.cuda.synchronize()
        res[dtype] = (float(gsr["loss"]), float(got_t), g_s.store.grad.clone().float(), g_t.store.grad.clone().float())
        if dtype == torch.float32:
            for k, v in sr["kd"].items():
                assert abs(float(gsr["kd"][k]) - float(v)) <= 5e-4 * abs(float(v)) + 1e-7, ("t2s", k)
            for k, v in want_t_acc.items():
                assert abs(float(got_t_acc[k]) - float(v)) <= 5e-4 * abs(float(v)) + 1e-7, ("s2t", k)
            assert abs(float(gsr["loss"]) - float(sr["loss"])) <= 3e-4 * abs(float(sr["loss"]))
            assert abs(float(got_t) - float(want_t)) <= 3e-4 * abs(float(want_t))
            a, b = gsr["out"]["nav_logits"].cpu(), sr["out"]["nav_logits"]
            assert torch.equal(a.argmax(1), b.argmax(1)) and (torch.nan_to_num(a, neginf=0) - torch.nan_to_num(b, neginf=0).float()).abs().max() < 1e-3
            for g_m, o_m, nm in ((g_s, o_s, "student"), (g_t, o_t, "teacher")):
                params = dict(g_m.named_parameters())
                gmax = max(p.grad.abs().max().item() for p in o_m.parameters() if p.grad is not None)
                n = 0
                for name, p in o_m.named_parameters():
                    if p.grad is None:
                        continue
                    g = params[name].grad.float().cpu()
                    assert torch.allclose(g, p.grad.float(), rtol=5e-3, atol=3e-3 * p.grad.abs().max().item() + 5e-6 * gmax), (nm, name)
                    n += 1
                assert n > 150, (nm, n)
        del g_t, g_s
    (ls32, lt32, gs32, gt32), (ls16, lt16, gs16, gt16) = res[torch.float32], res[torch.bfloat16]
    assert abs(ls16 - ls32) <= 3e-2 * abs(ls32) and abs(lt16 - lt32) <= 3e-2 * abs(lt32), (ls16, ls32, lt16, lt32)
    assert F.cosine_similarity(gs16, gs32, dim=0).item() > 0.98 and F.cosine_similarity(gt16, gt32, dim=0).item() > 0.98
