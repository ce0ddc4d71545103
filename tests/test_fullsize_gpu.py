"""BASELINE-size checks (B = 48, <= 80 tokens, 36 views x 768, vocab 50265, MAGIC-S student + H=256 teacher) where the fp64
oracle is too slow to be the checker: size-independent properties of the path (-m gpu).
  * bf16 (the benchmarked arithmetic) against the fp32 engine, which the small-size tests pin to the oracle;
  * MKRW linearity: every distillation term is linear in its ability weight, the supervised loss does not depend on them;
  * task-unused parameters receive exactly zero gradient (what replaces DDP's find_unused_parameters);
  * replaying a captured step is deterministic given the dropout seed, and the graph step equals the eager step."""
import pytest
import torch
import torch.nn.functional as F

import magic_amd  # noqa: F401
from magic_amd.host import synth
from magic_amd.host.config import make_config
from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining, KD_SLOTS
from magic_amd.host.plan import build_plan
from tests.test_model_gpu import KDL

pytestmark = pytest.mark.gpu
DEV = "cuda"
RW = [1.3, 0.7, 1.1, 0.9, 1.0]


def models(dtype, p_drop=0.0):
    kw = dict(hidden_dropout_prob=p_drop, attention_probs_dropout_prob=p_drop)
    t = GlocalTextPathCMTPreTraining(make_config(256, role="teacher", **kw), device=DEV, compute_dtype=dtype, seed=0)
    s = GlocalTextPathCMTPreTraining(make_config(128, role="student", teacher_hidden_size=256, kdl=KDL, **kw), device=DEV, compute_dtype=dtype, seed=1)
    return t, s


def step(t, s, batch, task, plan, rw=RW):
    with torch.no_grad():
        gt = t(batch, task, compute_loss=False, return_outputs=True, plan=plan)
    s.store.zero_grad()
    out = s(batch, task, compute_loss=True, teacher_outputs=gt, rw=rw, plan=plan)
    s.backward()
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("task", ["sap", "mlm", "cfp"])
def test_bf16_tracks_fp32_at_benchmark_size(task):
    batch = synth.make_batch(task, batch_size=48, seed=1234, step=0)
    plan = build_plan(batch, task, torch.device(DEV))
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        t, s = models(dtype)
        s.keep_mlm_logits = True
        out = step(t, s, batch, task, plan)
        res[dtype] = (out, s.store.grad.clone())
        del t, s
    (o32, g32), (o16, g16) = res[torch.float32], res[torch.bfloat16]
    assert torch.isfinite(g16).all() and torch.isfinite(g32).all()
    for k in ("loss", "supervised_loss", "kdl_loss"):
        a, b = float(o32[k].detach()), float(o16[k].detach())
        assert abs(a - b) <= 2e-2 * abs(a) + 1e-4, (k, a, b)
    assert F.cosine_similarity(g32, g16, dim=0).item() > 0.99
    if task == "sap":
        for k in ("global_logits", "local_logits", "fused_logits"):
            a, b = o32["outputs"][k], o16["outputs"][k]
            assert torch.equal(torch.isinf(a), torch.isinf(b))
            agree = (a.argmax(1) == b.argmax(1)).float().mean().item()
            assert agree >= 0.95, (k, agree)              # random-init logits are near-ties; bf16 may flip a few
            d = (torch.nan_to_num(a, neginf=0) - torch.nan_to_num(b, neginf=0)).abs().max().item()
            assert d < 3e-2, (k, d)
    elif task == "mlm":
        a, b = o32["outputs"]["predict"].float(), o16["outputs"]["predict"].float()
        assert a.shape[1] == 50265 and (a - b).abs().max().item() < 5e-2


def test_mkrw_linearity_and_unused_parameter_gradients():
    task = "sap"
    batch = synth.make_batch(task, batch_size=48, seed=1234, step=1)
    plan = build_plan(batch, task, torch.device(DEV))
    t, s = models(torch.float32)
    base = step(t, s, batch, task, plan, rw=[1.0] * 5)
    g_base = s.store.grad.clone()
    scaled = step(t, s, batch, task, plan, rw=[2.0, 3.0, 0.5, 4.0, 1.5])
    factor = dict(txt=2.0, img=3.0, avg_img=3.0, **{"global": 0.5}, local=4.0, predict=1.5)
    for k in KD_SLOTS:
        f = factor[k.split("_")[0] if not k.startswith("avg") else "avg_img"]
        a, b = float(base["kdl_terms"][k]), float(scaled["kdl_terms"][k])
        assert abs(b - f * a) <= 1e-5 * max(1.0, abs(f * a)), (k, a, b, f)
    assert abs(float(base["supervised_loss"]) - float(scaled["supervised_loss"])) < 1e-6
    # task-unused heads: exactly zero gradient under SAP (MLM transform / decoder bias, CFP heads)
    params = dict(s.named_parameters())
    for name in ("mlm_head.predictions.transform.dense.weight", "mlm_head.predictions.bias", "cfp_heads.txt.weight", "cfp_heads.gmap.weight"):
        assert params[name].grad.abs().max().item() == 0.0, name
    used = params["global_sap_head.net.0.weight"].grad
    assert used.abs().max().item() > 0
    # word embeddings: only rows of tokens present in the batch receive gradient (the tied decoder is idle under SAP)
    wg = params["bert.embeddings.word_embeddings.weight"].grad
    present = torch.zeros(wg.shape[0], dtype=torch.bool, device=DEV)
    present[batch["txt_ids"].to(DEV).unique()] = True
    assert wg[~present].abs().max().item() == 0.0 and wg[present].abs().sum().item() > 0
    assert not torch.equal(g_base, s.store.grad)


def test_captured_step_equals_eager_step_at_benchmark_size():
    from magic_amd.host.trainer import PretrainStep
    task = "sap"
    batch = synth.batch_to(synth.make_batch(task, batch_size=48, seed=1234, step=2), torch.device(DEV))
    cpu_batch = synth.make_batch(task, batch_size=48, seed=1234, step=2)
    plan = build_plan(cpu_batch, task, torch.device(DEV))
    finals = []
    for mode in ("eager", "graph"):
        t, s = models(torch.bfloat16)                         # dropout 0: the two paths must then agree bit for bit... up to atomics
        tr = PretrainStep(s, t, warmup_steps=10, num_train_steps=100)
        rw = torch.tensor(RW, device=DEV)
        if mode == "eager":
            for _ in range(3):
                tr.step(batch, task, rw=rw, plan=plan)
        else:
            cs = tr.capture(batch, task, plan, rw=rw)
            for _ in range(3):
                tr.replay(cs)
        torch.cuda.synchronize()
        finals.append(s.store.flat.clone())
        del t, s, tr
    a, b = finals
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    # fp32 atomics reorder sums between runs: compare to the size of three Adam updates (lr 5e-5 * warm-up fraction)
    assert (a - b).abs().max().item() < 2e-5, (a - b).abs().max().item()
    assert F.cosine_similarity(a, b, dim=0).item() > 0.999999
