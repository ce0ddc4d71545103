"""MRC proxy task (masked region classification; MrcDataset / mrc_collate tasks.py:189-310, validate_mrc
train_r2r_magic.py:476-500) on the HIP engine vs the fp64 oracle (-m gpu).  The shipped pretrain JSON trains mlm/sap/cfp;
MRC is reachable by listing it in `tasks` (train_r2r_magic.py:50-52), which also creates the `image_classifier` head."""
import pytest
import torch
import torch.nn.functional as F

import magic_amd  # noqa: F401
from magic_amd.host import ops as O
from magic_amd.host import synth
from magic_amd.host.config import make_config
from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining
from oracle import model_ref as R
from tests.test_model_gpu import KDL, RW, close, to64

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(dtype):
    kw = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=600, num_l_layers=2, num_x_layers=1,
              num_pano_layers=1, pretrain_tasks={"mlm", "mrc", "sap"})
    tcfg = make_config(256, role="teacher", **kw)
    scfg = make_config(128, role="student", teacher_hidden_size=256, kdl=KDL, **kw)
    torch.manual_seed(0)
    o_t, o_s = R.RefPretrainModel(tcfg).eval(), R.RefPretrainModel(scfg).eval()
    with torch.no_grad():
        for m in (o_t, o_s):
            for n, p in m.named_parameters():
                if n.endswith("bias"):
                    p.normal_(0, 0.02)
    mk = lambda cfg, o: GlocalTextPathCMTPreTraining.from_pretrained(None, config=cfg, state_dict=o.state_dict(), device=DEV, compute_dtype=dtype)
    return o_t, o_s, mk(tcfg, o_t), mk(scfg, o_s)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_softkl_rows_kernel(dtype):
    g = torch.Generator().manual_seed(1)
    M, N = 37, 1000
    x = (torch.randn(M, N, generator=g) * 2).to(DEV).to(dtype)
    t = torch.softmax(torch.randn(M, N, generator=g) * 2, -1).to(DEV)
    t[3, :500] = 0                                    # zeros in the target contribute nothing (xlogy convention)
    t[5] *= 0.5                                       # a row that does not sum to one
    xr = x.float().clone().requires_grad_(True)
    want = F.kl_div(F.log_softmax(xr, -1), t, reduction="none").sum(1)
    (want.sum() * 0.37).backward()
    rows, d = torch.empty(M, device=DEV), torch.empty(M, N, dtype=dtype, device=DEV)
    O.softkl_rows(x, M, N, N, t, coef=0.37, loss_row=rows, dlogits=d, ldd=N)
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-3)
    assert torch.allclose(rows, want.detach(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(d.float(), xr.grad, **tol)


def test_mrc_fp32_forward_loss_gradients_match_oracle_and_validate_contract():
    o_t, o_s, g_t, g_s = build(torch.float32)
    assert "image_classifier.net.3.weight" in dict(g_s.named_parameters())
    batch = synth.make_batch("mrc", batch_size=5, seed=8, vocab=600, min_len=8, max_len=19, min_steps=2, max_steps=4)
    o_t, o_s = o_t.double(), o_s.double()
    b64 = to64(batch)
    rw = torch.tensor(RW, dtype=torch.float64)
    # --- validate_mrc contract: (view_logits, view_targets, None, None)
    with torch.no_grad():
        vl, vt, ol, ot_ = g_s(batch, task="mrc", compute_loss=False)
        wl, wt, _, _ = o_s(b64, "mrc", compute_loss=False)
    assert ol is None and ot_ is None
    n_feat = int(batch["vp_view_mrc_masks"].sum())
    assert vl.shape == (n_feat, 1000) and vt.shape == (n_feat, 1000)
    close(vl, wl, "view logits", 1e-4, 2e-5)
    close(vt, wt, "view targets", 0, 0)
    kl_got = F.kl_div(F.log_softmax(vl.float(), -1), vt, reduction="sum").item()            # train_r2r_magic.py:484-485
    kl_want = F.kl_div(F.log_softmax(wl, -1), wt, reduction="sum").item()
    assert abs(kl_got - kl_want) < 1e-3 * max(1.0, abs(kl_want))
    assert torch.equal(vl.argmax(-1).cpu(), wl.argmax(-1))                                   # accuracy numerator, bit-exact
    # --- training step with MAKD (local / txt / img abilities; no map branch in this task)
    with torch.no_grad():
        gt = g_t(batch, "mrc", compute_loss=False, return_outputs=True)
        ot = o_t(b64, "mrc", compute_loss=True)["outputs"]
    assert "gmap_embeds" not in gt
    want = o_s(b64, "mrc", compute_loss=True, teacher_outputs=ot, rw=rw)
    want["loss"].backward()
    g_s.store.zero_grad()
    got = g_s(batch, "mrc", compute_loss=True, teacher_outputs=gt, rw=RW, plan=gt["plan"])
    close(got["supervised_loss"], want["supervised_loss"], "mrc loss", 1e-4, 1e-6)
    assert set(k for k, v in want["kdl_terms"].items()) <= set(got["kdl_terms"])
    for k, v in want["kdl_terms"].items():
        close(got["kdl_terms"][k], v, f"kd term {k}", 2e-4, 1e-7)
    for k in ("global_emb_loss", "global_attn_loss", "predict_loss"):
        assert float(got["kdl_terms"][k]) == 0.0
    close(got["loss"], want["loss"], "total loss", 1e-4, 1e-6)
    got["loss"].backward()
    torch.cuda.synchronize()
    params = dict(g_s.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in o_s.parameters() if p.grad is not None)
    n = 0
    for name, p in o_s.named_parameters():
        g = params[name].grad
        if p.grad is None:
            assert g.abs().max().item() == 0.0, name
            continue
        close(g, p.grad, f"grad {name}", 2e-3, 2e-4 * p.grad.abs().max().item() + 2e-6 * gmax)
        n += 1
    assert n > 50 and params["image_classifier.net.3.weight"].grad.abs().max() > 0
    assert params["bert.global_encoder.encoder.crossattention.0.attention.self.query.weight"].grad.abs().max() == 0    # unused branch


def test_mrc_bf16_tracks_fp32():
    _, _, t32, s32 = build(torch.float32)
    _, _, t16, s16 = build(torch.bfloat16)
    batch = synth.make_batch("mrc", batch_size=4, seed=9, vocab=600, min_len=8, max_len=19, min_steps=2, max_steps=4)
    outs = []
    for t, s in ((t32, s32), (t16, s16)):
        with torch.no_grad():
            gt = t(batch, "mrc", compute_loss=False, return_outputs=True)
        s.store.zero_grad()
        o = s(batch, "mrc", compute_loss=True, teacher_outputs=gt, rw=RW, plan=gt["plan"])
        o["loss"].backward()
        outs.append((float(o["loss"]), float(o["supervised_loss"]), s.store.grad.clone()))
    assert abs(outs[0][0] - outs[1][0]) < 2e-2 * abs(outs[0][0])
    assert abs(outs[0][1] - outs[1][1]) < 2e-2 * abs(outs[0][1])
    cos = F.cosine_similarity(outs[0][2], outs[1][2], dim=0).item()
    assert cos > 0.98, cos
