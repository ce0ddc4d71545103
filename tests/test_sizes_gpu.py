"""Wider configurations of the same path (-m gpu): the B/L model sizes (H = 384 / 768: 6 / 12 heads, NIT 3 / 6, the
LayerNorm-fused GEMM at H=384 and its fallback at H=768, head slicing min(h_s, h_t) in the attention distillation) and
RxR-length instructions (> 128 tokens: the fused attention kernels report 'unsupported' and the engine takes the
GEMM + softmax path).  fp32 engine vs fp64 oracle, same bars as tests/test_model_gpu.py."""
import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import synth
from magic_amd.host.config import make_config
from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining
from oracle import model_ref as R
from tests.test_model_gpu import KDL, RW, close, to64, view_outputs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def run_case(hs, ht, task, batch, layers=(1, 1, 1), vocab=400, **over):
    kw = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=vocab, num_l_layers=layers[0], num_x_layers=layers[1], num_pano_layers=layers[2])
    kw.update(over)
    tcfg = make_config(ht, role="teacher", **kw)
    scfg = make_config(hs, role="student", teacher_hidden_size=ht, kdl=KDL, **kw)
    torch.manual_seed(1)
    o_t, o_s = R.RefPretrainModel(tcfg).eval(), R.RefPretrainModel(scfg).eval()
    g_t = GlocalTextPathCMTPreTraining.from_pretrained(None, config=tcfg, state_dict=o_t.state_dict(), device=DEV, compute_dtype=torch.float32)
    g_s = GlocalTextPathCMTPreTraining.from_pretrained(None, config=scfg, state_dict=o_s.state_dict(), device=DEV, compute_dtype=torch.float32)
    g_s.keep_mlm_logits = True
    o_t, o_s = o_t.double(), o_s.double()
    b64 = to64(batch)
    with torch.no_grad():
        ot = o_t(b64, task)["outputs"]
    want = o_s(b64, task, teacher_outputs=ot, rw=torch.tensor(RW, dtype=torch.float64))
    want["loss"].backward()
    with torch.no_grad():
        gt = g_t(batch, task, compute_loss=False, return_outputs=True)
    g_s.store.zero_grad()
    got = g_s(batch, task, compute_loss=True, teacher_outputs=gt, rw=RW, plan=gt["plan"])
    for k, v in view_outputs(got["outputs"], gt["plan"], hs).items():
        close(v, want["outputs"][k], f"student {k}", 3e-4, 3e-5)
    for k, v in want["kdl_terms"].items():
        close(got["kdl_terms"][k], v, f"kd {k}", 3e-4, 1e-7)
    close(got["loss"], want["loss"], "loss", 2e-4, 1e-6)
    if task == "sap":
        a, b = got["outputs"]["fused_logits"].cpu(), want["outputs"]["fused_logits"]
        assert torch.equal(a.argmax(1), b.argmax(1))
    got["loss"].backward()
    torch.cuda.synchronize()
    params = dict(g_s.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in o_s.parameters() if p.grad is not None)
    for name, p in o_s.named_parameters():
        if p.grad is not None:
            close(params[name].grad, p.grad, f"grad {name}", 3e-3, 2e-3 * p.grad.abs().max().item() + 3e-6 * gmax)


def test_magic_b_student_with_magic_l_teacher():
    batch = synth.make_batch("sap", batch_size=3, seed=9, vocab=400, min_len=6, max_len=12, min_steps=2, max_steps=3)
    run_case(384, 768, "sap", batch)


def test_rxr_length_instructions_take_the_unfused_attention_path():
    batch = synth.make_batch("sap", batch_size=2, seed=4, vocab=400, min_len=140, max_len=170, min_steps=2, max_steps=3)
    assert batch["txt_ids"].shape[1] > 128
    run_case(128, 256, "sap", batch)
    batch = synth.make_batch("mlm", batch_size=2, seed=5, vocab=400, min_len=140, max_len=170, min_steps=2, max_steps=3)
    run_case(128, 256, "mlm", batch)


@pytest.mark.parametrize("task", ["sap", "mlm", "cfp"])
def test_edge_shapes_single_sample_single_step_and_ragged_view_counts(task):
    """B = 1 with one-step trajectories (the map is [stop] + one visited node + its candidates), and batches whose panoramas
    have 36 or 37 view tokens (two candidates seen in one discretised view, dataset.py:742-756) -- padded per batch by the
    collate and masked through traj_vp_view_lens / vp_lens."""
    b1 = synth.make_batch(task, batch_size=1, seed=3, vocab=400, min_len=5, max_len=5, min_steps=1, max_steps=1)
    assert len(b1["traj_step_lens"]) == 1 and b1["traj_step_lens"][0] == 1
    run_case(128, 256, task, b1)
    br = synth.make_batch(task, batch_size=4, seed=6, vocab=400, min_len=6, max_len=14, min_steps=2, max_steps=4, dup_view_prob=0.5)
    lens = br["traj_vp_view_lens"].tolist()
    assert br["traj_view_img_fts"].shape[1] == 37 and 36 in lens and 37 in lens, lens
    run_case(128, 256, task, br)


def test_maximum_instruction_length_512_tokens():
    """the hard cap of the text encoder: 512 tokens = max_position_embeddings 514 - 2 (RoBERTa offset); rows of 512 keys are
    the widest the masked-softmax kernel serves in one wave"""
    batch = synth.make_batch("sap", batch_size=1, seed=12, vocab=400, min_len=512, max_len=512, min_steps=2, max_steps=2)
    assert batch["txt_ids"].shape[1] == 512
    run_case(128, 256, "sap", batch)


@pytest.mark.parametrize("task", ["sap", "cfp"])
def test_masked_mean_panorama_fusion(task):
    """adaptive_pano_fusion = false (r2r_magic_model_config.json:57 switched off): the visited-node embedding is the masked mean of the
    view embeddings; ragged view counts so the mask matters; the fusion's scoring parameters must receive no gradient"""
    b = synth.make_batch(task, batch_size=4, seed=8, vocab=400, min_len=6, max_len=14, min_steps=2, max_steps=4, dup_view_prob=0.5)
    run_case(128, 256, task, b, adaptive_pano_fusion=False)
