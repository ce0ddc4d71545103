"""Oracle model: block-level second opinion (HF BertLayer) + end-to-end contract on synthetic batches."""
import copy

import pytest
import torch

import magic_amd  # noqa: F401
from magic_amd.host import synth
from magic_amd.host.config import make_config
from oracle import model_ref as R

KDL = dict(knowledge_distillation=True, kd_alpha=0.5, kd_temperature=2, teacher_sample_hard_mining=True,
           t_sample_preprocess_exp_decay=0.7, rw_temp=4,
           kdl_tasks=["txt", "img", "local", "global", "predict"], kdl_task_types=["emb", "attn"])


def small_cfgs(vocab=1000):
    tasks = {"mlm", "mrc", "sap", "cfp"}
    t = make_config(128, role="teacher", vocab_size=vocab, num_l_layers=2, num_x_layers=1, num_pano_layers=1, pretrain_tasks=tasks)
    s = make_config(64, role="student", teacher_hidden_size=128, vocab_size=vocab, num_l_layers=2, num_x_layers=1,
                    num_pano_layers=1, kdl=KDL, pretrain_tasks=tasks)
    return t, s


def test_self_layer_matches_hf_bertlayer():
    tr = pytest.importorskip("transformers")
    from transformers.models.bert.modeling_bert import BertConfig, BertLayer
    cfg = make_config(128)
    ours = R.RefSelfLayer(cfg).eval()
    hf_cfg = BertConfig(hidden_size=128, num_attention_heads=2, intermediate_size=512, layer_norm_eps=1e-12,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hf_cfg._attn_implementation = "eager"
    hf = BertLayer(hf_cfg).eval()
    missing = hf.load_state_dict(ours.state_dict(), strict=True)
    x = torch.randn(2, 20, 128)
    mask = torch.ones(2, 20, dtype=torch.bool)
    mask[1, 13:] = False
    kb = R.key_bias(mask)
    want = hf(x, attention_mask=kb)
    want = want[0] if isinstance(want, tuple) else want
    got, _ = ours(x, kb)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("task", ["mlm", "mrc", "sap", "cfp"])
def test_oracle_forward_contract(task):
    torch.manual_seed(0)
    tcfg, scfg = small_cfgs()
    teacher, student = R.RefPretrainModel(tcfg).eval(), R.RefPretrainModel(scfg).eval()
    batch = synth.make_batch(task, batch_size=4, seed=3, vocab=1000, min_len=6, max_len=14, min_steps=2, max_steps=4)
    with torch.no_grad():
        t_out = teacher(batch, task, compute_loss=True)["outputs"]
    out = student(batch, task, compute_loss=True, teacher_outputs=t_out, rw=torch.ones(5))
    assert torch.isfinite(out["loss"])
    out["loss"].backward()
    n_grad = sum(p.grad is not None for p in student.parameters())
    assert n_grad > 20
    inf = student(batch, task, compute_loss=False)
    if task == "mlm":
        assert inf["predict"].shape == (int((batch["txt_labels"] != -1).sum()), 1000)
    elif task == "mrc":
        n = int(batch["vp_view_mrc_masks"].sum())
        assert inf[0].shape == (n, 1000) and inf[1].shape == (n, 1000) and inf[2] is None and inf[3] is None
        assert torch.allclose(inf[1].sum(1), torch.ones(n), atol=1e-5)
    elif task == "sap":
        B, K = batch["gmap_step_ids"].shape
        assert inf["fused_logits"].shape == (B, K)
        # masked entries are -inf, argmax lands on an allowed node
        am = inf["fused_logits"].argmax(1)
        assert not batch["gmap_visited_masks"][torch.arange(B), am].any()
    else:
        assert len(inf) == 4 and inf[0].shape == (4, 64)
