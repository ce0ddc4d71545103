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


def test_text_branch_matches_hf_roberta_model():
    """The reference initialises `bert.embeddings` + `bert.lang_encoder.layer.*` from METER's `text_transformer` = a RoBERTa
    (pretrain_src/train_r2r_magic.py:189-198: 'text_transformer.embeddings' -> 'bert.embeddings', 'text_transformer.encoder' ->
    'bert.lang_encoder').  So the oracle's whole text branch -- RoBERTa position offset (open choice O1), token type 0, embedding
    LayerNorm, N post-LN layers, additive -10000 key mask (O3), last-layer attention maps (O2) -- must equal transformers'
    RobertaModel under the same key names.  Pads are id 0 as the reference's collate writes them (tasks.py: padding_value=0)."""
    pytest.importorskip("transformers")
    from transformers import RobertaConfig, RobertaModel
    cfg = make_config(128, vocab_size=300, num_l_layers=3, num_x_layers=1, num_pano_layers=1)
    torch.manual_seed(1)
    ours = R.RefMagicBert(cfg).eval()
    for n, p in ours.named_parameters():
        if p.dim() > 1:
            torch.nn.init.normal_(p, 0, 0.05)
        elif "LayerNorm" in n or "layer_norm" in n:
            torch.nn.init.normal_(p, 0.5 if n.endswith("weight") else 0.0, 0.2)
        else:
            torch.nn.init.normal_(p, 0, 0.1)
    hc = RobertaConfig(vocab_size=300, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=512,
                       max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
                       layer_norm_eps=cfg.layer_norm_eps, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, pad_token_id=1)
    hc._attn_implementation = "eager"
    hf = RobertaModel(hc, add_pooling_layer=False).eval()
    sd = {}
    for k, v in ours.state_dict().items():
        if k.startswith("embeddings."):
            sd[k] = v
        elif k.startswith("lang_encoder."):
            sd[k.replace("lang_encoder.", "encoder.", 1)] = v
    res = hf.load_state_dict(sd, strict=False)
    assert not [k for k in res.missing_keys if "position_ids" not in k and "token_type_ids" not in k], res.missing_keys
    assert not res.unexpected_keys, res.unexpected_keys
    g = torch.Generator().manual_seed(0)
    B, L = 4, 17
    lens = torch.tensor([17, 9, 12, 5])
    ids = torch.randint(3, 300, (B, L), generator=g)
    mask = torch.arange(L)[None] < lens[:, None]
    ids[~mask] = 0
    ids[:, 0] = 0                                                      # <s>
    with torch.no_grad():
        got, got_p = ours.text(ids, mask)
        want = hf(input_ids=ids, attention_mask=mask.long(), output_attentions=True)
    for b in range(B):
        n = int(lens[b])
        torch.testing.assert_close(got[b, :n], want.last_hidden_state[b, :n], rtol=1e-5, atol=2e-5)
        torch.testing.assert_close(got_p[b, :, :n, :n], want.attentions[-1][b, :, :n, :n], rtol=1e-5, atol=1e-6)


def test_cross_layer_matches_hf_bertlayer_with_cross_attention():
    """METER's BertCrossLayer (the `cross_modal_image_layers` the reference copies into BOTH `global_encoder.encoder.crossattention.*`
    and `local_encoder.encoder.crossattention.*`, train_r2r_magic.py:199-202) is HF's BertLayer with cross attention: self-attention ->
    cross-attention over the other modality -> FFN, same parameter names.  Open choice O6 pinned to transformers' implementation."""
    pytest.importorskip("transformers")
    from transformers.models.bert.modeling_bert import BertConfig, BertLayer
    cfg = make_config(128)
    torch.manual_seed(2)
    ours = R.RefCrossLayer(cfg).eval()
    for n, p in ours.named_parameters():
        torch.nn.init.normal_(p, 0.3 if ("LayerNorm.weight" in n) else 0.0, 0.08)
    hc = BertConfig(hidden_size=128, num_attention_heads=2, intermediate_size=512, layer_norm_eps=cfg.layer_norm_eps,
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, is_decoder=True, add_cross_attention=True)
    hc._attn_implementation = "eager"
    hf = BertLayer(hc).eval()
    hf.load_state_dict(ours.state_dict(), strict=True)
    g = torch.Generator().manual_seed(3)
    x, ctx = torch.randn(3, 11, 128, generator=g), torch.randn(3, 19, 128, generator=g)
    xm = torch.arange(11)[None] < torch.tensor([11, 7, 4])[:, None]
    cm = torch.arange(19)[None] < torch.tensor([19, 19, 6])[:, None]
    sb, cb = R.key_bias(xm), R.key_bias(cm)
    with torch.no_grad():
        got, got_p = ours(x, sb, ctx, cb)
        want = hf(x, attention_mask=sb, encoder_hidden_states=ctx, encoder_attention_mask=cb, output_attentions=True)
    out = want[0] if isinstance(want, tuple) else want
    for b, n in enumerate((11, 7, 4)):
        torch.testing.assert_close(got[b, :n], out[b, :n], rtol=1e-5, atol=2e-5)
