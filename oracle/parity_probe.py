"""TEST INFRASTRUCTURE (checker only; never on the product path): runs the HIP engine and the fp64 CPU oracle on the SAME
weights and the SAME synthetic batch at the true MAGIC-S pretraining configuration -- 6 text / 3 cross-modal / 2 panorama
layers, vocabulary 50265, student H=128 (2 heads) against the frozen H=256 (4 heads) teacher, <= 80 instruction tokens,
36 views x 768 (pretrain_src/config/r2r_magic_model_config.json:10-13,26,33-43) -- and reports what the north star
(BASELINE.json) states its tolerance on: action-logit |delta| and action argmax agreement, plus the loss terms.

Used by tests/test_fullsize_oracle_gpu.py (assertions) and by bench.py's parity leg (the numbers printed next to the
throughput of the same arithmetic mode).  The oracle (oracle/model_ref.py) is a restatement: the reference withholds its
model source (readme.md:75), so "the reference" here is the committed oracle -- DESIGN.md section 0.
"""
import torch

KDL = dict(knowledge_distillation=True, kd_alpha=0.5, kd_temperature=2, teacher_sample_hard_mining=True,
           t_sample_preprocess_exp_decay=0.7, rw_temp=4,
           kdl_tasks=["txt", "img", "local", "global", "predict"], kdl_task_types=["emb", "attn"])   # r2r_magic_pretrain.json:62-87
RW = [1.3, 0.7, 1.1, 0.9, 1.0]


def full_configs(**over):
    from magic_amd.host.config import make_config
    kw = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    kw.update(over)
    return make_config(256, role="teacher", **kw), make_config(128, role="student", teacher_hidden_size=256, kdl=KDL, **kw)


def to64(batch):
    return {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}


def oracle_models(seed=0, perturb=True, **over):
    """fp64 oracle teacher + student with non-trivial biases / LayerNorm parameters"""
    from oracle import model_ref as R
    tcfg, scfg = full_configs(**over)
    torch.manual_seed(seed)
    o_t, o_s = R.RefPretrainModel(tcfg).eval(), R.RefPretrainModel(scfg).eval()
    if perturb:
        with torch.no_grad():
            for m in (o_t, o_s):
                for n, p in m.named_parameters():
                    if n.endswith("bias"):
                        p.normal_(0, 0.02)
                    if "LayerNorm.weight" in n or "layer_norm.weight" in n or n.endswith("net.2.weight") or n.endswith("embeddings.1.weight"):
                        p.add_(torch.randn_like(p) * 0.05)
    return tcfg, scfg, o_t.double(), o_s.double()


def engine_models(tcfg, scfg, o_t, o_s, dtype, device="cuda"):
    from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining as M
    sd = lambda m: {k: v.float() for k, v in m.state_dict().items()}
    g_t = M.from_pretrained(None, config=tcfg, state_dict=sd(o_t), device=device, compute_dtype=dtype)
    g_s = M.from_pretrained(None, config=scfg, state_dict=sd(o_s), device=device, compute_dtype=dtype)
    g_s.keep_mlm_logits = True
    return g_t, g_s


def oracle_step(o_t, o_s, batch, task, backward=False):
    b64 = to64(batch)
    with torch.no_grad():
        ot = o_t(b64, task, compute_loss=True)["outputs"]
    for p in o_s.parameters():
        p.grad = None
    want = o_s(b64, task, compute_loss=True, teacher_outputs=ot, rw=torch.tensor(RW, dtype=torch.float64))
    if backward:
        want["loss"].backward()
    return ot, want


def engine_step(g_t, g_s, batch, task, backward=False):
    with torch.no_grad():
        gt = g_t(batch, task, compute_loss=False, return_outputs=True)
    g_s.store.zero_grad()
    got = g_s(batch, task, compute_loss=True, teacher_outputs=gt, rw=RW, plan=gt["plan"])
    if backward:
        g_s.backward()
        gs = float(getattr(g_s, "grad_scale", 1.0))
        if gs != 1.0:                     # fp16 engine: the flat buffer holds grad_scale x the gradient (the optimizer divides)
            g_s.store.grad.mul_(1.0 / gs)
    torch.cuda.synchronize()
    return gt, got


def logit_stats(got_outputs, want_outputs, keys=("global_logits", "local_logits", "fused_logits")):
    """per logit tensor: max |delta| over the valid (non -inf) entries, argmax agreement, identical -inf pattern; plus the
    smallest top-1 / top-2 gap of the oracle's fused logits (how close the nearest tie is)"""
    r = {}
    for k in keys:
        a, b = got_outputs[k].detach().double().cpu(), want_outputs[k].detach().double()
        same_inf = bool(torch.equal(torch.isinf(a), torch.isinf(b)))
        d = (torch.nan_to_num(a, neginf=0.0) - torch.nan_to_num(b, neginf=0.0)).abs().max().item()
        ia, ib = a.argmax(1), b.argmax(1)
        agree = (ia == ib).double().mean().item()
        # where the picks differ: how far apart the ORACLE itself rates the two candidates (a pick can only flip inside a band of
        # twice the logit error: got[j] >= got[i] and |got - want| <= delta imply want[i] - want[j] <= 2 delta)
        rows = torch.arange(a.shape[0])
        flip_gap = (b[rows, ib] - b[rows, ia])[ia != ib]
        r[k] = {"max_abs_delta": d, "argmax_agreement": agree, "same_inf_mask": same_inf, "rows": int(a.shape[0]),
                "worst_flip_gap": flip_gap.max().item() if flip_gap.numel() else 0.0}
    top2 = want_outputs["fused_logits"].detach().double().topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1])
    r["oracle_min_top2_gap"] = gap[torch.isfinite(gap)].min().item()
    return r


def sap_parity(dtype, batch_size=8, seeds=(1234,), device="cuda", models=None):
    """Full-size SAP step of the engine in `dtype` against the fp64 oracle; returns the aggregated statistics over `seeds`
    (worst delta, mean agreement) and the loss deltas of the last seed."""
    from magic_amd.host import synth
    if models is None:
        tcfg, scfg, o_t, o_s = oracle_models()
    else:
        tcfg, scfg, o_t, o_s = models
    g_t, g_s = engine_models(tcfg, scfg, o_t, o_s, dtype, device)
    out = {"dtype": str(dtype).replace("torch.", ""), "batch_size": batch_size, "seeds": list(seeds), "layers": "6/3/2", "vocab": scfg.vocab_size,
           "max_abs_logit_delta": 0.0, "argmax_agreement": 1.0, "same_inf_mask": True, "rows": 0}
    agree_n = 0.0
    for sd in seeds:
        batch = synth.make_batch("sap", batch_size=batch_size, seed=sd, step=0)
        _, want = oracle_step(o_t, o_s, batch, "sap")
        _, got = engine_step(g_t, g_s, batch, "sap")
        st = logit_stats(got["outputs"], want["outputs"])
        for k in ("global_logits", "local_logits", "fused_logits"):
            out["max_abs_logit_delta"] = max(out["max_abs_logit_delta"], st[k]["max_abs_delta"])
            out["same_inf_mask"] = out["same_inf_mask"] and st[k]["same_inf_mask"]
        agree_n += st["fused_logits"]["argmax_agreement"] * st["fused_logits"]["rows"]
        out["rows"] += st["fused_logits"]["rows"]
        out["fused_max_abs_delta"] = max(out.get("fused_max_abs_delta", 0.0), st["fused_logits"]["max_abs_delta"])
        out["worst_flip_gap"] = max(out.get("worst_flip_gap", 0.0), st["fused_logits"]["worst_flip_gap"])
        out["oracle_min_top2_gap"] = min(out.get("oracle_min_top2_gap", 1e9), st["oracle_min_top2_gap"])
        out["loss_rel_delta"] = abs(float(got["loss"].detach()) - float(want["loss"])) / max(abs(float(want["loss"])), 1e-12)
        out["kdl_rel_delta"] = abs(float(got["kdl_loss"].detach()) - float(want["kdl_loss"])) / max(abs(float(want["kdl_loss"])), 1e-12)
    out["argmax_agreement"] = agree_n / max(out["rows"], 1)
    del g_t, g_s
    return out
