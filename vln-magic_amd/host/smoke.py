"""__graft_entry__.smoke(): one small MAKD training step (teacher fwd, student fwd+loss, explicit backward,
fused AdamW) of the hot path on cuda:0 in bf16, checked against the CPU oracle (oracle/ is the checker only)."""
import torch

from . import lib as L
from . import synth
from .config import make_config
from .model_pretrain import GlocalTextPathCMTPreTraining
from .trainer import PretrainStep

KDL = dict(knowledge_distillation=True, kd_alpha=0.5, kd_temperature=2, teacher_sample_hard_mining=True,
           t_sample_preprocess_exp_decay=0.7, rw_temp=4,
           kdl_tasks=["txt", "img", "local", "global", "predict"], kdl_task_types=["emb", "attn"])


def run_smoke():
    from oracle import model_ref as R          # checker only
    L.load()
    kw = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=500, num_l_layers=2, num_x_layers=1, num_pano_layers=1)
    tcfg = make_config(256, role="teacher", **kw)
    scfg = make_config(128, role="student", teacher_hidden_size=256, kdl=KDL, **kw)
    torch.manual_seed(0)
    o_t, o_s = R.RefPretrainModel(tcfg).eval(), R.RefPretrainModel(scfg).eval()
    rw = [1.0, 1.2, 0.8, 1.1, 0.9]
    for dtype, tol in ((torch.float32, 2e-4), (torch.bfloat16, 5e-2), (torch.float16, 5e-3)):
        g_t = GlocalTextPathCMTPreTraining.from_pretrained(None, config=tcfg, state_dict=o_t.state_dict(), device="cuda:0", compute_dtype=dtype)
        g_s = GlocalTextPathCMTPreTraining.from_pretrained(None, config=scfg, state_dict=o_s.state_dict(), device="cuda:0", compute_dtype=dtype)
        trainer = PretrainStep(g_s, g_t, lr=1e-4, warmup_steps=1, num_train_steps=10)
        batch = synth.make_batch("sap", batch_size=4, seed=1, vocab=500, min_len=6, max_len=12, min_steps=2, max_steps=3)
        with torch.no_grad():
            ot = o_t(batch, "sap")["outputs"]
            want = o_s(batch, "sap", teacher_outputs=ot, rw=torch.tensor(rw))
        out = trainer.step(batch, "sap", rw=rw)
        torch.cuda.synchronize()
        assert trainer.check_health()["skipped_optimizer_steps"] == 0
        got, ref = float(out["loss"]), float(want["loss"])
        assert abs(got - ref) <= tol * max(1.0, abs(ref)), f"smoke loss mismatch ({dtype}): {got} vs oracle {ref}"
        a = out["outputs"]["fused_logits"].float().cpu()
        b = want["outputs"]["fused_logits"]
        assert torch.equal(torch.isinf(a), torch.isinf(b))
        if dtype == torch.float32:
            assert torch.equal(a.argmax(1), b.argmax(1)), "action argmax differs from oracle"
        print(f"smoke[{str(dtype).split('.')[-1]}]: loss {got:.6f} (oracle {ref:.6f}) ok")
