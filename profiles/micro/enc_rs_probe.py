"""whole-encoder forward: per-sample form vs row-split form, text segment alone / panorama segment alone / both in one launch (B=48 bench batch)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import magic_amd  # noqa
from magic_amd.host import ops as O, synth
from magic_amd.host.config import make_config
from magic_amd.host.model_pretrain import GlocalTextPathCMTPreTraining
from magic_amd.host.plan import build_plan

dev = torch.device("cuda")
pd = float(os.environ.get("PDROP", "0.1"))
cfg = make_config(128, role="student", teacher_hidden_size=256, hidden_dropout_prob=pd, attention_probs_dropout_prob=pd)
m = GlocalTextPathCMTPreTraining(cfg, device=dev, compute_dtype=torch.bfloat16, seed=1)
m.train()
batch = synth.make_batch("sap", batch_size=48, seed=1234, step=1)
plan = build_plan(batch, "sap", dev)
inp = m._inputs(synth.batch_to(batch, dev), plan)
m.store.sync_shadow()
seed = torch.tensor([1, 2], dtype=torch.int32, device=dev)
n = m.net
print("B", plan["B"], "L", plan["L"], "Np", plan["Np"], "V", plan["V"])


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rs in (False, True):
    O.ENC_ROW_SPLIT = rs
    n.set_dropout(seed if pd > 0 else None, pd, pd)
    ct = n.text_fwd(plan, defer=True)
    cp = n.pano_fwd(plan, inp.feats, inp.loc, defer=True)
    st, _ = n._enc_segment(*ct.pending)
    sp, _ = n._enc_segment(*cp.pending)
    t_txt = timeit(lambda: n._enc_launch([st]))
    t_pano = timeit(lambda: n._enc_launch([sp]))
    t_both = timeit(lambda: n._enc_launch([st, sp]))
    print(f"row_split={rs}: text alone {t_txt:.1f} us, pano alone {t_pano:.1f} us, both {t_both:.1f} us", flush=True)
    if rs:      # the mixed launch with one of its halves shrunk to a single sample: what each half costs INSIDE that kernel
        sp1 = dict(sp, nsamp=1, x=sp["x"][:sp["N"]])
        st1 = dict(st, nsamp=1, x=st["x"][:st["N"]])
        for k in (8, 16, 32, 64, 128, 200, 256):
            spk = dict(sp, nsamp=k, x=sp["x"][:k * sp["N"]])
            print(f"   mixed kernel: all text tiles + {k} panoramas {timeit(lambda: n._enc_launch([st, spk])):.1f} us", flush=True)
        print(f"   mixed kernel: text tiles + 1 panorama {timeit(lambda: n._enc_launch([st, sp1])):.1f} us, 1 instruction + all panoramas {timeit(lambda: n._enc_launch([st1, sp])):.1f} us", flush=True)
