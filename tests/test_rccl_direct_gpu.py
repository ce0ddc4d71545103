"""RCCL through its C ABI (host/rccl.py) and the bucket collectives INSIDE the captured step graph (-m gpu; a child process with its own world-1 `nccl`
group, the way bench.py --dp-structure runs): the communicator comes up and passes its self-test; sum all-reduce / all-gather are the identity at world 1
in and out of a group launch; a step captured with its exchange inside ONE graph (trainer.capture_student, MAGIC_DP_STRUCTURE=1) replays to the SAME
weights as the plain single-graph step and as the cut-graph data-parallel form with torch.distributed's calls between the replays.  N > 1 over RCCL
has not run on hardware (no multi-GPU node): tests/test_distributed_cpu.py and test_ddp_overlap_gpu.py cover the exchange's arithmetic with gloo ranks."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, socket, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import magic_amd
from magic_amd.host import rccl, synth
from magic_amd.host.plan import build_plan
from magic_amd.host.trainer import PretrainStep
from tests.test_model_gpu import RW, build
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
c = rccl.make(dev)
assert c is not None and c.world == 1, "direct communicator"
assert c.graph_ok, "the captured self-test (an all-reduce inside a twice-replayed graph, forked to a side stream) must pass on this box"
x = torch.randn(1000, device=dev); ref = x.clone()
c.all_reduce_(x)
with c.group():
    c.all_reduce_(x[:500]); c.all_reduce_(x[500:])
ids = torch.arange(7, dtype=torch.int64, device=dev); out = torch.empty(7, dtype=torch.int64, device=dev)
rows = torch.randn(7, 16, device=dev); orow = torch.empty_like(rows)
with c.group():
    c.all_gather(out, ids); c.all_gather(orow.view(-1), rows.view(-1))
torch.cuda.synchronize()
assert torch.equal(x, ref) and torch.equal(out, ids) and torch.equal(orow, rows)
try:
    c.all_reduce_(x.double()); raise SystemExit("fp64 all-reduce must be refused")
except rccl.RcclError:
    pass

rw = torch.tensor(RW, device=dev)
SCHED = [("sap", 0), ("mlm", 1), ("cfp", 2), ("sap", 3)]
batches = [synth.make_batch(t, batch_size=4, seed=31, step=s, vocab=600, min_len=8, max_len=15, min_steps=2, max_steps=3) for t, s in SCHED]

def run(form):
    os.environ.pop("MAGIC_DP_STRUCTURE", None); os.environ.pop("MAGIC_DDP_GRAPH_RCCL", None); os.environ.pop("MAGIC_RCCL_DIRECT", None)
    if form != "plain":
        os.environ["MAGIC_DP_STRUCTURE"] = "1"
    if form == "cut":
        os.environ["MAGIC_DDP_GRAPH_RCCL"] = "0"; os.environ["MAGIC_RCCL_DIRECT"] = "0"
    _, _, g_t, g_s = build(torch.bfloat16, pretrain_tasks={"mlm", "sap", "cfp"})
    g_s.keep_mlm_logits = False
    tr = PretrainStep(g_s, g_t, lr=5e-5, warmup_steps=2, num_train_steps=40, sparse_embedding_rows=4 * 80)
    assert (tr.sync.rccl is not None) == (form == "in_graph"), form
    losses = []
    for (task, _), b in zip(SCHED, batches):
        bd, plan = synth.batch_to(b, dev), build_plan(b, task, dev)
        t_out = tr.teacher_forward(bd, task, plan)
        cs = tr.capture_student((bd, task, plan), t_out, rw=rw)
        assert bool(getattr(cs, "rccl_in_graph", False)) == (form == "in_graph") and (cs.graph2 is not None) == (form == "cut"), form
        out = tr.replay_student(cs)
        torch.cuda.synchronize()
        losses.append(float(out["loss"]))
    return losses, g_s.store.flat.clone()

want_l, want_w = run("plain")
for form in ("in_graph", "cut"):
    got_l, got_w = run(form)
    assert all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(got_l, want_l)), (form, got_l, want_l)
    d = (got_w - want_w).abs().max().item()
    assert d < 1e-6, (form, d)            # identical arithmetic; the embedding-stage atomics order is the only run-to-run freedom
c.destroy()
dist.destroy_process_group()
print("RCCL_DIRECT_OK")
""" % ROOT


def test_direct_rccl_communicator_and_the_collectives_inside_the_step_graph():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MAGIC_DP_STRUCTURE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_DIRECT_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
