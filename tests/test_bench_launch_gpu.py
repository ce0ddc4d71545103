"""The data-parallel bench path through bench.py's own launcher, 2 ranks on the ONE card of the test box (gloo carries the exchange: RCCL needs
one GPU per rank): split student graph + bucketed exchange between the halves + eager optimizer, rank roll-call in the line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_two_ranks_one_card_through_self_launcher():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "2", "--pool", "3",
                        "--no-cpu-baseline", "--no-parity", "--no-secondary", "--no-profile"],
                       capture_output=True, text=True, timeout=850, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["config"]["parallelism"] == "dp2" and j["config"]["global_batch"] == 96
    assert sorted(j["rccl"]["ranks_seen"]) == [0, 1] and j["value"] > 0
    # two ranks on one card: the trainer must have noticed and kept the launches that need a whole grid resident at once off -- a neighbour's
    # kernels can take the slots their late workgroups need (bench.py ends with trainer.check_health(), which raises if a hand-off gave up)
    assert j["health"]["card_shared_with_other_ranks"] is True and j["health"]["row_split_encoder_launches"] is False
