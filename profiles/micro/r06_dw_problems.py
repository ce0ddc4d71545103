"""Round 6: the problems of the benchmarked step's ONE weight-gradient launch (csrc/gemm.hip gemm_dw_batch_kernel), per proxy task: rows M, dW shape
N x K, K-splits, workgroups, algorithmic bytes (every dY / X element once + the fp32 dW once), and the bytes the launch's 64 x 64 tiles request
(every tile reads its dY column panel and its X column panel over its split's rows; the workspace round trip of the deterministic seam).
Run on the GPU box:  python profiles/micro/r06_dw_problems.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from magic_amd.host import ops as O, synth  # noqa: E402
from magic_amd.host.plan import build_plan  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    tcfg, scfg, teacher, student, trainer = B.build_models(torch.bfloat16, dev, 0.1, 1)
    keep = O.dw_grouped
    seen = []

    def spy(dt, arr, n, device, deterministic=None):
        seen.append([(arr[j].M, arr[j].N, arr[j].K, arr[j].splitk, arr[j].lda, arr[j].ldb) for j in range(n)])
        return keep(dt, arr, n, device, deterministic)

    O.dw_grouped = spy
    for i, task in enumerate(B.TASKS):
        b = synth.make_batch(task, batch_size=48, seed=1234, step=i)
        plan = build_plan(b, task, dev)
        bd = synth.batch_to(b, dev)
        trainer._zero_grad()
        del seen[:]
        trainer._fwd_bwd(bd, task, None, plan)
        torch.cuda.synchronize()
        for li, probs in enumerate(seen):
            alg = req = wgs = wsb = 0
            rows = {}
            for (M, N, K, sk, lda, ldb) in probs:
                nx, ny = (K + 63) // 64, (N + 63) // 64
                a = M * (N + K) * 2 + 4 * N * K
                r = M * 64 * 2 * 2 * nx * ny                     # each tile: 64 dY columns + 64 X columns over all rows (summed over its splits)
                w = (2 * sk * nx * ny * 64 * 64 * 4) if sk > 1 else 0
                alg += a; req += r; wgs += nx * ny * sk; wsb += w
                key = (M, N, K, sk)
                rows[key] = rows.get(key, 0) + 1
            print(f"{task} launch {li}: {len(probs)} problems, {wgs} workgroups, algorithmic {alg / 1e6:.1f} MB, tile requests {req / 1e6:.1f} MB, "
                  f"workspace round trip {wsb / 1e6:.1f} MB", flush=True)
            for (M, N, K, sk), c in sorted(rows.items(), key=lambda kv: -kv[0][0] * (kv[0][1] + kv[0][2]) * kv[1]):
                nx, ny = (K + 63) // 64, (N + 63) // 64
                print(f"    {c:3d} x  M {M:6d}  dW {N:6d} x {K:5d}  splitk {sk:3d}  tiles {nx * ny:5d}  rows/split {M // sk:5d}  alg {c * (M * (N + K) * 2 + 4 * N * K) / 1e6:7.2f} MB  "
                      f"requested {c * M * 256 * nx * ny / 1e6:7.2f} MB")


if __name__ == "__main__":
    main()
