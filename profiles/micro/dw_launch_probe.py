"""Round 6: the benchmarked step's weight-gradient launch ALONE -- the sap step's 84 problems (profiles/micro/r06_dw_problems.txt) with random bf16
operands in buffers of their own (290 MB), launched through magic_gemm_dw_grouped as the step does, timed with events; 320 MB written between
launches so that the operands do not sit in L2 (they may sit in the 256 MB Infinity Cache, as most of a step's saved activations do).
  python profiles/micro/dw_launch_probe.py [--rows-per-split R] [--no-bias] [--pow2 up|down|0] [--reps 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SAP = [(6, 3840, 128, 512), (6, 3840, 512, 128), (2, 10116, 128, 512), (2, 10116, 512, 128), (6, 3840, 384, 128), (2, 10116, 384, 128), (7, 3840, 256, 128),
       (1, 10116, 128, 768), (6, 3840, 128, 128), (2, 10116, 128, 128), (10, 1776, 128, 128), (1, 10116, 256, 128), (3, 1776, 128, 512), (3, 1776, 512, 128),
       (10, 1296, 128, 128), (3, 1776, 384, 128), (3, 1296, 128, 512), (3, 1296, 512, 128), (3, 1296, 384, 128), (1, 1776, 256, 128), (1, 1296, 256, 128),
       (1, 281, 256, 128), (2, 48, 128, 128)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-per-split", type=int, default=0, help="0: the product's rule (host/ops.py _splitk)")
    ap.add_argument("--max-split", type=int, default=64)
    ap.add_argument("--no-bias", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--label", default="")
    ap.add_argument("--shared-operands", action="store_true", help="every problem reads the SAME dY / X buffers (3 MB: L2-resident) -- what the launch costs without memory traffic")
    ap.add_argument("--no-flush", action="store_true")
    a = ap.parse_args()
    from magic_amd.host import lib as L, ops as O
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    probs = []
    pool = (torch.randn(10116 * 768, device=dev, generator=g) * 0.1).to(torch.bfloat16) if a.shared_operands else None
    for c, M, N, K in SAP:
        for _ in range(c):
            dy = (torch.randn(M, N, device=dev, generator=g) * 0.1).to(torch.bfloat16)
            x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
            if pool is not None:
                dy, x = pool[:M * N].view(M, N), pool[:M * K].view(M, K)
            dW = torch.zeros(N, K, device=dev)
            db = None if a.no_bias else torch.zeros(N, device=dev)
            tiles = ((N + 63) // 64) * ((K + 63) // 64)
            if a.rows_per_split:
                sk = max(1, min(a.max_split, round(M / a.rows_per_split)))
                if O.SPLITK["pow2"] and sk < 8:
                    lo = 1 << (sk.bit_length() - 1)
                    sk = lo if (sk == lo or O.SPLITK["pow2"] == "down") else 2 * lo
            else:
                sk = O._splitk(tiles, M)
            probs.append((dy, x, dW, db, M, N, K, N, K, K, sk))
    arr = (L.DwDesc * len(probs))()
    wgs = 0
    for j, (dy, x, dW, db, M, N, K, lda, ldb, ldc, sk) in enumerate(probs):
        arr[j] = L.DwDesc(L.P(dy), L.P(x), L.P(dW), L.P(db), M, N, K, lda, ldb, ldc, sk)
        wgs += ((N + 63) // 64) * ((K + 63) // 64) * sk
    flush = torch.empty(80 << 20, dtype=torch.float32, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
    for r in range(a.reps + 3):
        if not a.no_flush:
            flush.add_(1.0)
        if r >= 3:
            ev[r - 3][0].record()
        O.dw_grouped(torch.bfloat16, arr, len(probs), dev)
        if r >= 3:
            ev[r - 3][1].record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
    print("   ", "shared operands" if a.shared_operands else "", "no flush" if a.no_flush else "")
    ref = probs[0][0].float().t() @ probs[0][1].float() * (a.reps + 3)
    err = float((probs[0][2] - ref).abs().max() / ref.abs().max())
    print(f"{a.label or 'dw launch'}: rows/split {a.rows_per_split or 'product rule'} bias {'no' if a.no_bias else 'yes'} pow2 {O.SPLITK['pow2'] or 'off'} "
          f"groups {os.environ.get('MAGIC_DW_XCD_GROUPS', '1')}: {wgs} workgroups, median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f}, max {ts[-1]:.1f}; rel err of dW[0] {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
