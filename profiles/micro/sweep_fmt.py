"""merge two gemm_tile_sweep.py logs (MAGIC_GEMM_BIG=0 and =2) into the table kept as profiles/micro/r01_gemm_tile_sweep.txt"""
import json
import sys


def rows(path):
    return [json.loads(l) for l in open(path) if l.startswith("{")]


a, b = rows(sys.argv[1]), rows(sys.argv[2])
print("profiles/micro/gemm_tile_sweep.py on MI355X, bf16, us per launch, 20 launches graph-replayed back to back (warm operands)")
print("64 = 64x64 block tile (MAGIC_GEMM_BIG=0), wide = 128x128 LDS-DMA tile forced on (MAGIC_GEMM_BIG=2), torch = torch.matmul (hipBLASLt)")
print("NT = linear forward y[M,N] = x[M,K] W[N,K]^T; NN = dx[M,K] = dy[M,N] W[N,K]; TN = dW[N,K] += dy^T x (fp32 atomics, host-chosen split-K)")
print("* = what the default rule (mode 1: >= 192 wide tiles and K >= 512, forward / input gradient only) picks\n")
print("     M     N     K |    NT64  NTwide torchNT |    NN64  NNwide |    TN64  TNwide torchTN")
for x, y in zip(a, b):
    M, N, K = x["M"], x["N"], x["K"]
    t_nt = ((M + 127) // 128) * ((N + 127) // 128)
    t_nn = ((M + 127) // 128) * ((K + 127) // 128)
    s_nt = "*" if (t_nt >= 192 and K >= 512 and M >= 128 and N >= 128) else " "
    s_nn = "*" if (t_nn >= 192 and N >= 512 and M >= 128 and K >= 128) else " "
    print(f"{M:6d}{N:6d}{K:6d} | {x['nt_us']:7.1f} {y['nt_us']:6.1f}{s_nt} {x['torch_nt_us']:7.1f} | {x['nn_us']:7.1f} {y['nn_us']:6.1f}{s_nn} |"
          f" {x['tn_us']:7.1f} {y['tn_us']:7.1f} {x['torch_tn_us']:7.1f}")
