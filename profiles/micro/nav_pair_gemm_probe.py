"""the navigator step's PAIRED GEMMs (the map branch's and the viewpoint branch's twin Linears in one grouped launch, lib.group): the plain
64 x 64 grouped kernel against its K-group form (csrc/gemm.hip gemm_grouped_kg_kernel; MAGIC_GEMM_KG_GROUP=0/1 per process), and against the two
launches one after the other.  Graph-replayed back to back, weights rotating through a set larger than the L2s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import magic_amd  # noqa: E402,F401
from magic_amd.host import lib as L  # noqa: E402
from magic_amd.host import ops as O  # noqa: E402
from gemm_tile_sweep import timed  # noqa: E402

NW = 24                                                   # distinct weights per shape: 24 x 1.2-4.7 MB, well past 8 x 4 MB of L2
for (Ma, Mb), N, K in [((624, 256), 768, 768), ((624, 512), 768, 768), ((624, 1024), 768, 768), ((624, 512), 3072, 768), ((624, 512), 768, 3072),
                       ((624, 1024), 3072, 768), ((624, 1024), 768, 3072), ((624, 512), 2304, 768), ((592, 592), 768, 768)]:
    xa, xb = (torch.randn(M, K, device="cuda", dtype=torch.bfloat16) for M in (Ma, Mb))
    Wa = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05 for _ in range(NW)]
    Wb = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05 for _ in range(NW)]
    b = torch.zeros(N, device="cuda")
    oa, ob = (torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for M in (Ma, Mb))
    dya, dyb = (torch.randn(M, N, device="cuda", dtype=torch.bfloat16) for M in (Ma, Mb))
    dxa, dxb = (torch.empty(M, K, device="cuda", dtype=torch.bfloat16) for M in (Ma, Mb))
    it = {"i": 0}

    def nxt():
        it["i"] = (it["i"] + 1) % NW
        return it["i"]

    def pair_nt():
        i = nxt()
        with L.group():
            O.linear_fwd(xa, Wa[i], b, Ma, out=oa)
            O.linear_fwd(xb, Wb[i], b, Mb, out=ob)

    def solo_nt():
        i = nxt()
        O.linear_fwd(xa, Wa[i], b, Ma, out=oa)
        O.linear_fwd(xb, Wb[i], b, Mb, out=ob)

    def pair_nn():
        i = nxt()
        with L.group():
            O.linear_dx(dya, Wa[i], Ma, out=dxa)
            O.linear_dx(dyb, Wb[i], Mb, out=dxb)

    def solo_nn():
        i = nxt()
        O.linear_dx(dya, Wa[i], Ma, out=dxa)
        O.linear_dx(dyb, Wb[i], Mb, out=dxb)
    r = dict(Ma=Ma, Mb=Mb, N=N, K=K, kg_group=os.environ.get("MAGIC_GEMM_KG_GROUP", "1"))
    r["pair_nt_us"], r["two_nt_us"] = round(timed(pair_nt, NW), 2), round(timed(solo_nt, NW), 2)
    r["pair_nn_us"], r["two_nn_us"] = round(timed(pair_nn, NW), 2), round(timed(solo_nn, NW), 2)
    print(json.dumps(r), flush=True)
