"""The fused input-embedding backward (csrc/rowops.hip embed_in_bwd_kernel) against its parts at the headline shapes: panorama stage alone, text
embedding alone (magic_ln_bwd), both in one launch, and the per-op sequence (ln_bwd pair + ln_bwd + smallk_ln_bwd).  Random operands, HIP events,
back to back (so operands are L2-warm: the in-step numbers are higher).   python profiles/micro/embed_bwd_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import magic_amd  # noqa: F401
from magic_amd.host import lib as L
from magic_amd.host import ops as O

dev, H, Kin = "cuda", 128, 7
M, Mt, Lt = 290 * 36, 48 * 80, 80
g = torch.Generator().manual_seed(0)
bf = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev).bfloat16().contiguous()
f32 = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev).contiguous()
pos = lambda *s: (torch.rand(*s, generator=g) + 0.5).to(dev).contiguous()
z = lambda *s: torch.zeros(*s, device=dev)
pano = dict(M=M, Kin=Kin, dy=bf(M, H), X0=bf(M, H), rstd3=pos(M), g3=pos(H), b3=f32(H), dg3=z(H), db3=z(H),
            nav_idx=torch.randint(0, 3, (M,), generator=g).to(dev).int(), d_nav=z(3, H), d_tok=z(1, H),
            A1=bf(M, H), rstd1=pos(M), g1=pos(H), b1=f32(H), dg1=z(H), db1=z(H), dP0=torch.empty(M, H, dtype=torch.bfloat16, device=dev),
            A2=bf(M, H), rstd2=pos(M), g2=pos(H), b2=f32(H), dg2=z(H), db2=z(H), loc=f32(M, Kin), dW=z(H, Kin), dbl=z(H))
ids = torch.randint(3, 50265, (Mt,), generator=g)
ids[torch.rand(Mt, generator=g) < 0.35] = 0                  # ~35 % padding tokens
text = dict(M=Mt, dy=bf(Mt, H), y=bf(Mt, H), gamma=pos(H), beta=f32(H), rstd=pos(Mt), dx=None, dgamma=z(H), dbeta=z(H),
            dtabs=((ids.to(dev).int(), 0, 0, z(50265, H), 0), (None, Lt, 2, z(514, H), 0), (None, 0, 0, pano["d_tok"], 0)), hot0=0)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def text_alone():
    t = dict(text)
    M_ = t.pop("M"); dy = t.pop("dy")
    O.ln_bwd(M_, H, dy, **t)


def per_op():
    with L.group():
        text_alone()
        dsum = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
        O.ln_bwd(M, H, pano["dy"], y=pano["X0"], gamma=pano["g3"], beta=pano["b3"], rstd=pano["rstd3"], dx=dsum, dgamma=pano["dg3"], dbeta=pano["db3"],
                 dtabs=((pano["nav_idx"], 0, 0, pano["d_nav"], 1), (None, 0, 0, pano["d_tok"], 0), None))
    O.ln_bwd(M, H, dsum, y=pano["A1"], gamma=pano["g1"], beta=pano["b1"], rstd=pano["rstd1"], dx=pano["dP0"], dgamma=pano["dg1"], dbeta=pano["db1"])
    O.smallk_ln_bwd(M, H, Kin, pano["loc"], dsum, pano["A2"], pano["g2"], pano["b2"], pano["rstd2"], pano["dW"], pano["dbl"], pano["dg2"], pano["db2"])


print(f"panorama stage alone (fused)      {timeit(lambda: O.embed_in_bwd(H, pano)):.1f} us")
if os.environ.get("PIB_VARIANTS"):       # timing variants of the fused kernel (built with -DPIB_VARIANT=1: no tail, 2: no atomics): where its time goes
    import ctypes as C
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    CS, BIN = os.path.join(ROOT, "vln-magic_amd", "csrc"), os.path.join(ROOT, "profiles", "micro", "_bin")
    os.makedirs(BIN, exist_ok=True)
    for v in (1, 2):
        so = os.path.join(BIN, f"libmagic_pib{v}.so")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value", f"-DPIB_VARIANT={v}", "-c",
                        os.path.join(CS, "rowops.hip"), "-o", os.path.join(BIN, f"rowops_v{v}.o")], check=True)
        objs = [os.path.join(CS, f) for f in os.listdir(CS) if f.endswith(".o") and f != "rowops.o"]
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(BIN, f"rowops_v{v}.o")] + objs + ["-o", so], check=True)
        lib = C.CDLL(so, mode=os.RTLD_NOW | os.RTLD_DEEPBIND)       # its OWN kernel stubs, not those of the libmagic_hip.so loaded above
        lib.magic_embed_in_bwd.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        a = L.PanoInBwd()
        a.M, a.Kin = M, Kin
        for k in ("dy", "X0", "rstd3", "g3", "b3", "dg3", "db3", "nav_idx", "d_nav", "d_tok", "A1", "rstd1", "g1", "b1", "dg1", "db1", "dP0",
                  "A2", "rstd2", "g2", "b2", "dg2", "db2", "loc", "dW", "dbl"):
            setattr(a, k, pano[k].data_ptr())
        st = torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize()
        t = (C.c_longlong * 8)()
        lib.magic_debug_pib_ticks(t)
        tk = [t[i] for i in range(7)]
        print("   block 0 ticks (us): start->prologue %.2f, it0 loads %.2f, it0 compute %.2f, it1 loads %.2f, it1 compute->end %.2f" %
              ((tk[1] - tk[0]) / 100, (tk[3] - tk[2]) / 100, (tk[4] - tk[3]) / 100, (tk[5] - tk[4]) / 100, (tk[6] - tk[5]) / 100))
        print(f"  variant {v} ({'no tail' if v == 1 else 'no atomics'})  {timeit(lambda: lib.magic_embed_in_bwd(1, H, C.addressof(a), None, 0, None, None, None, st)):.1f} us")
print(f"text embedding alone (ln_bwd)     {timeit(text_alone):.1f} us")
print(f"both in one launch                {timeit(lambda: O.embed_in_bwd(H, pano, text)):.1f} us")
print(f"per-op sequence (3 launches)      {timeit(per_op):.1f} us")
