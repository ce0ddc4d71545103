"""Where the fused attention backward (csrc/attention.hip attn_bwd_body) spends its time at the text shape of the headline step (B = 48, 2 heads,
80 x 80, dropout on): wall_clock64 marks of workgroup (0, 0) + the average launch duration back to back.  Builds its own copy of the library with
-DMAGIC_ATTN_TIMING (profiles/micro/_bin/libmagic_attn_timing.so) and drives it through ctypes directly.
    python profiles/micro/attn_bwd_probe.py"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "vln-magic_amd", "csrc")
BIN = os.path.join(ROOT, "profiles", "micro", "_bin")
os.makedirs(BIN, exist_ok=True)
so = os.path.join(BIN, "libmagic_attn_timing.so")
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value"]
subprocess.run(["hipcc"] + flags + ["-DMAGIC_ATTN_TIMING", "-c", os.path.join(CS, "attention.hip"), "-o", os.path.join(BIN, "attention_t.o")], check=True)
objs = [os.path.join(CS, f) for f in os.listdir(CS) if f.endswith(".o") and f != "attention.o"]
subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(BIN, "attention_t.o")] + objs + ["-o", so], check=True)
lib = C.CDLL(so)
dev = "cuda"
B, nh, N, H = 48, 2, 80, 128
ldp = 80
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev).bfloat16().contiguous()
qkv, dctx = rnd(B * N, 3 * H), rnd(B * N, H)
P = torch.softmax(torch.randn(B, nh, N, ldp, generator=g), -1).to(dev).bfloat16().contiguous()
dqkv = torch.empty(B * N, 3 * H, dtype=torch.bfloat16, device=dev)
seed = torch.tensor([123, 456], dtype=torch.int32, device=dev)
vp, i32, f32, u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint
lib.magic_attn_bwd.argtypes = [i32, i32, i32, i32, i32, vp, i32, vp, vp, i32, vp, i32, vp, i32, f32, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, f32, u32, vp]
lib.magic_attn_bwd.restype = i32
st = torch.cuda.current_stream().cuda_stream


def launch(drop=0.1):
    e = 2      # bytes
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + H * e, qkv.data_ptr() + 2 * H * e
    dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + H * e, dqkv.data_ptr() + 2 * H * e
    rc = lib.magic_attn_bwd(1, B, nh, N, N, q, 3 * H, k, v, 3 * H, P.data_ptr(), ldp, dctx.data_ptr(), H, 0.125, None, dq, 3 * H, dk, dv, 3 * H,
                            None, None, None, seed.data_ptr() if drop > 0 else None, drop, 77, st)
    assert rc == 0, rc


for drop in (0.1, 0.0):
    for _ in range(5):
        launch(drop)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        launch(drop)
    e1.record()
    torch.cuda.synchronize()
    t = (C.c_longlong * 8)()
    assert lib.magic_debug_attn_ticks(t) == 0
    tk = [t[i] for i in range(6)]
    names = ["loads -> LDS + barrier", "phase 1 (dP, dS) + barrier", "dQ + stores", "dK, dV + stores issued", "stores drained"]
    print(f"dropout {drop}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch back to back; workgroup (0,0): " +
          ", ".join(f"{nm} {(tk[i + 1] - tk[i]) / 100:.2f} us" for i, nm in enumerate(names)) + f"; total {(tk[5] - tk[0]) / 100:.2f} us")
