"""ctypes binding of libmagic_hip.so (the C ABI of include/magic_hip.h).

The product path has NO CPU fallback: if the library is missing or a kernel returns an error code the
call raises.  Tensors are passed as raw device pointers; the stream is torch's current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libmagic_hip.so")

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_longlong, C.c_float

# name -> argument ctypes (mirrors include/magic_hip.h exactly; tests/test_abi.py checks every symbol)
SIGNATURES = {
    "magic_abi_version": [],
    "magic_device_info": [vp, vp, vp, i32],
    "magic_gemm": [i32, i32, i32, i32, i32, i32, i32, vp, i32, i64, i64, vp, i32, i64, i64, vp, i32, i64, i64, i32, i32,
                   vp, i32, vp, i32, vp, i32, vp, i32, f32, i32, vp, vp],
    "magic_gemm_dw_grouped": [i32, i32, vp, vp],
    "magic_linear_ln": [i32, i32, i32, i32, vp, i32, vp, i32, vp, vp, i32, vp, vp, f32, vp, vp, vp],
    "magic_ln_fwd": [i32, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, f32, vp, vp, i32, vp],
    "magic_ln_bwd": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, i32, vp, i32, i32, vp, i32,
                     vp, i32, i32, vp, i32, i32, vp],
    "magic_ln_pgrad": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp],
    "magic_smallk_ln_fwd": [i32, i32, i32, i32, vp, vp, vp, vp, vp, f32, vp, vp, vp],
    "magic_smallk_ln_bwd": [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "magic_softmax_fwd": [i32, i32, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp],
    "magic_softmax_bwd": [i32, i32, i32, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp],
    "magic_attn_supported": [i32, i32, i32, i32],
    "magic_attn_fwd": [i32, i32, i32, i32, i32, vp, i32, vp, vp, i32, vp, i32, vp, i32, f32, vp, vp, vp, vp, vp],
    "magic_attn_bwd": [i32, i32, i32, i32, i32, vp, i32, vp, vp, i32, vp, i32, vp, i32, f32, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp],
    "magic_head_mean_fwd": [i32, i32, i32, i64, vp, vp, vp],
    "magic_head_mean_bwd": [i32, i32, i64, vp, vp, i32, vp],
    "magic_lndot_fwd": [i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp],
    "magic_lndot_bwd": [i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp],
    "magic_ce_rows": [i32, i32, i32, vp, i32, vp, i32, f32, vp, vp, vp, i32, i32, vp, f32, vp],
    "magic_kd_rows": [i32, i32, vp, vp, i32, f32, vp, f32, f32, vp, vp, vp, i32, vp],
    "magic_mse": [i32, i32, i64, i64, vp, i64, vp, i64, vp, i64, f32, f32, vp, vp, vp, i64, i32, vp],
    "magic_csr_gather": [i32, i32, i32, vp, vp, vp, vp, vp, i32, vp],
    "magic_pano_fuse_fwd": [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp],
    "magic_pano_fuse_bwd": [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "magic_sap_fuse_fwd": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp],
    "magic_sap_fuse_bwd": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp],
    "magic_sumsq": [i64, vp, vp, vp],
    "magic_adamw": [i64, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, vp, f32, f32, vp, vp],
    "magic_sched_step": [vp, f32, i32, i32, f32, f32, vp, vp],
    "magic_cast": [i32, i64, vp, vp, vp],
    "magic_add": [i32, i64, vp, vp, vp],
    "magic_dact": [i32, i32, i64, vp, vp, vp, vp],
}



class DwDesc(C.Structure):
    """mirror of `magic_dw_desc` (include/magic_hip.h)"""
    _fields_ = [("dY", vp), ("X", vp), ("dW", vp), ("db", vp), ("M", i32), ("N", i32), ("K", i32),
                ("lda", i32), ("ldb", i32), ("ldc", i32), ("splitk", i32)]


_ERR = {-1: "MAGIC_ERR_ARG", -2: "MAGIC_ERR_LAUNCH", -3: "MAGIC_ERR_UNSUPPORTED"}
_lib = None


class MagicHipError(RuntimeError):
    pass


def load():
    """Load the shared library (fails loudly; there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MagicHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    _lib = lib
    return lib


def stream():
    return torch.cuda.current_stream().cuda_stream


def P(t):
    """device pointer of a tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


PROFILE = {"on": False, "events": []}     # bench.py: per-launch HIP-event timing on the launch stream


def call(name, *args):
    if PROFILE["on"]:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(load(), name)(*args)
        e1.record()
        PROFILE["events"].append((name, args[1] if name == "magic_gemm" else -1, e0, e1))
    else:
        rc = getattr(load(), name)(*args)
    if rc != 0:
        raise MagicHipError(f"{name} failed: {_ERR.get(rc, rc)}")


def dt(dtype):
    if dtype == torch.float32:
        return 0
    if dtype == torch.bfloat16:
        return 1
    raise MagicHipError(f"unsupported compute dtype {dtype}")
