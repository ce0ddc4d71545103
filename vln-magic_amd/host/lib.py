"""ctypes binding of libmagic_hip.so (the C ABI of include/magic_hip.h).

The product path has NO CPU fallback: if the library is missing or a kernel returns an error code the
call raises.  Tensors are passed as raw device pointers; the stream is torch's current HIP stream.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MAGIC_LIB_FILE: load another build of the library (same-box A/B of a kernel change: `MAGIC_LIB_FILE=.../libmagic_hip_prev.so
# MAGIC_ALLOW_STALE_LIB=1 python bench.py ...`); the build-id check still runs against whatever is loaded
LIB_PATH = os.environ.get("MAGIC_LIB_FILE") or os.path.join(os.path.dirname(_HERE), "libmagic_hip.so")

vp, i32, i64, f32, u32, u64 = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_uint, C.c_ulonglong

# name -> argument ctypes (mirrors include/magic_hip.h exactly; tests/test_abi.py checks every symbol)
SIGNATURES = {
    "magic_abi_version": [],
    "magic_build_id": [vp, i32],
    "magic_device_info": [vp, vp, vp, i32],
    "magic_gemm": [i32, i32, i32, i32, i32, i32, i32, vp, i32, i64, i64, vp, i32, i64, i64, vp, i32, i64, i64, i32, i32,
                   vp, i32, vp, i32, vp, i32, vp, i32, f32, i32, vp, vp],
    "magic_gemm_set_big": [i32],
    "magic_gemm_dw_ws_need": [i32, i32, vp, vp, vp],
    "magic_gemm_dw_grouped": [i32, i32, vp, vp, i64, vp, i32, vp],
    "magic_linear_ln": [i32, i32, i32, i32, vp, i32, vp, i32, vp, vp, i32, vp, vp, f32, vp, vp, vp, f32, u32, vp],
    "magic_linear_act_ln": [i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, f32, vp, vp, vp],
    "magic_linear_lnbwd": [i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, u32, vp],
    "magic_dropout": [i32, i64, i32, i32, vp, vp, vp, f32, u32, vp],
    "magic_ln_fwd": [i32, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, f32, vp, vp, i32,
                     vp, f32, u32, u32, vp, vp],
    "magic_ln_bwd": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, i32, vp, i32, i32, vp, i32,
                     vp, i32, i32, vp, i32, i32, vp, f32, u32, u32, vp, i32, i32, vp],
    "magic_ln_bwd_tail": [i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp],
    "magic_ln_bwd_blocks": [i32, i32, i32],
    "magic_rowbwd_attn_supported": [i32, i32, i32, i32, i32],
    "magic_colsum_add_v": [i32, vp, vp, vp, vp, vp, vp],
    "magic_smallk_ln_bwd_blocks": [i32, i32, i32],
    "magic_ln_pgrad": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp],
    "magic_smallk_ln_fwd": [i32, i32, i32, i32, vp, vp, vp, vp, vp, f32, vp, vp, vp],
    "magic_smallk_ln_bwd": [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "magic_softmax_fwd": [i32, i32, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp],
    "magic_softmax_bwd": [i32, i32, i32, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp],
    "magic_attn_supported": [i32, i32, i32, i32],
    "magic_attn_fwd": [i32, i32, i32, i32, i32, vp, i32, vp, vp, i32, vp, i32, vp, i32, f32, vp, vp, vp, vp, vp, f32, u32, vp, vp],
    "magic_attn_bwd": [i32, i32, i32, i32, i32, vp, i32, vp, vp, i32, vp, i32, vp, i32, f32, vp, vp, i32, vp, vp, i32, vp, vp, vp,
                       vp, f32, u32, vp],
    "magic_attn_bwd_ks": [i32, i32, i32, i32, i32, vp, i32, vp, vp, i32, vp, i32, vp, vp, i32, f32, vp, i32, vp, vp, i32, i32, vp, f32, u32, vp],
    "magic_head_mean_fwd": [i32, i32, i32, i64, vp, vp, vp],
    "magic_head_mean_bwd": [i32, i32, i64, vp, vp, i32, vp],
    "magic_lndot_fwd": [i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp],
    "magic_lndot_bwd": [i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "magic_lndot_bwd_blocks": [i32],
    "magic_ce_rows": [i32, i32, i32, vp, i32, vp, i32, f32, vp, vp, vp, i32, i32, vp, f32, vp],
    "magic_softkl_rows": [i32, i32, i32, vp, i32, vp, i32, f32, vp, vp, vp, i32, vp],
    "magic_kd_rows": [i32, i32, vp, vp, i32, f32, vp, f32, f32, vp, vp, vp, i32, vp],
    "magic_mse": [i32, i32, i64, i64, vp, i64, vp, i64, vp, i64, f32, f32, vp, vp, vp, i64, i32, vp],
    "magic_mse_multi": [i32, i32, vp, vp],
    "magic_step_rng": [u64, vp, f32, vp, vp, vp, vp, f32, f32, i32, vp],
    "magic_seed_scale": [vp],
    "magic_rowgate_fwd": [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "magic_rowgate_bwd": [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "magic_gemm_dw_cat": [i32, i32, vp, i32, vp, vp, vp, vp],
    "magic_loss_assemble": [vp, i32, vp, f32, vp, i32, vp, vp, f32, i32, vp, vp],
    "magic_cfp_loss": [i32, i32, i32, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp],
    "magic_node_in_fwd": [i32, i32, i32, vp, vp],
    "magic_embed_in_fwd": [i32, i32, vp, vp, vp],
    "magic_embed_in_bwd_supported": [i32, i32],
    "magic_embed_in_bwd_blocks": [i32, i32, i32, i32],
    "magic_embed_in_bwd": [i32, i32, vp, vp, i32, vp, vp, vp, vp],
    "magic_csr_gather_multi": [i32, i32, i32, vp, vp],
    "magic_smallk_ln_bwd_pair": [i32, i32, vp, vp],
    "magic_csr_gather": [i32, i32, i32, vp, vp, vp, vp, vp, i32, vp],
    "magic_pano_fuse_fwd": [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp],
    "magic_sap_fuse_loss": [vp, i32, vp],
    "magic_pano_fuse_bwd": [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "magic_pano_fuse_bwd_blocks": [i32],
    "magic_sap_fuse_fwd": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp],
    "magic_sap_fuse_bwd": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp],
    "magic_sumsq": [i64, vp, vp, vp],
    "magic_adamw": [i64, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, f32, f32, vp, f32, f32, vp, i64, i32, vp, vp, vp, i32, vp],
    "magic_sumsq_sched": [i64, vp, vp, vp, f32, i32, i32, f32, f32, vp, vp],
    "magic_sched_step": [vp, f32, i32, i32, f32, f32, vp, vp, vp],
    "magic_add_n": [i32, i64, i32, vp, vp, vp],
    "magic_cast": [i32, i32, i64, vp, vp, vp],
    "magic_add": [i32, i64, vp, vp, vp],
    "magic_dact": [i32, i32, i64, vp, vp, vp, vp],
    "magic_view_gather": [i32, i32, i32, i32, vp, i32, vp, vp, vp, vp],
    "magic_set_f32_mfma": [i32],
    "magic_get_f32_mfma": [],
    "magic_encoder_supported": [i32, i32, i32, i32, i32, i32],
    "magic_encoder_params_bytes": [],
    "magic_encoder_fwd": [i32, vp, i32, vp],
    "magic_encoder_start_gate": [i32, i32, vp, vp],
    "magic_encoder_health": [vp, vp],
    "magic_stream_probe": [vp, i32, vp, vp],
    "magic_chain_supported": [i32, i32, i32],
    "magic_chain_fwd": [i32, vp, i32, vp],
    "magic_chain_tile_rows": [i32],
    "magic_xencoder_supported": [i32, i32, i32, i32, i32, i32, i32],
    "magic_xencoder_params_bytes": [],
    "magic_xencoder_fwd": [i32, vp, i32, vp],
    "magic_rowbwd_supported": [i32, i32, i32],
    "magic_rowbwd_params_bytes": [],
    "magic_rowbwd": [i32, vp, i32, vp],
    "magic_rowbwd_rows": [i64],
    "magic_colsum_add": [i32, i32, vp, vp, vp, vp],
    "magic_transpose_spans": [vp, vp, i32, vp, vp, vp, vp],
    "magic_pack_frag_spans": [vp, vp, i32, vp, vp, vp, vp],
    "magic_layout_spans": [vp, vp, vp, i32, vp, vp, vp, vp, vp],
    "magic_group_begin": [],
    "magic_group_end": [vp],
}



class DwDesc(C.Structure):
    """mirror of `magic_dw_desc` (include/magic_hip.h)"""
    _fields_ = [("dY", vp), ("X", vp), ("dW", vp), ("db", vp), ("M", i32), ("N", i32), ("K", i32),
                ("lda", i32), ("ldb", i32), ("ldc", i32), ("splitk", i32)]


class DwCatProb(C.Structure):
    """mirror of `magic_dwcat_prob` (include/magic_hip.h)"""
    _fields_ = [("dW", vp), ("db", vp), ("N", i32), ("K", i32), ("lda", i32), ("ldb", i32), ("ldc", i32)]


class MseDesc(C.Structure):
    """mirror of `magic_mse_desc` (include/magic_hip.h)"""
    _fields_ = [("g_f32", i32), ("outer", i64), ("inner", i64), ("s", vp), ("s_stride", i64), ("t", vp), ("t_stride", i64),
                ("w", vp), ("rows_per_w", i64), ("norm", f32), ("coef", f32), ("coef_dev", vp), ("loss", vp), ("ds", vp),
                ("g_stride", i64), ("accumulate", i32), ("valid_dev", vp), ("norm_dev", vp), ("valid_mod", i64)]


NODE_IN_PTRS = ("A", "rstd", "out", "add0", "src1", "ptr1", "idx1", "w1", "src2", "ptr2", "idx2", "w2", "tab", "tab_idx")


class NodeIn(C.Structure):
    """mirror of `magic_node_in` (include/magic_hip.h)"""
    _fields_ = ([("M", i32), ("Kin", i32), ("x", vp), ("W", vp), ("b", vp), ("gamma", vp), ("beta", vp), ("eps", f32), ("pad_", i32)]
                + [(n, vp) for n in NODE_IN_PTRS])


class DropD(C.Structure):
    """mirror of `magic_drop_desc`"""
    _fields_ = [("seed", vp), ("site", u32), ("p", f32)]


class PanoIn(C.Structure):
    """mirror of `magic_pano_in` (include/magic_hip.h)"""
    _fields_ = [("M", i32), ("Kin", i32), ("eps", f32), ("pad_", i32)] + \
               [(n, vp) for n in ("P0", "g1", "b1", "A1", "rstd1", "loc", "W", "b", "g2", "b2", "A2", "rstd2", "nav_tab", "nav_idx", "tok_tab",
                                  "g3", "b3", "X0", "rstd3", "X0d")] + [("dout", DropD)]


class LnIn(C.Structure):
    """mirror of `magic_ln_in` (include/magic_hip.h)"""
    _fields_ = [("M", i32), ("do_ln", i32), ("in0", vp), ("in1", vp), ("tab", vp * 3), ("idx", vp * 3), ("mod", i32 * 3), ("off", i32 * 3),
                ("gamma", vp), ("beta", vp), ("eps", f32), ("pad_", i32), ("out", vp), ("rstd", vp),
                ("drop_seed", vp), ("drop_p", f32), ("site_in0", u32), ("site_out", u32), ("pad2_", u32), ("out_drop", vp)]


class PanoInBwd(C.Structure):
    """mirror of `magic_pano_in_bwd` (include/magic_hip.h)"""
    _fields_ = [("M", i32), ("Kin", i32), ("pad0_", i32), ("pad1_", i32), ("dy", vp), ("ddy", DropD)] + \
               [(n, vp) for n in ("X0", "rstd3", "g3", "b3", "dg3", "db3", "nav_idx", "d_nav", "d_tok", "A1", "rstd1", "g1", "b1", "dg1", "db1", "dP0",
                                  "A2", "rstd2", "g2", "b2", "dg2", "db2", "loc", "dW", "dbl", "part")]


class LnBwdIn(C.Structure):
    """mirror of `magic_ln_bwd_in` (include/magic_hip.h)"""
    _fields_ = [("M", i32), ("do_ln", i32)] + [(n, vp) for n in ("dy", "y", "gamma", "beta", "rstd", "dx", "dgamma", "dbeta")] + \
               [("idx", vp * 3), ("mod", i32 * 3), ("off", i32 * 3), ("d", vp * 3), ("small", i32 * 3),
                ("drop_seed", vp), ("drop_p", f32), ("site_dy", u32), ("site_dx", u32), ("hot0", i32), ("dxm", vp), ("partial", i32)]


class CsrProb(C.Structure):
    """mirror of `magic_csr_prob` (include/magic_hip.h)"""
    _fields_ = [("n_out", i32), ("accumulate", i32), ("src1", vp), ("ptr1", vp), ("idx1", vp), ("w1", vp),
                ("src2", vp), ("ptr2", vp), ("idx2", vp), ("w2", vp), ("out", vp)]


class SkbProb(C.Structure):
    """mirror of `magic_skb_prob` (include/magic_hip.h)"""
    _fields_ = [("M", i32), ("Kin", i32)] + [(n, vp) for n in ("x", "dy", "y", "gamma", "beta", "rstd", "dW", "db", "dgamma", "dbeta", "part")]


class EncLayer(C.Structure):
    """mirror of `magic_enc_layer` (include/magic_hip.h)"""
    _fields_ = [(n, vp) for n in ("Wqkv", "bqkv", "Wo", "bo", "g1", "be1", "W1", "bi", "W2", "bo2", "g2", "be2",
                                  "qkv", "P", "Pd", "ctx", "a", "z", "g", "out", "rstd_a", "rstd_o")] + \
               [("site_attn", u32), ("site_ao", u32), ("site_out", u32), ("pad_", u32)]


class EncSeg(C.Structure):
    _fields_ = [("x", vp), ("kmask", vp), ("nsamp", i32), ("N", i32), ("ldp", i32), ("nlayers", i32), ("L", EncLayer * 6)]


class EncParams(C.Structure):
    _fields_ = [("seg", EncSeg * 2), ("nseg", i32), ("p_attn", f32), ("p_hidden", f32), ("eps", f32), ("scale", f32), ("seed", vp),
                ("sync", vp), ("sync_words", i32), ("pad2_", i32)]


class SapLossParams(C.Structure):
    """mirror of `magic_sap_loss_params` (include/magic_hip.h)"""
    _fields_ = [("B", i32), ("K", i32), ("Vp", i32), ("use_gate", i32)] + \
               [(n, vp) for n in ("g_raw", "l_raw", "fuse_raw", "gmask", "lmask", "fsrc", "bwmask", "gl", "ll", "fl", "glab", "llab")] + \
               [("ignore_index", i32), ("coef", f32)] + [(n, vp) for n in ("rows", "dgl", "dll", "dfl", "t_fused")] + \
               [("w_rate", f32), ("pad_", i32), ("w_out", vp), ("T", f32), ("kd_norm", f32), ("kd_coef", f32), ("pad2_", f32), ("kd_coef_dev", vp), ("kd_rows", vp)]


class ChainParams(C.Structure):
    """mirror of `magic_chain_params` (include/magic_hip.h)"""
    _fields_ = [("M", i32), ("ld_in", i32), ("Np", i32), ("pad_", i32)] + \
               [(n, vp) for n in ("inp", "res", "Wa", "ba", "g1", "b1", "y1", "W1", "bi", "W2", "bo2", "g2", "b2", "y2", "Wp", "bp", "proj")] + \
               [("eps", f32), ("pad2_", i32)]


XL_PTRS = ("Wqkv", "bqkv", "Wo", "bo", "g1", "be1", "Wq", "bq", "Wkv", "bkv", "Woc", "boc", "gc", "bec", "W1", "bi", "W2", "bo2", "g2", "be2",
           "qkv", "P", "Pd", "ctx", "a", "rstd_a", "q", "kv", "Pc", "Pdc", "cctx", "c", "rstd_c", "z", "g", "out", "rstd_o")


class XLayer(C.Structure):
    """mirror of `magic_xenc_layer` (include/magic_hip.h)"""
    _fields_ = [(n, vp) for n in XL_PTRS] + [(n, u32) for n in ("site_attn", "site_ao", "site_cattn", "site_co", "site_out", "pad_")]


class XSeg(C.Structure):
    _fields_ = [(n, vp) for n in ("x", "cx", "qmask", "cmask", "dist", "sprel_w", "sprel_b")] + \
               [(n, i32) for n in ("nsamp", "Nq", "Nk", "ldps", "ldpc", "nlayers")] + [("L", XLayer * 3)]


class XParams(C.Structure):
    _fields_ = [("seg", XSeg * 2), ("nseg", i32), ("p_attn", f32), ("p_hidden", f32), ("eps", f32), ("scale", f32), ("seed", vp),
                ("sync", vp), ("sync_words", i32), ("pad2_", i32)]


RBW_PTRS = ("dqkv_n", "WqkvT_n", "dao_n", "dfo_in", "dfod_in", "y2", "rstd2", "g2", "b2", "dg2", "db2", "z", "W2T", "W1T",
            "y1", "rstd1", "g1", "b1", "dg1", "db1", "WoT", "dfo", "dfod", "dz", "daod", "dao", "dctx")


RBW_DIST_PTRS = ("dist", "dsprel_w", "dsprel_b")          # the map encoder's graph-distance bias: its two gradients come out of the in-launch attention backward
RBW_ATT_PTRS = ("qkv_a", "P_a", "o_a", "dctx_a", "dP_init", "dqkv_out")      # round 6: the attention backward of the block above inside the launch


class RbwSeg(C.Structure):
    """mirror of `magic_rowbwd_seg` (include/magic_hip.h)"""
    _fields_ = ([("M", i32), ("kt", i32)] + [(n, vp) for n in RBW_PTRS] + [("site_out", u32), ("site_ao", u32)] +
                [("mode", i32), ("N", i32), ("ntile", i32), ("ldp", i32)] + [(n, vp) for n in RBW_ATT_PTRS] + [("site_attn", u32), ("pad_", u32)] +
                [(n, vp) for n in RBW_DIST_PTRS])


class RbwParams(C.Structure):
    _fields_ = [("seg", RbwSeg * 2), ("nseg", i32), ("blocks0", i32), ("p_hidden", f32), ("pad1", i32), ("seed", vp), ("p_attn", f32), ("scale", f32)]


_ERR = {-1: "MAGIC_ERR_ARG", -2: "MAGIC_ERR_LAUNCH", -3: "MAGIC_ERR_UNSUPPORTED"}
_lib = None


class MagicHipError(RuntimeError):
    pass


def source_build_id():
    """the id csrc/build_id.py derives from THIS tree's sources (content hashes; None when the sources are not there)"""
    path = os.path.join(os.path.dirname(_HERE), "csrc", "build_id.py")
    if not os.path.exists(path):
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("_magic_build_id", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_id()


def library_build_id(lib=None):
    """the id compiled into the loaded libmagic_hip.so (magic_build_id)"""
    lib = lib if lib is not None else load()
    buf = C.create_string_buffer(32)
    if lib.magic_build_id(buf, 32) != 0:
        raise MagicHipError("magic_build_id failed")
    return buf.value.decode()


def check_build_id(lib):
    """a library built from other sources than the tree it sits in is refused (MAGIC_ALLOW_STALE_LIB=1 overrides, for bisecting)"""
    want, got = source_build_id(), library_build_id(lib)
    if want is not None and want != got and not os.environ.get("MAGIC_ALLOW_STALE_LIB"):
        raise MagicHipError(f"{LIB_PATH} was built from other sources (library id {got}, this tree {want}): rebuild with "
                            "`python __graft_entry__.py` (make -C vln-magic_amd/csrc)")


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
        fn = getattr(lib, name, None)
        if fn is None:
            raise MagicHipError(f"{LIB_PATH} does not export {name}: a stale build -- run `python __graft_entry__.py`")
        fn.argtypes = args
        fn.restype = i32
    check_build_id(lib)
    _lib = lib
    _bind_fast(lib)
    return lib


FAST_PATH = os.environ.get("MAGIC_FASTCALL_PATH") or os.path.join(os.path.dirname(LIB_PATH), "_magic_fastcall.so")     # (env: the sanitizer build, csrc/Makefile `asan`)
_FN = {}          # entry point name -> callable: the generated CPython wrapper when there is one, else the ctypes function


def _fn(name):
    f = _FN.get(name)
    if f is None:
        load()                     # first use: loads the library and fills the table (fails loudly if it is missing)
        f = _FN[name]
    return f


def _bind_fast(lib):
    """`_magic_fastcall` (csrc/gen_fastcall.py -> fastcall.c, built by the same Makefile) converts the plain int / float / pointer
    arguments of a launch in 0.2-0.6 us where ctypes takes 0.5-2.6 us; it calls the SAME loaded libmagic_hip.so.  Entry points with
    struct arguments, and a tree without the extension (MAGIC_NO_FASTCALL=1, or not built), go through ctypes -- same library either way."""
    for name in SIGNATURES:
        _FN[name] = getattr(lib, name)
    if os.environ.get("MAGIC_NO_FASTCALL") or not os.path.exists(FAST_PATH):
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location("_magic_fastcall", FAST_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.bind(LIB_PATH)
    for name in SIGNATURES:
        f = getattr(mod, name, None)
        if f is not None:
            _FN[name] = f


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """raw handle of torch's current HIP stream on the current device (thread-local, honours torch.cuda.stream contexts).
    `torch.cuda.current_stream()` builds a Stream object and re-resolves the device on every call (~8 us): with ~400 launches
    per eager step that was a third of the host time."""
    if _raw_stream is not None:
        return _raw_stream(torch._C._cuda_getDevice())
    return torch.cuda.current_stream().cuda_stream


def P(t):
    """device pointer of a tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


PROFILE = {"on": False, "events": []}     # bench.py: per-launch HIP-event timing on the launch stream
PAIRABLE = {"magic_gemm", "magic_attn_fwd", "magic_attn_bwd", "magic_linear_ln", "magic_linear_lnbwd", "magic_ln_bwd", "magic_chain_fwd", "magic_ln_fwd"}
# calls lockstep cannot pair into one kernel but whose twin of the partner segment is independent and as long as the call itself (the key-split
# attention backward of the two cross-modal encoders: one 512-thread workgroup per (sample, head), 157 KB of LDS, 48 us of single-workgroup latency):
# when both threads have arrived at one, the partner's goes to the segment pair's side stream and runs BESIDE this thread's.  OPT-IN
# (MAGIC_LOCKSTEP_FORK=1): measured on the MAGIC-L navigator iteration it LOSES -- 143-193 ms against 129-148 on the same box
# (profiles/micro/ab_ks_fork_r05.sh): the fork / join edges inside the step graphs cost more than the half round of workgroups they hide
FORKABLE = {"magic_attn_bwd_ks"} if os.environ.get("MAGIC_LOCKSTEP_FORK", "0") == "1" else set()
_tls = threading.local()


_DEBUG_SYNC = bool(os.environ.get("MAGIC_DEBUG_SYNC"))     # print + synchronize around every launch (localises a faulting kernel)


def _raw_call(name, args):
    if getattr(_tls, "in_group", False) and name in PAIRABLE:
        rc = _fn(name)(*args)             # recorded by the C side, launched by magic_group_end
        if rc != 0:
            raise MagicHipError(f"{name} failed while recording a group: {_ERR.get(rc, rc)}")
        return
    if _DEBUG_SYNC:
        print("[magic]", name, flush=True)
        rc = _fn(name)(*args)
        torch.cuda.synchronize()
        if rc != 0:
            raise MagicHipError(f"{name} failed: {_ERR.get(rc, rc)}")
        return
    if PROFILE["on"]:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = _fn(name)(*args)
        e1.record()
        PROFILE["events"].append((name, args[1] if name == "magic_gemm" else -1, e0, e1))
        if PROFILE.get("shapes") is not None:       # profiles/micro/nav_kernel_breakdown.py: leading integer arguments (dtype, sizes) per launch
            PROFILE["shapes"].append(tuple(a for a in args[:7] if isinstance(a, int) and abs(a) < (1 << 24)))
    else:
        rc = _fn(name)(*args)
    if rc != 0:
        raise MagicHipError(f"{name} failed: {_ERR.get(rc, rc)}")


class solo:
    """`with L.solo():` -- inside a lockstep segment, launch this thread's groupable calls alone instead of offering them to the partner:
    keeps two segments whose launch sequences differ by a prefix (the panorama encoder's image projection) in phase, so that their layers
    pair kind for kind (QKV with QKV, attention with attention, chain with chain)"""

    def __enter__(self):
        self.prev = getattr(_tls, "solo", False)
        _tls.solo = True
        return self

    def __exit__(self, et, ev, tb):
        _tls.solo = self.prev
        return False


def call(name, *args):
    ls = getattr(_tls, "lockstep", None)
    if ls is not None and (name in PAIRABLE or (name in FORKABLE and ls.side is not None)) and not getattr(_tls, "solo", False):
        return ls.submit(ls.index(), name, args)
    if ls is not None and PROFILE["on"]:
        with ls.cv:            # instrumented pass: keep this launch's event pair free of the partner thread's launches
            return _raw_call(name, args)
    _raw_call(name, args)


class group:
    """`with L.group():` -- the groupable calls (PAIRABLE) issued by THIS thread inside the block are recorded and launched
    together at exit: independent GEMMs of one layout become one grouped launch (<= 8 per block).  The calls must not depend
    on each other, and nothing inside the block may read their outputs.  No-op inside a lockstep segment."""

    def __enter__(self):
        self.on = getattr(_tls, "lockstep", None) is None and not _DEBUG_SYNC
        if self.on:
            if PROFILE["on"]:
                self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                self.ev[0].record()
            if load().magic_group_begin() != 0:
                raise MagicHipError("magic_group_begin failed (nested grouping?)")
            _tls.in_group = True
        return self

    def __exit__(self, et, ev, tb):
        if self.on:
            _tls.in_group = False
            rc = load().magic_group_end(stream())
            if PROFILE["on"]:
                self.ev[1].record()
                PROFILE["events"].append(("magic_gemm+group", 0, self.ev[0], self.ev[1]))
            if rc != 0 and et is None:
                raise MagicHipError(f"magic_group_end failed: {_ERR.get(rc, rc)}")
        return False


class Lockstep:
    """Run two independent, same-structured segments (e.g. the global and the local co-attention encoder) on two host
    threads and issue their groupable kernel calls PAIRWISE: when both threads have arrived at a groupable call, the second
    arriver records both through magic_group_begin/…/magic_group_end, which launches ONE kernel serving both problems.
    Non-groupable calls launch immediately.  If one segment finishes first (or raises) the other simply continues alone.

    The two threads ALTERNATE, they never run at the same time: each holds `cv` for as long as it runs (lib.lockstep) and lets go of it only
    while it waits for its partner in `submit`.  The segments are issued into a stream that is being captured, and two threads adding
    nodes to one capturing stream at once can lose one of them from the stream's dependency chain (both read the same last node): the
    capture then ends with hipErrorStreamCaptureUnjoined -- seen once in some hundred captures before the baton."""

    def __init__(self, side=None):
        self.cv = threading.Condition()
        self.pending = [None, None]
        self.done = [False, False]
        self.gen = 0
        self.pairs = 0
        self.forks = 0
        self.side = side              # a stream for the partner's half of a FORKABLE twin (None: forkable calls launch where they are)

    def index(self):
        return _tls.idx

    def _meet(self, idx, name, args):
        """the partner is parked at a call: launch both (one group, forked twins, or one after the other); True when that happened"""
        other = 1 - idx
        if True:
            if self.pending[other] is not None and (name in FORKABLE or self.pending[other][0] in FORKABLE):
                oname, oargs = self.pending[other]
                if name in FORKABLE and oname in FORKABLE:       # twins: the partner's on the side stream, ours here, joined at once
                    cur = torch.cuda.current_stream()
                    self.side.wait_stream(cur)
                    _raw_call(oname, tuple(oargs[:-1]) + (self.side.cuda_stream,))      # (the last argument of every entry point is its stream)
                    _raw_call(name, args)
                    cur.wait_stream(self.side)
                    self.forks += 1
                else:                                            # the segments are out of step here: one after the other, segment 0's first
                    for n_, a_ in (((oname, oargs), (name, args)) if other == 0 else ((name, args), (oname, oargs))):
                        _raw_call(n_, a_)
                self.pending[other] = None
                self.gen += 1
                return True
            if self.pending[other] is not None:                  # partner is waiting: launch both as one group
                oname, oargs = self.pending[other]
                first, second = ((oname, oargs), (name, args)) if other == 0 else ((name, args), (oname, oargs))
                lib = load()
                ev = None
                if PROFILE["on"]:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                if lib.magic_group_begin() != 0:
                    raise MagicHipError("magic_group_begin failed (nested grouping?)")
                try:
                    for n_, a_ in (first, second):
                        rc = _fn(n_)(*a_)
                        if rc != 0:
                            raise MagicHipError(f"{n_} failed while recording a group: {_ERR.get(rc, rc)}")
                finally:
                    rc = lib.magic_group_end(stream())
                if rc != 0:
                    raise MagicHipError(f"magic_group_end failed: {_ERR.get(rc, rc)}")
                if ev is not None:
                    ev[1].record()
                    same = first[0] == second[0]
                    PROFILE["events"].append((first[0] + ("+pair" if same else "+" + second[0]), first[1][1] if first[0] == "magic_gemm" else -1, ev[0], ev[1]))
                self.pairs += 1
                self.pending[other] = None
                self.gen += 1
                return True
        return False

    def submit(self, idx, name, args):
        other = 1 - idx
        with self.cv:
            if self._meet(idx, name, args):
                self.cv.notify_all()
                return
            if self.done[other]:
                _raw_call(name, args)
                return
            self.pending[idx] = (name, args)
            gen = self.gen
            while self.gen == gen and not self.done[other]:
                self.cv.wait()
            if self.pending[idx] is not None:                    # partner finished without pairing: launch alone
                self.pending[idx] = None
                _raw_call(name, args)

    def finish(self, idx):
        with self.cv:
            self.done[idx] = True
            self.cv.notify_all()


try:
    import greenlet as _greenlet
except ImportError:                       # (the threaded form below needs nothing but the standard library)
    _greenlet = None
LOCKSTEP_FORM = os.environ.get("MAGIC_LOCKSTEP", "greenlets" if _greenlet is not None else "threads")


class LockstepOneThread(Lockstep):
    """The same pairing with NO second thread (round 6): the two segments are two greenlets of the calling thread.  A segment that arrives at a groupable call
    whose twin has not arrived parks the call and switches to the other segment; that one runs until it meets the parked call (both go out as one launch, and
    it carries on), parks a call of its own (and switches back) or ends.  One thread adds nodes to the capturing stream, in an order that is a function of the
    two launch sequences alone -- nothing for a baton to protect.  Segments share the thread's torch state (current stream, grad mode): neither changes it
    around a groupable call (the forked twins' side stream is handled inside `_meet`)."""

    def __init__(self, side=None):
        super().__init__(side)
        self.gl = [None, None]

    def index(self):
        return 0 if _greenlet.getcurrent() is self.gl[0] else 1

    def submit(self, idx, name, args):
        other = 1 - idx
        if self._meet(idx, name, args):
            return
        if self.done[other]:
            _raw_call(name, args)
            return
        self.pending[idx] = (name, args)
        self.gl[other].switch()                               # back here once the partner has met this call, parked one of its own, or ended
        if self.pending[idx] is not None:                     # the partner ended without a twin for it: launch alone
            self.pending[idx] = None
            _raw_call(name, args)


def _lockstep_one_thread(fn_a, fn_b, side):
    ls = LockstepOneThread(side)
    box = {}

    def seg(i, fn):
        def run():
            try:
                box[i] = fn()
            finally:
                ls.done[i] = True
        return run
    ls.gl = [_greenlet.greenlet(seg(0, fn_a)), _greenlet.greenlet(seg(1, fn_b))]      # both children of this greenlet: an ended segment returns here
    _tls.lockstep = ls
    try:
        while not (ls.gl[0].dead and ls.gl[1].dead):
            # start / resume the segment that can run: the first, unless it is parked behind a live partner (then the partner is the one mid-flight)
            nxt = 0 if not ls.gl[0].dead and (ls.pending[0] is None or ls.gl[1].dead) else 1
            if ls.gl[nxt].dead:
                nxt = 1 - nxt
            ls.gl[nxt].switch()
    except BaseException:
        for g in ls.gl:                   # unwind the other segment's stack (its `finally` blocks run) before the error leaves
            if g is not None and not g.dead:
                try:
                    g.throw(_greenlet.GreenletExit)
                except BaseException:     # noqa: BLE001
                    pass
        raise
    finally:
        _tls.lockstep = None
    return box[0], box[1]


def lockstep(fn_a, fn_b, side=None):
    """returns (fn_a(), fn_b()) with their groupable launches paired (see Lockstep).  Default: both segments on the calling thread (LockstepOneThread);
    MAGIC_LOCKSTEP=threads (or no `greenlet` module): fn_b on a helper thread bound to the caller's device and current stream.  side: a stream for the
    partner's half of FORKABLE twins."""
    if getattr(_tls, "lockstep", None) is not None:              # no nesting: run sequentially inside an outer lockstep
        return fn_a(), fn_b()
    if LOCKSTEP_FORM == "greenlets" and _greenlet is not None:
        return _lockstep_one_thread(fn_a, fn_b, side)
    ls = Lockstep(side)
    cur = torch.cuda.current_stream()
    dev = torch.cuda.current_device()
    grad = torch.is_grad_enabled()
    box = {}

    def worker():
        try:
            with ls.cv:                # the baton (see Lockstep): runs only while the caller's thread waits in submit() or has finished
                torch.cuda.set_device(dev)
                _tls.lockstep, _tls.idx = ls, 1
                with torch.cuda.stream(cur), torch.set_grad_enabled(grad):
                    box["b"] = fn_b()
        except BaseException as e:     # noqa: BLE001 - re-raised on the caller's thread
            box["err"] = e
        finally:
            _tls.lockstep = None
            ls.finish(1)

    t = threading.Thread(target=worker, name="magic-lockstep")
    _tls.lockstep, _tls.idx = ls, 0
    try:
        with ls.cv:
            t.start()
            a = fn_a()
    finally:
        _tls.lockstep = None
        ls.finish(0)
        t.join()
    if "err" in box:
        raise box["err"]
    return a, box["b"]


class capture(torch.cuda.graph):
    """`torch.cuda.graph` with Python's cyclic garbage collector OFF for the duration of the capture.  torch collects garbage itself right BEFORE a capture
    begins, but allocations inside a long capture can trigger a collection again, and finalizing an unrelated dead object there (a HIP graph of an earlier
    test / an evicted bucket graph, an event) runs HIP calls the capturing stream does not permit: the destructor throws, the process aborts ("Fatal Python
    error: Aborted ... Garbage-collecting", seen once in round 6 in a full GPU suite run, inside StreamStep's capture).  The garbage is collected after the capture ends."""

    def __enter__(self):
        import gc
        self._gc_was = gc.isenabled()
        r = super().__enter__()
        gc.disable()
        return r

    def __exit__(self, *exc):
        import gc
        try:
            return super().__exit__(*exc)
        finally:
            if self._gc_was:
                gc.enable()


def set_f32_mfma(mode):
    """contraction arithmetic of the fp32 storage mode: 'exact' (v_mfma_f32_16x16x4_f32) or 'bf16x3' (split-bf16, three bf16 MFMAs per
    product, ~2^-17 relative error).  Process-wide; returns the previous mode."""
    lib = load()
    prev = "bf16x3" if lib.magic_get_f32_mfma() else "exact"
    want = {"exact": 0, "bf16x3": 1}[mode]
    torch.cuda.synchronize()
    rc = lib.magic_set_f32_mfma(want)
    if rc != 0:
        raise MagicHipError(f"magic_set_f32_mfma failed: {_ERR.get(rc, rc)}")
    return prev


HALF = (torch.bfloat16, torch.float16)     # the two 16-bit storage types: every MFMA kernel exists for both (csrc/common.hpp H16<>)


def dt(dtype):
    if dtype == torch.float32:
        return 0
    if dtype == torch.bfloat16:
        return 1
    if dtype == torch.float16:
        return 2
    raise MagicHipError(f"unsupported compute dtype {dtype}")
