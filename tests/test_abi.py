"""C-ABI checks that need no GPU: the library loads, exports every symbol include/magic_hip.h declares,
and the ctypes signatures in host/lib.py agree with the header argument by argument."""
import ctypes
import os
import re

import pytest

import magic_amd  # noqa: F401
from magic_amd.host import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_header():
    src = open(os.path.join(ROOT, "include", "magic_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(magic_\w+)\s*\(([^)]*)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    types.append(L.vp)
                elif a.startswith("long long"):
                    types.append(L.i64)
                elif a.startswith("float"):
                    types.append(L.f32)
                elif a.startswith("unsigned long long"):
                    types.append(L.u64)
                elif a.startswith("unsigned"):
                    types.append(L.u32)
                elif a.startswith("int"):
                    types.append(L.i32)
                else:
                    raise AssertionError(f"unparsed arg {a!r} in {name}")
        protos[name] = types
    return protos


def test_header_and_ctypes_signatures_agree():
    protos = parse_header()
    assert set(protos) == set(L.SIGNATURES)
    for name, types in protos.items():
        assert types == L.SIGNATURES[name], name


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in parse_header():
        assert hasattr(lib, name), name
    assert lib.magic_abi_version() == 1


def test_library_carries_the_id_of_this_source_tree(monkeypatch):
    """magic_build_id() == csrc/build_id.py's content hash of the sources: the binary under test is the binary of this tree; a library
    whose id differs is refused at load"""
    L.load()
    assert L.library_build_id() == L.source_build_id() and len(L.library_build_id()) == 16
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "source_build_id", lambda: "0" * 16)
    with pytest.raises(L.MagicHipError, match="other sources"):
        L.load()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libmagic_hip.so")
    with pytest.raises(L.MagicHipError):
        L.load()


def test_fastcall_extension_binds_the_same_library_and_checks_arity():
    """the generated CPython marshalling layer (csrc/gen_fastcall.py): built, bound to the SAME libmagic_hip.so, one wrapper per
    entry point without struct arguments, argument count and types enforced before the C call"""
    L.load()
    assert os.path.exists(L.FAST_PATH), "csrc/Makefile builds _magic_fastcall.so next to libmagic_hip.so"
    fast = {n for n, f in L._FN.items() if type(f).__name__ == "builtin_function_or_method"}
    slow = set(L.SIGNATURES) - fast
    assert slow == {"magic_device_info", "magic_build_id", "magic_gemm_dw_grouped", "magic_gemm_dw_ws_need", "magic_mse_multi"}, slow
    assert L._FN["magic_abi_version"]() == L.load().magic_abi_version() == 1
    for args in ((1, 64, 80, 80), (0, 64, 600, 80), (1, 48, 80, 80), (1, 64, 5000, 80)):       # pure host function: both bindings agree
        assert L._FN["magic_attn_supported"](*args) == L.load().magic_attn_supported(*args)
    with pytest.raises(TypeError):
        L._FN["magic_attn_supported"](1, 64, 80)
    with pytest.raises(TypeError):
        L._FN["magic_attn_supported"](1, 64, 80, "80")
    n = len(L.SIGNATURES["magic_gemm"])
    assert L._FN["magic_gemm"](*([1, 0, 1, 1, 0, 128, 128] + [None if L.SIGNATURES["magic_gemm"][i] is L.vp else 0 for i in range(7, n)])) == -1   # M = 0 -> MAGIC_ERR_ARG
