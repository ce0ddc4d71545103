"""SURVEY 5.2 ("ASAN build of the C-ABI lib in build's own tests"): the two host-side C objects -- csrc/hostplan.c (the navigator planner's index arithmetic over
numpy buffers, bound with ctypes) and the generated csrc/fastcall.c (CPython argument marshalling of every launch) -- rebuilt with AddressSanitizer +
UndefinedBehaviorSanitizer (`make -C vln-magic_amd/csrc asan`) and exercised by the planner and ABI tests in a child interpreter with libasan preloaded.
CPU only, build container only: no GPU code is sanitised (the pool has no GPU sanitizer) and the test skips itself on a box with a GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vln-magic_amd")

CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import pytest
import magic_amd
from magic_amd.host import hostplan, lib
assert hostplan.PATH.endswith("_magic_hostplan_asan.so") and lib.FAST_PATH.endswith("_magic_fastcall_asan.so")
assert hostplan.lib() is not None
lib.load()
maps = open("/proc/self/maps").read()
for need in ("_magic_hostplan_asan.so", "_magic_fastcall_asan.so", "libasan"):
    assert need in maps, need
rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", os.path.join(%r, "tests", "test_navplan_cpu.py"), os.path.join(%r, "tests", "test_abi.py")])
sys.exit(int(rc))
""" % (ROOT, ROOT, ROOT)


def test_host_side_c_objects_run_the_planner_and_abi_tests_clean_under_asan_and_ubsan():
    if os.path.exists("/dev/kfd") or shutil.which("gcc") is None:
        pytest.skip("sanitizer run belongs to the CPU build container")
    libs = [subprocess.run(["gcc", f"-print-file-name={n}"], capture_output=True, text=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    if not all(os.path.isabs(p) and os.path.exists(p) for p in libs):
        pytest.skip("gcc's sanitizer runtimes are not installed")
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "asan"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=":".join(libs),
               ASAN_OPTIONS="detect_leaks=0:alloc_dealloc_mismatch=0:detect_odr_violation=0:abort_on_error=1",       # (CPython / torch allocate for the process's lifetime)
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               MAGIC_HOSTPLAN_PATH=os.path.join(PKG, "_magic_hostplan_asan.so"), MAGIC_FASTCALL_PATH=os.path.join(PKG, "_magic_fastcall_asan.so"))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in r.stdout, tail
