"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU side (SURVEY section 5 "sanitizer-equivalent
discipline").  Never on a GPU box (the pool has no device-side ASan): everything here is host code.

  * the checker itself: oracle/librts_oracle_asan.so (g++ -fsanitize=address,undefined -fno-sanitize-recover=all), driven through
    the known-answer, golden-vector and capture-branch tests -- the whole miss / shade / aggregate restatement, brute force and BVH;
  * the product's host side: rts_amd/librts_amd_asan.so (hipcc --offload-host-only: the kernels' host stubs without device code,
    clang's sanitizers): mesh builders (rts_mesh.cpp), the host SAH builder (rts_sah.cpp, through rts_build_hierarchy_host), the
    interval plan, the group algebra, argument checking, the error paths of rts_file_mesh (tests/test_host_logic.py).

Each suite runs in a child interpreter with the matching sanitizer runtime preloaded (the interpreter is not instrumented;
leak checking is off: CPython keeps memory until exit).  A finding aborts the child (-fno-sanitize-recover) and fails the test
with the sanitizer's report."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_suite(preload, env_extra, tests, timeout=1500):
    env = {k: v for k, v in os.environ.items() if k not in ("RTS_AMD_LIB", "RTS_ORACLE_LIB")}
    env.update(env_extra)
    env["LD_PRELOAD"] = preload
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    return r.stdout


def test_oracle_under_asan_ubsan():
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    rt = subprocess.run([gxx, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("g++ has no libasan here")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    out = _run_suite(rt, {"RTS_ORACLE_LIB": os.path.join(ROOT, "oracle", "librts_oracle_asan.so")},
                     ["tests/test_oracle_kat.py", "tests/test_golden.py", "tests/test_capture_branches.py"])
    assert " passed" in out


def test_product_host_side_under_asan_ubsan():
    rts = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not rts or not os.path.exists(hipcc):
        pytest.skip("no hipcc / clang asan runtime here")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "rts_amd", "csrc"), "-s", "-j", "6", "asan"])
    out = _run_suite(rts[0], {"RTS_AMD_LIB": os.path.join(ROOT, "rts_amd", "librts_amd_asan.so")}, ["tests/test_host_logic.py"])
    assert " passed" in out
