"""Sanitizer coverage of the CPU side (GPU sanitizers are not available on the pool, so: the CPU builds only).

  * the oracle (oracle/rrrmc_oracle.c, 3 000 lines of C that every parity test trusts) rebuilt with -fsanitize=address,undefined
    (`make -C oracle asan`) and driven by the oracle's own contract / model tests and the tape replays;
  * the HIP-free host code of the library — chunk planner and batch cutter, acceptance thresholds, shard bounds, Philox streams
    (rrrmc.jl_amd/csrc/host_plan.hpp, philox.hpp) — compiled with g++ -fsanitize=address,undefined into tests/host_sanitize.cpp and run
    over a sweep of call shapes with the invariants the kernels rely on."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_gcc():
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("no gcc / g++")


def test_host_planning_code_under_asan_and_ubsan(tmp_path):
    _need_gcc()
    exe = str(tmp_path / "host_sanitize")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wno-unknown-pragmas",
                           os.path.join(ROOT, "tests", "host_sanitize.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all invariants hold" in out.stdout


def test_oracle_under_asan_and_ubsan():
    _need_gcc()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    so = os.path.join(ROOT, "oracle", "_build", "librrrmc_oracle_asan.so")
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    if not (os.path.isabs(asan) and os.path.exists(asan)):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, RRRMC_ORACLE_LIB=so, LD_PRELOAD=asan + (":" + ubsan if os.path.isabs(ubsan) and os.path.exists(ubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the oracle's own suites (contract, every model family) and the tape replays, all through the sanitized library
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_oracle_contract.py"), os.path.join(ROOT, "tests", "test_oracle_models.py"),
           os.path.join(ROOT, "tests", "test_tapes.py"), os.path.join(ROOT, "tests", "test_sk_block_emulation.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-3000:] + out.stderr[-3000:])
    assert "passed" in out.stdout and "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
