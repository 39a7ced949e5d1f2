"""The C ABI driven from C: tests/abi_smoke.c is compiled by gcc against include/rrrmc_hip.h (-Wall -Wextra -Werror) and linked to the
library, then run on the GPU; every number it prints is compared with the CPU oracle (small instances of BASELINE configs 2, 3, 5, and
the same jobs through rrrmc_ctx_create_multi with two shards)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "rrrmc.jl_amd", "lib")


def build_smoke(tmp):
    exe = os.path.join(tmp, "abi_smoke")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_smoke.c"),
           "-o", exe, "-L", LIBDIR, "-lrrrmc_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_abi_smoke_compiles_against_the_header(pkg, tmp_path):
    """CPU part: the header is valid C11 on its own and the program links against the in-tree library (no GPU needed for that)"""
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pkg.lib()                                   # makes sure the library is built
    assert os.path.exists(build_smoke(str(tmp_path)))


@pytest.mark.gpu
def test_abi_smoke_matches_the_oracle(pkg, oracle, tmp_path):
    seed = 20241
    Xq = pkg.GraphQuant(pkg.GraphRRG(32, 3, seed=seed), 4, 0.5, 2.0)
    exe = build_smoke(str(tmp_path))
    out = subprocess.run([exe, str(seed), repr(Xq.fourK)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = {}
    for ln in out.stdout.splitlines():
        k, *v = ln.split()
        rows[k] = v
    assert rows["bad_create"][1] == "1" and int(rows["bad_create"][3]) > 10          # RRRMC_ERR_INVALID_ARG with a message
    assert rows["c2_multi_equal"] == ["1"] and rows["c5_multi_equal"] == ["1"]
    i64 = lambda k: np.array([int(x) for x in rows[k]], np.int64)
    f64 = lambda k: np.array([float.fromhex(x) for x in rows[k]], np.float64)
    # ---- config 2 small
    N, K, R, iters, step = 256, 3, 96, 4096, 512
    A = oracle.gen_rrg(N, K, seed)
    J = oracle.gen_couplings(A, seed)
    C0 = oracle.init_configs(seed, 0, R, N)
    Es_ref, ch_ref, acc_ref = oracle.standard_mc_sparse_batch(A, J, 1.0, iters, step, seed, C0)
    assert (i64("c2_E0") == [oracle.sparse_energy(A, J, C0[r]) for r in range(R)]).all()
    assert (i64("c2_Es").reshape(R, -1) == Es_ref).all() and (i64("c2_acc") == acc_ref).all()
    assert (np.array([int(x) for x in rows["c2_C1"]], np.uint64).reshape(R, -1) == ch_ref).all()
    assert (i64("c2_E1") == [oracle.sparse_energy(A, J, ch_ref[r]) for r in range(R)]).all()
    # ---- config 3 small
    N, R, iters, step = 64, 16, 2048, 256
    Jm = oracle.gen_sk_gauss(N, seed)
    C0 = oracle.init_configs(seed, 0, R, N)
    Es_ref, ch_ref, acc_ref, _ = oracle.standard_mc_skn_batch(Jm, 1.0, iters, step, seed, C0)
    assert (f64("c3_Es").reshape(R, -1) == Es_ref).all() and (i64("c3_acc") == acc_ref).all()
    assert (f64("c3_E1") == [oracle.skn_energy(Jm, ch_ref[r]) for r in range(R)]).all()
    # ---- config 5 small
    Nk, M, R, iters, step = 32, 4, 40, 3000, 500
    A = oracle.gen_rrg(Nk, 3, seed)
    J = oracle.gen_couplings(A, seed)
    C0 = oracle.init_configs(seed, 0, R, Nk * M)
    Es, acc, staged = f64("c5_Es").reshape(R, -1), i64("c5_acc"), i64("c5_staged")
    for r in (0, 31, 32, 39):                    # both shards of the multi-device run (replicas 0-31 and 32-39)
        ref = oracle.rrr_mc_quant(A, J, M, Xq.fourK, 2.0, iters, step, seed, C0[r], replica=r)
        assert (Es[r] == ref[0]).all() and acc[r] == ref[2] and staged[r] == ref[3]
