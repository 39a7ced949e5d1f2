"""Randomised parity check (GPU box) of standardMC on +-J GraphRRG / GraphEA — the headline path in all its builds (table in LDS / in
HBM, one word per site, HBM-resident state) — against the oracle: random sizes, degrees, temperatures, call lengths and sample steps,
two calls per case (the second continues the streams), a few replicas per case compared bit for bit (energies, accepted counts, final
configuration).  python tools/stress_sweep.py [cases] [seed] [big]
With `big` every case runs the big-N kernels (bign_kernels.hpp): forced at small N, natural at N = 35 000 .. 140 000 (16, 8 and 4
replicas per workgroup of big_apply_kernel), alternating between big_mask_kernel + big_apply_kernel and big_sweep_kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
O = e.load_oracle()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
big_only = len(sys.argv) > 3 and sys.argv[3] == "big"
bad = 0
for case in range(ncases):
    seed = int(rng.integers(1, 1 << 30))
    for v in ("RRRMC_FORCE_WIDE", "RRRMC_FORCE_SINGLE", "RRRMC_FORCE_BIG", "RRRMC_BIG_NO_MASKS"):
        os.environ.pop(v, None)
    if case % 3 == 2:
        Lx, D = [(4, 3), (6, 3), (8, 3), (16, 2), (10, 3), (5, 2)][int(rng.integers(0, 6))]
        X, form, desc = pkg.GraphEA(Lx, D, seed=seed), "ea", "GraphEA(%d, %d)" % (Lx, D)
    else:
        K = int(rng.choice([3, 3, 3, 4, 5, 6]))       # (gen_RRG's pairing retries rarely succeed beyond K = 6)
        N = int(rng.choice([64, 200, 1024, 2500, 4096, 6000, 9000, 12000] + ([35000, 70000, 140000] if big_only else []))) // 2 * 2
        X, form, desc = pkg.GraphRRG(N, K, seed=seed), "rrg", "GraphRRG(%d, %d)" % (N, K)
    force = rng.random()
    if big_only:
        os.environ["RRRMC_FORCE_BIG"] = "1"; desc += " big"
        if case % 2:
            os.environ["RRRMC_BIG_NO_MASKS"] = "1"; desc += " nomasks"
    elif force < 0.15:
        os.environ["RRRMC_FORCE_WIDE"] = "1"; desc += " wide"
    elif force < 0.3 and X.N <= 32767:
        os.environ["RRRMC_FORCE_SINGLE"] = "1"; desc += " single"
    elif force < 0.4:
        os.environ["RRRMC_FORCE_BIG"] = "1"; desc += " big"
    beta = float(rng.choice([0.3, 1.0, 2.0]))
    R = int(rng.choice([32, 33, 64, 96]))
    iters = int(rng.integers(500, 40000)) * (8 if big_only and X.N > 30000 else 1)
    step = int(rng.choice([1, 7, 100, 1000, 4096, iters]))
    step = max(1, min(step, iters))
    iters2 = int(rng.integers(1, 5000))
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config().s.copy()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config().s.copy()
        Es2, acc2 = eng.standard_mc(beta, iters2, max(1, iters2 // 3))
        C2 = eng.get_config().s.copy()
    A, J = X.A, X.J.astype(np.int32)
    ok = True
    for r in sorted({0, int(rng.integers(0, R)), R - 1}):
        ref = O.standard_mc_sparse(A, J, beta, iters, step, seed, C0[r], replica=r, form=form)
        ok &= bool((Es[r] == ref[0]).all() and (C1[r] == ref[1]).all() and acc[r] == ref[2])
        ref2 = O.standard_mc_sparse(A, J, beta, iters2, max(1, iters2 // 3), seed, ref[1], it0=iters, replica=r, form=form)
        ok &= bool((Es2[r] == ref2[0]).all() and (C2[r] == ref2[1]).all() and acc2[r] == ref2[2])
    bad += 0 if ok else 1
    print("%-3d %-28s R=%d beta=%.1f iters=%d step=%d +%d  %s" % (case, desc, R, beta, iters, step, iters2, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
