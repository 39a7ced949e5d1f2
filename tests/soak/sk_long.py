"""One-off (GPU box): dense-SK standardMC calls longer than one segment of the blocked kernel (65 536 iterations) against the oracle,
Gaussian and binary couplings, sample steps that straddle segment boundaries.  python tests/soak/sk_long.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
O.build()
bad = 0
for binary, N, R, iters, step in ((False, 256, 9, 200000, 777), (True, 300, 5, 150001, 65536), (False, 1024, 4, 140000, 1000), (True, 64, 12, 131073, 1)):
    seed = 991 + N
    X = pkg.GraphSK(N, seed=seed) if binary else pkg.GraphSKNormal(N, seed=seed)
    run = O.standard_mc_skb if binary else O.standard_mc_skn
    with pkg.Engine(X, R) as eng:
        eng.seed(seed); eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(1.0, iters, step)
        C1 = eng.get_config(); lf1 = eng.fields()
    for r in (0, R - 1):
        ref = run(X.J, 1.0, iters, step, seed, C0.s[r], replica=r)
        n = iters // step
        ok = (Es[r][:n] == ref[0][:n]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and (lf1[r] == ref[3]).all()
        print("binary" if binary else "gauss", N, R, iters, step, "replica", r, "ok" if ok else "MISMATCH", flush=True)
        bad += 0 if ok else 1
sys.exit(1 if bad else 0)
