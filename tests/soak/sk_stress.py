"""One-off stress of the dense-SK standardMC kernels against the oracle (GPU box): seeded random (N, R, beta, iters, step) incl. tiny N
(duplicate sites inside a 64-attempt block, consecutive moves at one site: the array-swap undo), beta = 0 (every move accepted) and large beta,
sample points on and off block boundaries, resumed calls.  python tests/soak/sk_stress.py [cases] [seed] [binary]  (binary: GraphSK, the integer-field model)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
O.build()
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
binary = len(sys.argv) > 3 and sys.argv[3] == "binary"
bad = 0
t0 = time.time()
for c in range(ncase):
    N = int(rng.choice([2, 3, 5, 17, 63, 64, 65, 127, 128, 255, 256, 257, 300, 511, 512, 513, 700, 1000, 1023, 1024, 1025, 1500, 2048]))
    R = int(rng.integers(1, 34))
    beta = float(rng.choice([0.0, 0.1, 0.5, 1.0, 2.0, 5.0, 50.0]))
    iters = int(rng.choice([1, 63, 64, 65, 127, 128, 1000, 2500, 4096, 5000]))
    step = int(rng.choice([1, 7, 64, 100, 128, 1000]))
    seed = int(rng.integers(1, 1 << 40))
    X = pkg.GraphSK(N, seed=seed) if binary else pkg.GraphSKNormal(N, seed=seed)
    run = O.standard_mc_skb if binary else O.standard_mc_skn
    with pkg.Engine(X, R) as eng:
        eng.seed(seed); eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config(); lf1 = eng.fields()
        Es2, acc2 = eng.standard_mc(beta, iters // 3 + 1, step)
        C2 = eng.get_config()
    ok = True
    ch_ref = {}
    for r in sorted({0, R // 2, R - 1}):
        ref = run(X.J, beta, iters, step, seed, C0.s[r], replica=r)
        ch_ref[r] = ref[1]
        n1, n2 = iters // step, (iters // 3 + 1) // step
        ok = ok and (Es[r][:n1] == ref[0][:n1]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and (lf1[r] == ref[3]).all()
        ref2 = run(X.J, beta, iters // 3 + 1, step, seed, ref[1], it0=iters, replica=r)
        ok = ok and (Es2[r][:n2] == ref2[0][:n2]).all() and (C2.s[r] == ref2[1]).all() and acc2[r] == ref2[2]
    if not ok:
        bad += 1
        print("MISMATCH", dict(N=N, R=R, beta=beta, iters=iters, step=step, seed=seed), flush=True)
    if c % 25 == 24: print("case", c + 1, "bad", bad, "%.0fs" % (time.time() - t0), flush=True)
print("done: %d cases, %d mismatches" % (ncase, bad))
sys.exit(1 if bad else 0)
