"""debug helper (GPU box): where does the blocked SK kernel first deviate from the oracle?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as e
pkg = e.load_package(); O = e.load_oracle()
def run(N, R, iters, step, beta=1.0, seed=77):
    X = pkg.GraphSKNormal(N, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed); eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config(); lf = eng.fields()
    bad = []
    for r in range(R):
        ref = O.standard_mc_skn(X.J, beta, iters, step, seed, C0.s[r], replica=r)
        okE = (Es[r] == ref[0]).all(); okC = (C1.s[r] == ref[1]).all(); okA = acc[r] == ref[2]; okF = (lf[r] == ref[3]).all()
        if not (okE and okC and okA and okF):
            fe = int(np.argmax(Es[r] != ref[0])) if not okE else -1
            nbf = int((lf[r] != ref[3]).sum())
            wf = np.nonzero(lf[r] != ref[3])[0][:6].tolist()
            bad.append((r, fe, okC, okA, nbf, wf))
    print("N=%d R=%d iters=%d step=%d threads=%s: %d bad replicas %s" % (N, R, iters, step, os.environ.get("RRRMC_SK_THREADS", "-"), len(bad), bad[:4]))
for N in (512, 513, 640, 1024):
    for iters in (64, 256, 2048):
        run(N, 8, iters, 16)
