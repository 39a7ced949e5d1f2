"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must reproduce them on the
CPU, the HIP path must reproduce them on the GPU."""
import glob
import os

import numpy as np
import pytest

FILES = sorted(f for f in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz"))
               if os.path.basename(f) != "models.npz")        # models.npz: tests/test_golden_models.py


def test_fixtures_present():
    assert len(FILES) >= 5


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    A, J = np.ascontiguousarray(g["A"]), np.ascontiguousarray(g["J"].astype(np.int32))
    seed = int(g["seed"])
    R, N = g["chunks0"].shape[0], A.shape[0]
    assert (oracle.init_configs(seed, 0, R, N) == g["chunks0"]).all()
    Es, ch, acc = oracle.standard_mc_sparse_batch(A, J, float(g["beta"]), int(g["iters"]), int(g["step"]), seed, g["chunks0"],
                                                  form=str(g["kind"]))
    assert (Es == g["Es"]).all() and (ch == g["chunks1"]).all() and (acc == g["accepted"]).all()
    # the graph itself is part of the fixture: regenerate it from the streams
    if str(g["kind"]) == "rrg":
        assert (oracle.gen_rrg(N, A.shape[1], seed) == A).all()
    assert (oracle.gen_couplings(A, seed) == J).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_reproduces_golden(pkg, path):
    g = np.load(path)
    X = pkg.GraphRRG.from_AJ(g["A"], g["J"])
    R = g["chunks0"].shape[0]
    with pkg.Engine(X, R) as eng:
        eng.seed(int(g["seed"]))
        eng.init_spins_random()
        assert (eng.get_config().s == g["chunks0"]).all()
        Es, acc = eng.standard_mc(float(g["beta"]), int(g["iters"]), int(g["step"]))
        C1 = eng.get_config()
    assert (Es == g["Es"]).all() and (C1.s == g["chunks1"]).all() and (acc == g["accepted"]).all()
