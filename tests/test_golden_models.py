"""Golden vectors of the model families beyond the +-J sparse path (tests/golden/models.npz, made by
tests/golden/make_golden_models.py): the oracle must reproduce them on the CPU, the HIP path on the GPU."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "models.npz")


def _load():
    return {k: v for k, v in np.load(PATH).items()}


def test_oracle_reproduces_model_goldens(oracle):
    spec = importlib.util.spec_from_file_location("make_golden_models", os.path.join(HERE, "golden", "make_golden_models.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    new, old = mod.compute(), _load()
    assert set(new) == set(old)
    for k in old:
        assert np.array_equal(np.asarray(new[k]), old[k]), k


def _run(pkg, X, C0, fn):
    with pkg.Engine(X, C0.shape[0]) as eng:
        eng.seed(int(_G["seed"]))
        eng.set_config(pkg.Config(X.N, C0.shape[0], C0.copy()))
        Es = fn(eng)[0]
        return np.asarray(Es), eng.get_config().s


_G = {}


@pytest.mark.gpu
def test_hip_reproduces_model_goldens(pkg):
    g = _load()
    _G.update(g)
    beta, iters, step = float(g["beta"]), int(g["iters"]), int(g["step"])
    A, C10, C80 = g["rrg_A"], g["C10"], g["C80"]

    def check(name, X, C0, fn, conv=lambda x: x):
        Es, C = _run(pkg, X, C0, fn)
        assert np.array_equal(conv(Es), g[name + "_Es"]), name
        assert np.array_equal(C, g[name + "_C"]), name

    X = pkg.GraphRRG.from_AJ(A, g["rrg_J"])
    check("rrg_rrr", X, C10, lambda e: e.rrr_mc(beta, iters, step))
    check("rrg_bkl", X, C10, lambda e: e.bkl_mc(beta, iters, step))
    check("rrg_wtm", X, C10, lambda e: e.wtm_mc(beta, iters // step, float(step)))
    check("rrg_eo", X, C10, lambda e: e.extremal_opt(1.3, iters, step))
    XL = pkg.GraphRRG.from_AJ(A, g["lev_J"], (-1.0, 0.0, 1.0))
    units = lambda Es: np.rint(Es * XL.lev_div / XL.lev_mul).astype(np.int64)       # the fixture stores level units
    check("lev_std", XL, C10, lambda e: e.standard_mc(beta, iters, step), units)
    check("lev_rrr", XL, C10, lambda e: e.rrr_mc(beta, iters, step), units)
    XF = pkg.GraphRRGNormal.from_AJ(A, g["cJ"])
    check("spf_std", XF, C10, lambda e: e.standard_mc(beta, iters, step))
    check("spf_rrr", XF, C10, lambda e: e.rrr_mc(beta, iters, step))
    XD = pkg.GraphRRGNormalDiscretized(10, 3, (-1, 0, 1), seed=int(g["seed"]))
    assert np.array_equal(XD.A, A) and np.array_equal(XD.cJ, g["cJ"])
    check("dbl_rrr", XD, C10, lambda e: e.rrr_mc(beta, iters, step))
    check("dbl_std", XD, C10, lambda e: e.standard_mc(beta, iters, step))
    XN = pkg.GraphSKNormal.from_J(g["skn_J"])
    check("skn_std", XN, C10, lambda e: e.standard_mc(beta, iters, step))
    check("skn_rrr", XN, C10, lambda e: e.rrr_mc(beta, iters, step))
    XB = pkg.GraphSK(10, seed=int(g["seed"]))
    assert np.array_equal(np.asarray(XB.J).reshape(-1), np.asarray(g["skb_J"]).reshape(-1))
    check("skb_std", XB, C10, lambda e: e.standard_mc(beta, iters, step))
    XQ = pkg.GraphQuant(X, 8, 0.5, beta)
    assert XQ.fourK == float(g["fourK"])
    check("quant_rrr", XQ, C80, lambda e: e.rrr_mc(beta, iters, step))
    check("quant_std", XQ, C80, lambda e: e.standard_mc(beta, iters, step))
