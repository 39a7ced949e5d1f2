#!/usr/bin/env python3
"""Golden vectors for the model families beyond the +-J sparse path (tests/golden/models.npz), made with the CPU oracle.

One small chain family per graph type of test/runtests.jl:36-81 that the library covers, at the test's parameters (beta = 2.0,
step = 100, :132-136): the energies at every sample and the final configuration of each of R = 4 replicas.  The file pins both
sides: tests/test_golden_models.py checks that the oracle still reproduces it on the CPU and that the HIP path reproduces it on
the GPU — a change that moved oracle and kernels together (a stream tag, the deterministic exp, a level conversion) shows up here.
Run from the repo root:  python tests/golden/make_golden_models.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as O  # noqa: E402

SEED, R, BETA, ITERS, STEP = 8426732438942, 4, 2.0, 4000, 100


def chains(fn, C0):
    """fn(chunks, replica) -> (Es, chunks_out, ...); stacks Es and final chunks over the replicas"""
    outs = [fn(C0[r], r) for r in range(R)]
    return np.stack([np.asarray(o[0]) for o in outs]), np.stack([o[1] for o in outs])


def compute():
    out = {"seed": np.uint64(SEED), "R": R, "beta": BETA, "iters": ITERS, "step": STEP}
    A = O.gen_rrg(10, 3, SEED)
    J = O.gen_couplings(A, SEED)
    C10 = O.init_configs(SEED, 0, R, 10)
    out["rrg_A"], out["rrg_J"], out["C10"] = A, J.astype(np.int8), C10
    # GraphRRG(10, 3): rrrMC, bklMC, wtmMC, extremal_opt (standardMC is in rrg_n10_k3.npz)
    out["rrg_rrr_Es"], out["rrg_rrr_C"] = chains(lambda c, r: O.rrr_sparse(A, J, BETA, ITERS, STEP, SEED, c, replica=r), C10)
    out["rrg_bkl_Es"], out["rrg_bkl_C"] = chains(lambda c, r: O.rrr_sparse(A, J, BETA, ITERS, STEP, SEED, c, replica=r, bkl=True), C10)
    out["rrg_wtm_Es"], out["rrg_wtm_C"] = chains(lambda c, r: O.wtm_mc_sparse(A, J, BETA, ITERS // STEP, float(STEP), SEED, c, replica=r), C10)
    out["rrg_eo_Es"], out["rrg_eo_C"] = chains(lambda c, r: O.extremal_opt_sparse(A, J, 1.3, ITERS, STEP, SEED, c, replica=r), C10)
    # GraphRRG(10, 3, (-1.0, 0.0, 1.0)): DFloat64 levels, energies in level units (t / gcd)
    lev, mul, div = O.dfloat_units((-1.0, 0.0, 1.0))
    Jl = O.gen_couplings(A, SEED, lev)
    out["lev_J"] = Jl.astype(np.int8)
    out["lev_std_Es"], out["lev_std_C"] = chains(lambda c, r: O.standard_mc_lev(A, Jl, BETA, ITERS, STEP, SEED, c, replica=r, mul=mul, div=div), C10)
    out["lev_rrr_Es"], out["lev_rrr_C"] = chains(lambda c, r: O.rrr_sparse(A, Jl, BETA, ITERS, STEP, SEED, c, replica=r, lev=lev, mul=mul, div=div), C10)
    # GraphRRGNormal(10, 3) and GraphRRGNormalDiscretized(10, 3, (-1, 0, 1))
    cJ = O.gen_couplings_gauss(A, SEED)
    out["cJ"] = cJ
    out["spf_std_Es"], out["spf_std_C"] = chains(lambda c, r: O.standard_mc_spf(A, cJ, BETA, ITERS, STEP, SEED, c, replica=r), C10)
    out["spf_rrr_Es"], out["spf_rrr_C"] = chains(lambda c, r: O.cont_sparse("rrr", A, cJ, BETA, ITERS, STEP, SEED, c, replica=r), C10)
    dJ, rJ = O.discretize(cJ, (-1, 0, 1))
    out["dbl_rrr_Es"], out["dbl_rrr_C"] = chains(lambda c, r: O.rrr_double_sparse(A, dJ, rJ, (-1, 0, 1), BETA, ITERS, STEP, SEED, c, replica=r), C10)
    out["dbl_std_Es"], out["dbl_std_C"] = chains(lambda c, r: O.standard_mc_dbl(A, dJ, rJ, BETA, ITERS, STEP, SEED, c, replica=r), C10)
    # GraphSKNormal(10), GraphSK(10)
    Jn = O.gen_sk_gauss(10, SEED)
    out["skn_J"] = Jn
    out["skn_std_Es"], out["skn_std_C"] = chains(lambda c, r: O.standard_mc_skn(Jn, BETA, ITERS, STEP, SEED, c, replica=r), C10)
    out["skn_rrr_Es"], out["skn_rrr_C"] = chains(lambda c, r: O.rrr_mc_skn(Jn, BETA, ITERS, STEP, SEED, c, replica=r), C10)
    Jb = O.gen_sk_binary(10, SEED)
    out["skb_J"] = Jb
    out["skb_std_Es"], out["skb_std_C"] = chains(lambda c, r: O.standard_mc_skb(Jb, BETA, ITERS, STEP, SEED, c, replica=r), C10)
    # GraphQuant(10, 8, 0.5, 2.0, GraphRRG, 10, 3) (runtests.jl:78): rrrMC and standardMC
    M, Gamma = 8, 0.5
    fourK = O.quant_fourK(BETA, Gamma, M)
    C80 = O.init_configs(SEED, 0, R, 10 * M)
    out["C80"], out["fourK"] = C80, fourK
    out["quant_rrr_Es"], out["quant_rrr_C"] = chains(lambda c, r: O.rrr_mc_quant(A, J, M, fourK, BETA, ITERS, STEP, SEED, c, replica=r), C80)
    out["quant_std_Es"], out["quant_std_C"] = chains(lambda c, r: O.standard_mc_quant(A, J, M, fourK, BETA, ITERS, STEP, SEED, c, replica=r), C80)
    return out


def main():
    out = compute()
    np.savez_compressed(os.path.join(HERE, "models.npz"), **out)
    for k in sorted(out):
        if k.endswith("_Es"):
            print("%-14s E_final = %s" % (k, np.asarray(out[k])[:, -1]))


if __name__ == "__main__":
    main()
