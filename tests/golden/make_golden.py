#!/usr/bin/env python3
"""Generate the golden vectors of tests/golden/*.npz with the CPU oracle (oracle/rrrmc_oracle.c).

The reference's own tests hold no numeric golden vectors (test/runtests.jl only asserts the tracked-energy
invariant) and the reference cannot run here (Julia is not installed), so the fixtures are produced by the
oracle, whose trust rests on tests/test_oracle_*.py.  Each file stores the inputs (A, J, seed, beta, iters,
step, initial chunks) and the expected outputs (Es, final chunks, accepted) of standardMC.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as O  # noqa: E402

CASES = {
    # name: (graph kind, graph args, R, beta, iters, step, seed)
    "rrg_n10_k3": ("rrg", (10, 3), 32, 2.0, 10000, 100, 8426732438942),       # test/runtests.jl:36,125-130
    "ea_l2_d3": ("ea", (2, 3), 32, 2.0, 10000, 100, 8426732438942),            # runtests.jl:46 (doubled neighbours)
    "ea_l3_d2": ("ea", (3, 2), 32, 2.0, 10000, 100, 8426732438942),            # runtests.jl:56
    "rrg_n128_k3": ("rrg", (128, 3), 8, 1.0, 10000, 100, 0x5EED),              # BASELINE.json configs[0] shape
    "rrg_n4096_k3": ("rrg", (4096, 3), 32, 1.0, 16384, 4096, 0x5EED),          # BASELINE.json configs[1] graph
}


def main():
    for name, (kind, gargs, R, beta, iters, step, seed) in CASES.items():
        A = O.gen_rrg(*gargs, seed) if kind == "rrg" else O.gen_ea(*gargs)
        J = O.gen_couplings(A, seed)
        N = A.shape[0]
        ch0 = O.init_configs(seed, 0, R, N)
        Es, ch1, acc = O.standard_mc_sparse_batch(A, J, beta, iters, step, seed, ch0, form=kind)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind=kind, A=A, J=J.astype(np.int8), seed=np.uint64(seed),
                            beta=beta, iters=iters, step=step, chunks0=ch0, Es=Es, chunks1=ch1, accepted=acc)
        print(name, "E_final[0..3] =", Es[:4, -1], "acc mean", acc.mean())


if __name__ == "__main__":
    main()
