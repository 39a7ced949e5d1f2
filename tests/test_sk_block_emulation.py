"""The blocked dense-SK schedule (tests/emulate_sk_block.py, the algorithm of sk_block_kernel) reproduces the sequential oracle chain
bit for bit — window lanes with repeated sites, consecutive same-site acceptances (the array swap of src/graphs/SK.jl:247-250),
ragged last blocks and sample points inside a block."""
import numpy as np
import pytest

import emulate_sk_block as EB


@pytest.mark.parametrize("N,beta,iters,step,seed", [
    (3, 0.3, 700, 7, 11),          # tiny: nearly every block has same-site repeats and swaps
    (10, 2.0, 1000, 100, 5),       # test/runtests.jl:67 GraphSKNormal(10)
    (10, 0.0, 300, 1, 6),          # beta = 0: every attempt accepted, step = 1
    (24, 1.0, 1500, 37, 7),
    (70, 0.7, 900, 64, 8),
    (130, 1.0, 333, 50, 9),
])
@pytest.mark.parametrize("hform", [False, True], ids=["lfields", "hform"])
def test_blocked_schedule_equals_sequential_chain(oracle, N, beta, iters, step, seed, hform):
    J = oracle.gen_sk_gauss(N, seed)
    for replica in (0, 3):
        c0 = oracle.init_config(seed, replica, N)
        ref = oracle.standard_mc_skn(J, beta, iters, step, seed, c0, replica=replica)
        got = EB.run_chain(oracle, J, beta, iters, step, seed, c0, replica=replica, hform=hform)
        assert got[0].shape == ref[0].shape and (got[0] == ref[0]).all()
        assert (got[1] == ref[1]).all() and got[2] == ref[2]
        assert (got[3] == ref[3]).all()


@pytest.mark.parametrize("N,beta,seed", [(3, 0.0, 21), (3, 0.4, 22), (5, 0.1, 23), (40, 1.0, 24)])
def test_hform_bulk_in_resumed_pieces(oracle, N, beta, seed):
    """The H-form bulk phase cut into calls of odd lengths (the state handed over is lfields / lfields_last / move_last / E, as between the
    kernel's segments): lfields_last and move_last must be exact at every cut, or a same-site acceptance across it undoes the wrong array."""
    J = oracle.gen_sk_gauss(N, seed)
    c0 = oracle.init_config(seed, 1, N)
    pieces = [65, 1, 127, 64, 200, 3]
    ref = oracle.standard_mc_skn(J, beta, sum(pieces), 1, seed, c0, replica=1)
    ch, state, it0, Es, acc = c0, None, 0, [], 0
    for n in pieces:
        got = EB.run_chain(oracle, J, beta, n, 1, seed, ch, it0=it0, replica=1, state=state, hform=True)
        Es.extend(got[0]); ch = got[1]; acc += got[2]; state = got[4]; it0 += n
    assert (np.array(Es) == ref[0]).all() and (ch == ref[1]).all() and acc == ref[2] and (state[0] == ref[3]).all()


def test_swaps_do_occur(oracle):
    """the case above with N = 3 must exercise the swap branch, otherwise it proves nothing about it"""
    N, seed = 3, 11
    J = oracle.gen_sk_gauss(N, seed)
    c0 = oracle.init_config(seed, 0, N)
    sites = [oracle.site_of(seed, t, N) for t in range(1, 701)]
    assert any(a == b for a, b in zip(sites, sites[1:]))
    got = EB.run_chain(oracle, J, 0.3, 700, 7, seed, c0)
    assert got[2] > 300

