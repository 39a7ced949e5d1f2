#!/usr/bin/env python3
"""Write the RNG-free tapes tests/golden/tape_*.txt: a graph, a start configuration, every random draw of a run (pre-drawn from the
build's Philox streams) and the results the C oracle obtained.  Replayed by tests/tape_replay.py (plain Python after the Julia
sources), by tests/replay_tape.jl (the reference's OWN functions, wherever Julia + RRRMC.jl exist: `julia tests/replay_tape.jl
tests/golden/tape_rrg_n128.txt`) and by the HIP library (tests/test_tapes.py).

A tape is only written if (a) the plain-Python replay — libm exp, Float64 uniforms — reproduces the oracle's run exactly and
(b) no accept / class-pick decision of the run is closer than 1e-9 (relative) to its threshold, so that the last-bit differences
between Julia's exp, libm's and the build's det_exp cannot flip a decision of the replay.

  python tests/golden/make_tapes.py      (from the repo root; needs gcc for the oracle)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O          # noqa: E402
import tape_replay as TR    # noqa: E402


def fmt_array(name, vals):
    vals = list(vals)
    lines = ["@%s array %d" % (name, len(vals))]
    for i in range(0, len(vals), 16):
        lines.append(" ".join(vals[i:i + 16]))
    return "\n".join(lines)


def u53(u64):
    return float(int(u64) >> 11) * 2.0 ** -53


def write_standard(path, seed, N=128, K=3, beta=1.0, iters=8000, step=250, replica=0):
    A = O.gen_rrg(N, K, seed)
    J = O.gen_couplings(A, seed)
    C0 = O.init_config(seed, replica, N)
    Es, ch, acc, _lf, sites, flips = O.standard_mc_sparse(A, J, beta, iters, step, seed, C0, replica=replica, trace=True)
    us = [u53(O.accept_uniform(seed, g, replica)) for g in range(1, iters + 1)]
    body = ["# RRRMC tape v1 — standardMC(X::GraphRRG{Int,(-1,1),%d}, beta, iters; step, C0) with every random draw pre-drawn" % K,
            "# (sites 1-based; uniforms are consulted only when delta_energy > 0, src/RRRMC.jl:39).  Written by tests/golden/make_tapes.py",
            "@kind standardMC", "@N %d" % N, "@K %d" % K, "@beta %r" % beta, "@iters %d" % iters, "@step %d" % step,
            "@seed %d" % seed, "@replica %d" % replica,
            fmt_array("A", ("%d" % (v + 1) for v in A.reshape(-1))), fmt_array("J", ("%d" % v for v in J.reshape(-1))),
            fmt_array("C0", ("%016x" % int(c) for c in C0)),
            fmt_array("sites", ("%d" % (int(v) + 1) for v in sites)), fmt_array("uniforms", (repr(u) for u in us)),
            fmt_array("expected_Es", ("%d" % int(e) for e in Es)), fmt_array("expected_chunks", ("%016x" % int(c) for c in ch)),
            "@expected_accepted %d" % acc, fmt_array("expected_flips", ("%d" % int(f) for f in flips))]
    tmp = path + ".tmp"
    open(tmp, "w").write("\n".join(body) + "\n")
    got = TR.replay_standard_mc(TR.read_tape(tmp))
    assert got["Es"] == [int(e) for e in Es] and got["chunks"] == [int(c) for c in ch] and got["accepted"] == acc
    assert got["flips"] == [int(f) for f in flips]
    if got["min_margin"] < 1e-9:
        os.remove(tmp)
        return False
    os.replace(tmp, path)
    print("%s: %d iterations, accepted %d, closest decision margin %.2e" % (os.path.basename(path), iters, acc, got["min_margin"]))
    return True


def write_quant(path, seed, Nk=16, K=3, M=4, beta=2.0, Gamma=0.5, iters=3000, step=100, staged_thr=0.5, staged_thr_fact=5.0, replica=0):
    A = O.gen_rrg(Nk, K, seed)
    J = O.gen_couplings(A, seed)
    N = Nk * M
    fourK = O.quant_fourK(beta, Gamma, M)
    C0 = O.init_config(seed, replica, N)
    Es, ch, acc, staged, pos, sizes = O.rrr_mc_quant(A, J, M, fourK, beta, iters, step, seed, C0, replica=replica, staged_thr=staged_thr,
                                                     staged_thr_fact=staged_thr_fact, want_cache=True)
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32)
    ucls, umem, uacc = [], [], []
    for g in range(1, iters + 1):          # RRR stream: sub 0 = (class uniform, member word), sub 1 = acceptance uniform
        w0 = O.philox([g & 0xFFFFFFFF, g >> 32, replica, 8], key)
        w1 = O.philox([g & 0xFFFFFFFF, g >> 32, replica, 8 | (1 << 8)], key)
        ucls.append(u53((int(w0[0]) << 32) | int(w0[1])))
        umem.append((int(w0[2]) << 32) | int(w0[3]))
        uacc.append(u53((int(w1[0]) << 32) | int(w1[1])))
    body = ["# RRRMC tape v1 — rrrMC(X::GraphQuant over GraphRRG{Int,(-1,1),%d} slices, beta, iters; step, C0, staged_thr, staged_thr_fact)" % K,
            "# with every random draw pre-drawn: u_class = rand() of rand_move (src/DeltaE.jl:148), u_member -> rand(1:t) as",
            "# floor(u * t / 2^64) + 1 (src/ArraySets.jl:83), u_accept = rand() of accept(c, x) (src/RRRMC.jl:43, only when consulted).",
            "# All M slices share (A, J).  Written by tests/golden/make_tapes.py",
            "@kind rrrMC_quant", "@Nk %d" % Nk, "@K %d" % K, "@M %d" % M, "@beta %r" % beta, "@Gamma %r" % Gamma, "@fourK %r" % fourK,
            "@iters %d" % iters, "@step %d" % step, "@staged_thr %r" % staged_thr, "@staged_thr_fact %r" % staged_thr_fact,
            "@seed %d" % seed, "@replica %d" % replica,
            fmt_array("A", ("%d" % (v + 1) for v in A.reshape(-1))), fmt_array("J", ("%d" % v for v in J.reshape(-1))),
            fmt_array("C0", ("%016x" % int(c) for c in C0)),
            fmt_array("u_class", (repr(u) for u in ucls)), fmt_array("u_member", ("%d" % u for u in umem)),
            fmt_array("u_accept", (repr(u) for u in uacc)),
            fmt_array("expected_Es", (repr(float(e)) for e in Es)), fmt_array("expected_chunks", ("%016x" % int(c) for c in ch)),
            "@expected_accepted %d" % acc, "@expected_staged_its %d" % staged,
            fmt_array("expected_sizes", ("%d" % int(v) for v in sizes)), fmt_array("expected_pos", ("%d" % (int(v) + 1) for v in pos))]
    tmp = path + ".tmp"
    open(tmp, "w").write("\n".join(body) + "\n")
    got = TR.replay_rrr_quant(TR.read_tape(tmp))
    ok = (got["chunks"] == [int(c) for c in ch] and got["accepted"] == acc and got["staged_its"] == staged
          and got["sizes"] == [int(v) for v in sizes] and got["pos"] == [int(v) + 1 for v in pos]
          and np.allclose(got["Es"], Es, rtol=1e-12, atol=1e-12) and got["min_margin"] >= 1e-9)
    if not ok:
        os.remove(tmp)
        return False
    os.replace(tmp, path)
    print("%s: %d iterations, accepted %d, staged %d, closest decision margin %.2e" % (os.path.basename(path), iters, acc, staged, got["min_margin"]))
    return True


def gen_J_uniform(A, seed):
    """EA.gen_J(Float64, N, A) do 4 * rand() - 2 end (src/QAliases.jl:60-62 over src/graphs/EA.jl:45-71): one draw per bond x < y in (x, k) order,
    mirrored into the first free slot of row y.  The draws are numpy's Philox stream keyed by the seed (the tape stores the couplings)."""
    N, K = A.shape
    rng = np.random.Generator(np.random.Philox(key=int(seed)))
    J = np.full((N, K), np.nan)
    for x in range(N):
        for k in range(K):
            y = int(A[x, k])
            if x < y:
                J[x, k] = 4.0 * rng.random() - 2.0
                J[y, np.nonzero(np.isnan(J[y]))[0][0]] = J[x, k]
    return J


def write_quant_f64(path, seed, L=4, D=2, M=8, beta=2.0, Gamma=0.5, iters=3000, step=100, staged_thr=0.5, staged_thr_fact=5.0, replica=0):
    """rrrMC(X::DoubleGraph) on GraphQEAT(L, D, M, Γ, β) = GraphQuant{fourK,GraphEANormal{2D}} (src/QAliases.jl:50-83): M slices of ONE lattice
    with Float64 couplings, every slice with its own LocalFields{Float64} and the exact undo path of update_cache! (EA.jl:613-653) — the
    rejected direct moves of the run go through it.  The tape must hold staged and direct iterations, and undo swaps."""
    A = O.gen_ea(L, D)
    Nk, K = A.shape
    J = gen_J_uniform(A, seed)
    N = Nk * M
    fourK = O.quant_fourK(beta, Gamma, M)
    C0 = O.init_config(seed, replica, N)
    Es, ch, acc, staged, pos, sizes = O.rrr_mc_quant_spf(A, J, M, fourK, beta, iters, step, seed, C0, replica=replica, staged_thr=staged_thr,
                                                         staged_thr_fact=staged_thr_fact, want_cache=True)
    if not (0 < staged < iters and acc < iters):          # both branches, and rejected (undone) moves
        return False
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32)
    ucls, umem, uacc = [], [], []
    for g in range(1, iters + 1):          # RRR stream: sub 0 = (class uniform, member word), sub 1 = acceptance uniform
        w0 = O.philox([g & 0xFFFFFFFF, g >> 32, replica, 8], key)
        w1 = O.philox([g & 0xFFFFFFFF, g >> 32, replica, 8 | (1 << 8)], key)
        ucls.append(u53((int(w0[0]) << 32) | int(w0[1])))
        umem.append((int(w0[2]) << 32) | int(w0[3]))
        uacc.append(u53((int(w1[0]) << 32) | int(w1[1])))
    body = ["# RRRMC tape v1 — rrrMC(X::GraphQEAT = GraphQuant{fourK,GraphEANormal{%d}}, beta, iters; step, C0, staged_thr, staged_thr_fact) on the" % K,
            "# L = %d, D = %d lattice (src/QAliases.jl:50-83): all M slices share (A, J::Float64); draws as in the other rrrMC_quant tapes." % (L, D),
            "# Written by tests/golden/make_tapes.py",
            "@kind rrrMC_quant", "@slices f64", "@L %d" % L, "@D %d" % D, "@Nk %d" % Nk, "@K %d" % K, "@M %d" % M, "@beta %r" % beta, "@Gamma %r" % Gamma,
            "@fourK %r" % fourK, "@iters %d" % iters, "@step %d" % step, "@staged_thr %r" % staged_thr, "@staged_thr_fact %r" % staged_thr_fact,
            "@seed %d" % seed, "@replica %d" % replica,
            fmt_array("A", ("%d" % (v + 1) for v in A.reshape(-1))), fmt_array("J", (repr(float(v)) for v in J.reshape(-1))),
            fmt_array("C0", ("%016x" % int(c) for c in C0)),
            fmt_array("u_class", (repr(u) for u in ucls)), fmt_array("u_member", ("%d" % u for u in umem)),
            fmt_array("u_accept", (repr(u) for u in uacc)),
            fmt_array("expected_Es", (repr(float(e)) for e in Es)), fmt_array("expected_chunks", ("%016x" % int(c) for c in ch)),
            "@expected_accepted %d" % acc, "@expected_staged_its %d" % staged,
            fmt_array("expected_sizes", ("%d" % int(v) for v in sizes)), fmt_array("expected_pos", ("%d" % (int(v) + 1) for v in pos))]

    def check(t):
        got = TR.replay_rrr_quant(t)
        ok = (got["chunks"] == [int(c) for c in ch] and got["accepted"] == acc and got["staged_its"] == staged
              and got["sizes"] == [int(v) for v in sizes] and got["pos"] == [int(v) + 1 for v in pos]
              and got["Es"] == [float(e) for e in Es] and got["min_margin"] >= 1e-9)
        return ok, "%d iterations, accepted %d, staged %d, closest decision margin %.2e" % (iters, acc, staged, got["min_margin"])
    return _finish(path, body, check)


def _finish(path, body, check):
    tmp = path + ".tmp"
    open(tmp, "w").write("\n".join(body) + "\n")
    ok, msg = check(TR.read_tape(tmp))
    if not ok:
        os.remove(tmp)
        return False
    os.replace(tmp, path)
    print("%s: %s" % (os.path.basename(path), msg))
    return True


def write_standard_ea(path, seed, L=2, D=3, beta=1.0, iters=3000, step=100, replica=0):
    """standardMC on GraphEA(2, 3): every site lists each neighbour twice (doubled bonds, EA.jl:250-256) and on 8 sites the same spin is
    accepted twice in a row often, so the undo fast path (EA.jl:231-240) fires too."""
    A = O.gen_ea(L, D)
    N, K = A.shape
    J = O.gen_couplings(A, seed)
    C0 = O.init_config(seed, replica, N)
    Es, ch, acc, _lf, sites, flips = O.standard_mc_sparse(A, J, beta, iters, step, seed, C0, replica=replica, trace=True, form="ea")
    us = [u53(O.accept_uniform(seed, g, replica)) for g in range(1, iters + 1)]
    body = ["# RRRMC tape v1 — standardMC(X::GraphEA{Int,(-1,1),%d}, beta, iters; step, C0) on the L = %d, D = %d lattice: every neighbour is" % (K, L, D),
            "# listed twice (two bonds to the same site).  Sites 1-based; uniforms consulted only when delta_energy > 0 (src/RRRMC.jl:39).",
            "@kind standardMC", "@form ea", "@L %d" % L, "@D %d" % D, "@N %d" % N, "@K %d" % K, "@beta %r" % beta, "@iters %d" % iters,
            "@step %d" % step, "@seed %d" % seed, "@replica %d" % replica,
            fmt_array("A", ("%d" % (v + 1) for v in A.reshape(-1))), fmt_array("J", ("%d" % v for v in J.reshape(-1))),
            fmt_array("C0", ("%016x" % int(c) for c in C0)),
            fmt_array("sites", ("%d" % (int(v) + 1) for v in sites)), fmt_array("uniforms", (repr(u) for u in us)),
            fmt_array("expected_Es", ("%d" % int(e) for e in Es)), fmt_array("expected_chunks", ("%016x" % int(c) for c in ch)),
            "@expected_accepted %d" % acc, fmt_array("expected_flips", ("%d" % int(f) for f in flips))]

    def check(t):
        got = TR.replay_standard_mc_ea(t)
        ok = (got["Es"] == [int(e) for e in Es] and got["chunks"] == [int(c) for c in ch] and got["accepted"] == acc
              and got["flips"] == [int(f) for f in flips] and got["min_margin"] >= 1e-9 and got["undos"] >= 5)
        return ok, "%d iterations, accepted %d, %d undo swaps, closest decision margin %.2e" % (iters, acc, got["undos"], got["min_margin"])
    return _finish(path, body, check)


def write_standard_sk(path, seed, N, binary, beta, iters, step, replica=0, want_swaps=1):
    """standardMC on GraphSKNormal(N) (Float64 cache, SK.jl:212-276) or the binary GraphSK(N) (integer cache, SK.jl:62-135); the tape must
    contain a consecutive accepted pair at one site, so that the whole-array swap (SK.jl:247-250 / :106-109) fires."""
    C0 = O.init_config(seed, replica, N)
    if binary:
        Jb = O.gen_sk_binary(N, seed)
        Es, ch, acc, lf = O.standard_mc_skb(Jb, beta, iters, step, seed, C0, replica=replica)
        Jsec = fmt_array("J_chunks", ("%016x" % int(c) for c in Jb.reshape(-1)))
    else:
        Jm = O.gen_sk_gauss(N, seed)
        Es, ch, acc, lf = O.standard_mc_skn(Jm, beta, iters, step, seed, C0, replica=replica)
        Jsec = fmt_array("J", (repr(float(v)) for v in Jm.reshape(-1)))
    sites = [O.site_of(seed, g, N) for g in range(1, iters + 1)]
    us = [O.rand53(seed, g, replica) for g in range(1, iters + 1)]
    kind = "standardMC_skb" if binary else "standardMC_skn"
    body = ["# RRRMC tape v1 — standardMC(X::%s, beta, iters; step, C0) with every random draw pre-drawn" % ("GraphSK (binary couplings, J rows as BitVector chunks)" if binary else "GraphSKNormal (J row-major)"),
            "# (sites 1-based; uniforms consulted only when delta_energy > 0, src/RRRMC.jl:39).  Written by tests/golden/make_tapes.py",
            "@kind %s" % kind, "@N %d" % N, "@beta %r" % beta, "@iters %d" % iters, "@step %d" % step, "@seed %d" % seed, "@replica %d" % replica,
            Jsec, fmt_array("C0", ("%016x" % int(c) for c in C0)),
            fmt_array("sites", ("%d" % (v + 1) for v in sites)), fmt_array("uniforms", (repr(u) for u in us)),
            fmt_array("expected_Es", (repr(float(e)) for e in Es)), fmt_array("expected_chunks", ("%016x" % int(c) for c in ch)),
            "@expected_accepted %d" % acc,
            fmt_array("expected_lfields", (("%d" % int(v)) if binary else repr(float(v)) for v in lf))]

    def check(t):
        got = TR.replay_standard_mc_sk(t)
        ok = (got["chunks"] == [int(c) for c in ch] and got["accepted"] == acc and got["min_margin"] >= 1e-9 and got["swaps"] >= want_swaps
              and got["Es"] == [float(e) for e in Es] and got["lfields"] == [float(v) for v in lf])         # same IEEE operations: equal bit for bit
        return ok, "%d iterations, accepted %d, %d array swaps, closest decision margin %.2e" % (iters, acc, got["swaps"], got["min_margin"])
    ok = _finish(path, body, check)
    if ok:          # record how many swaps the tape exercises
        t = open(path).read().replace("@expected_accepted %d" % acc, "@expected_accepted %d\n@expected_swaps %d" % (acc, TR.replay_standard_mc_sk(TR.read_tape(path))["swaps"]))
        open(path, "w").write(t)
    return ok


def write_standard_rrgn(path, seed, N=16, K=3, beta=1.2, iters=6000, step=200, replica=0, want_undos=5):
    """standardMC on GraphRRGNormal(N, K) (Float64 couplings and local fields, RRG.jl:503-609): the model of csrc/spf_kernels.hpp and
    csrc/spf_team_kernel.hpp.  On 16 sites the same spin is accepted twice in a row often enough that the undo branch of update_cache!
    (RRG.jl:566-577: lfields <-> lfields_last over the neighbours) is on the tape."""
    A = O.gen_rrg(N, K, seed)
    J = O.gen_couplings_gauss(A, seed)
    C0 = O.init_config(seed, replica, N)
    Es, ch, acc, lf = O.standard_mc_spf(A, J, beta, iters, step, seed, C0, replica=replica)
    sites = [O.site_of(seed, g, N) for g in range(1, iters + 1)]
    us = [O.rand53(seed, g, replica) for g in range(1, iters + 1)]
    body = ["# RRRMC tape v1 — standardMC(X::GraphRRGNormal{%d}, beta, iters; step, C0) with every random draw pre-drawn: A and J as the" % K,
            "# fields of the reference's struct (rows of K neighbours / couplings), sites 1-based; uniforms consulted only when delta_energy > 0",
            "# (src/RRRMC.jl:39).  Written by tests/golden/make_tapes.py",
            "@kind standardMC_rrgn", "@N %d" % N, "@K %d" % K, "@beta %r" % beta, "@iters %d" % iters, "@step %d" % step, "@seed %d" % seed,
            "@replica %d" % replica,
            fmt_array("A", ("%d" % (v + 1) for v in A.reshape(-1))), fmt_array("J", (repr(float(v)) for v in J.reshape(-1))),
            fmt_array("C0", ("%016x" % int(c) for c in C0)),
            fmt_array("sites", ("%d" % (v + 1) for v in sites)), fmt_array("uniforms", (repr(u) for u in us)),
            fmt_array("expected_Es", (repr(float(e)) for e in Es)), fmt_array("expected_chunks", ("%016x" % int(c) for c in ch)),
            "@expected_accepted %d" % acc, fmt_array("expected_lfields", (repr(float(v)) for v in lf))]

    def check(t):
        got = TR.replay_standard_mc_rrgn(t)
        ok = (got["chunks"] == [int(c) for c in ch] and got["accepted"] == acc and got["min_margin"] >= 1e-9 and got["undos"] >= want_undos
              and got["Es"] == [float(e) for e in Es] and got["lfields"] == [float(v) for v in lf])         # same IEEE operations: equal bit for bit
        return ok, "%d iterations, accepted %d, %d undo swaps, closest decision margin %.2e" % (iters, acc, got["undos"], got["min_margin"])
    ok = _finish(path, body, check)
    if ok:
        t = open(path).read().replace("@expected_accepted %d" % acc, "@expected_accepted %d\n@expected_undos %d" % (acc, TR.replay_standard_mc_rrgn(TR.read_tape(path))["undos"]))
        open(path, "w").write(t)
    return ok


def write_rrr_skn(path, seed, N=10, beta=2.0, iters=2500, step=50, staged_thr=0.8, staged_thr_fact=5.0, replica=0):
    """rrrMC(X::SingleGraph) on GraphSKNormal(N) through DeltaECacheCont + DynamicSampler (DeltaE.jl:299-410, DynamicSamplers.jl): with N = 10
    every move re-weights all ten spins, so refresh! (every max(N, 100) setindex! calls) happens every tenth move."""
    Jm = O.gen_sk_gauss(N, seed)
    C0 = O.init_config(seed, replica, N)
    Es, ch, acc, staged, dEs, z = O.rrr_mc_skn(Jm, beta, iters, step, seed, C0, replica=replica, staged_thr=staged_thr,
                                               staged_thr_fact=staged_thr_fact, want_cache=True)
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32)
    umove, uacc = [], []
    for g in range(1, iters + 1):          # RRR stream: sub 0 words 0,1 = rand(dynsmp)'s uniform, sub 1 = rand() of `rand() < c`
        w0 = O.philox([g & 0xFFFFFFFF, g >> 32, replica, 8], key)
        w1 = O.philox([g & 0xFFFFFFFF, g >> 32, replica, 8 | (1 << 8)], key)
        umove.append(u53((int(w0[0]) << 32) | int(w0[1])))
        uacc.append(u53((int(w1[0]) << 32) | int(w1[1])))
    body = ["# RRRMC tape v1 — rrrMC(X::GraphSKNormal, beta, iters; step, C0, staged_thr, staged_thr_fact) (SingleGraph method, src/RRRMC.jl:149-219)",
            "# with every random draw pre-drawn: u_move = rand() inside rand(dynsmp) (src/DynamicSamplers.jl:154), u_accept = rand() of",
            "# `rand() < c` (src/RRRMC.jl:192,202; drawn at every iteration).  Written by tests/golden/make_tapes.py",
            "@kind rrrMC_skn", "@N %d" % N, "@beta %r" % beta, "@iters %d" % iters, "@step %d" % step, "@staged_thr %r" % staged_thr,
            "@staged_thr_fact %r" % staged_thr_fact, "@seed %d" % seed, "@replica %d" % replica,
            fmt_array("J", (repr(float(v)) for v in Jm.reshape(-1))), fmt_array("C0", ("%016x" % int(c) for c in C0)),
            fmt_array("u_move", (repr(u) for u in umove)), fmt_array("u_accept", (repr(u) for u in uacc)),
            fmt_array("expected_Es", (repr(float(e)) for e in Es)), fmt_array("expected_chunks", ("%016x" % int(c) for c in ch)),
            "@expected_accepted %d" % acc, "@expected_staged_its %d" % staged,
            fmt_array("expected_dEs", (repr(float(v)) for v in dEs)), "@expected_z %r" % z]

    def check(t):
        got = TR.replay_rrr_single_sk(t)
        ok = (got["chunks"] == [int(c) for c in ch] and got["accepted"] == acc and got["staged_its"] == staged and got["min_margin"] >= 1e-9
              and np.allclose(got["Es"], Es, rtol=1e-12, atol=1e-12) and np.allclose(got["dEs"], dEs, rtol=1e-12, atol=1e-12)
              and abs(got["z"] - z) <= 1e-12 * max(1.0, abs(z)) and got["refreshes"] >= 3 and 0 < staged < iters)
        return ok, "%d iterations, accepted %d, staged %d, %d refresh! calls, closest decision margin %.2e" % (iters, acc, staged, got["refreshes"], got["min_margin"])
    return _finish(path, body, check)


def write_rrr_rrgn(path, seed, N=64, K=3, beta=2.0, iters=3000, step=100, staged_thr=0.8, staged_thr_fact=5.0, replica=0):
    """rrrMC(X::SingleGraph) on GraphRRGNormal(N, K) through DeltaECacheCont + DynamicSampler: the model and size range of
    csrc/cont_wave_kernel.hpp (a wavefront per replica, N >= 64).  A move re-weights K + 1 = 4 spins, so refresh! (every max(N, 100)
    setindex! calls) happens every 25 moves."""
    A = O.gen_rrg(N, K, seed)
    J = O.gen_couplings_gauss(A, seed)
    C0 = O.init_config(seed, replica, N)
    Es, ch, stats, _t = O.cont_sparse("rrr", A, J, beta, iters, step, seed, C0, replica=replica, staged_thr=staged_thr, staged_thr_fact=staged_thr_fact)
    acc, staged = int(stats[0]), int(stats[1])
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32)
    umove, uacc = [], []
    for g in range(1, iters + 1):          # RRR stream: sub 0 words 0,1 = rand(dynsmp)'s uniform, sub 1 = rand() of `rand() < c`
        w0 = O.philox([g & 0xFFFFFFFF, g >> 32, replica, 8], key)
        w1 = O.philox([g & 0xFFFFFFFF, g >> 32, replica, 8 | (1 << 8)], key)
        umove.append(u53((int(w0[0]) << 32) | int(w0[1])))
        uacc.append(u53((int(w1[0]) << 32) | int(w1[1])))
    body = ["# RRRMC tape v1 — rrrMC(X::GraphRRGNormal{%d}, beta, iters; step, C0, staged_thr, staged_thr_fact) (SingleGraph method, src/RRRMC.jl:149-219)" % K,
            "# with every random draw pre-drawn: u_move = rand() inside rand(dynsmp) (src/DynamicSamplers.jl:154), u_accept = rand() of",
            "# `rand() < c` (src/RRRMC.jl:192,202; drawn at every iteration).  A and J are the fields of the reference's struct.  Written by tests/golden/make_tapes.py",
            "@kind rrrMC_rrgn", "@N %d" % N, "@K %d" % K, "@beta %r" % beta, "@iters %d" % iters, "@step %d" % step, "@staged_thr %r" % staged_thr,
            "@staged_thr_fact %r" % staged_thr_fact, "@seed %d" % seed, "@replica %d" % replica,
            fmt_array("A", ("%d" % (v + 1) for v in A.reshape(-1))), fmt_array("J", (repr(float(v)) for v in J.reshape(-1))),
            fmt_array("C0", ("%016x" % int(c) for c in C0)),
            fmt_array("u_move", (repr(u) for u in umove)), fmt_array("u_accept", (repr(u) for u in uacc)),
            fmt_array("expected_Es", (repr(float(e)) for e in Es)), fmt_array("expected_chunks", ("%016x" % int(c) for c in ch)),
            "@expected_accepted %d" % acc, "@expected_staged_its %d" % staged]

    def check(t):
        got = TR.replay_rrr_single_sk(t)
        ok = (got["chunks"] == [int(c) for c in ch] and got["accepted"] == acc and got["staged_its"] == staged and got["min_margin"] >= 1e-9
              and np.allclose(got["Es"], Es, rtol=1e-12, atol=1e-12) and got["refreshes"] >= 3 and 0 < staged < iters)
        return ok, "%d iterations, accepted %d, staged %d, %d refresh! calls, %d undo swaps, closest decision margin %.2e" % (
            iters, acc, staged, got["refreshes"], got["swaps"], got["min_margin"])
    return _finish(path, body, check)


def write_rrr_bkl_rrg(path, seed, bkl, N=64, K=3, beta=2.0, iters=4000, step=100, staged_thr=0.5, staged_thr_fact=5.0, replica=0):
    """rrrMC(X::SingleGraph) (bkl = False) or bklMC (bkl = True) on GraphRRG{Int,(-1,1),K} with DeltaECache{Int,L}."""
    A = O.gen_rrg(N, K, seed)
    J = O.gen_couplings(A, seed)
    C0 = O.init_config(seed, replica, N)
    Es, ch, acc, staged, its, pos, sizes = O.rrr_sparse(A, J.astype(np.int32), beta, iters, step, seed, C0, replica=replica, staged_thr=staged_thr,
                                                        staged_thr_fact=staged_thr_fact, bkl=bkl, want_cache=True)
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32)
    ndraw = acc + 1 if bkl else iters            # bklMC draws once per MOVE (the last one may end the run at a sample point)
    ucls, umem, uthird = [], [], []
    for g in range(1, ndraw + 1):                # RRR stream: sub 0 = (class uniform, member word), sub 1 = `rand() < c`, sub 2 = rand_skip
        w0 = O.philox([g & 0xFFFFFFFF, g >> 32, replica, 8], key)
        w1 = O.philox([g & 0xFFFFFFFF, g >> 32, replica, 8 | ((2 if bkl else 1) << 8)], key)
        ucls.append(u53((int(w0[0]) << 32) | int(w0[1])))
        umem.append((int(w0[2]) << 32) | int(w0[3]))
        uthird.append(u53((int(w1[0]) << 32) | int(w1[1])))
    kind = "bklMC_rrg" if bkl else "rrrMC_rrg"
    body = ["# RRRMC tape v1 — %s on GraphRRG{Int,(-1,1),%d} with DeltaECache{Int,L} (src/DeltaE.jl:62-295), every random draw pre-drawn:" % ("bklMC (src/RRRMC.jl:311-359)" if bkl else "rrrMC(X::SingleGraph) (src/RRRMC.jl:149-219)", K),
            "# u_class = rand() of rand_move (src/DeltaE.jl:148), u_member -> rand(1:t) as floor(u * t / 2^64) + 1 (src/ArraySets.jl:83),",
            "# " + ("u_skip = rand() of rand_skip (src/DeltaE.jl:141-144); one set of draws per MOVE." if bkl else "u_accept = rand() of `rand() < c` (src/RRRMC.jl:192,202).") + "  Written by tests/golden/make_tapes.py",
            "@kind " + kind, "@N %d" % N, "@K %d" % K, "@beta %r" % beta, "@iters %d" % iters, "@step %d" % step,
            "@staged_thr %r" % staged_thr, "@staged_thr_fact %r" % staged_thr_fact, "@seed %d" % seed, "@replica %d" % replica,
            fmt_array("A", ("%d" % (v + 1) for v in A.reshape(-1))), fmt_array("J", ("%d" % v for v in J.reshape(-1))),
            fmt_array("C0", ("%016x" % int(c) for c in C0)),
            fmt_array("u_class", (repr(u) for u in ucls)), fmt_array("u_member", ("%d" % u for u in umem)),
            fmt_array("u_skip" if bkl else "u_accept", (repr(u) for u in uthird)),
            fmt_array("expected_Es", ("%d" % int(e) for e in Es)), fmt_array("expected_chunks", ("%016x" % int(c) for c in ch)),
            "@expected_accepted %d" % acc, "@expected_staged_its %d" % staged, "@expected_iters_done %d" % its,
            fmt_array("expected_sizes", ("%d" % int(v) for v in sizes)), fmt_array("expected_pos", ("%d" % (int(v) + 1) for v in pos))]

    def check(t):
        got = TR.replay_rrr_bkl_rrg(t)
        ok = (got["Es"] == [int(e) for e in Es] and got["chunks"] == [int(c) for c in ch] and got["accepted"] == acc and got["staged_its"] == staged
              and got["iters_done"] == its and got["sizes"] == [int(v) for v in sizes] and got["pos"] == [int(v) + 1 for v in pos]
              and got["min_margin"] >= 1e-9 and (bkl or 0 < staged < iters))
        return ok, "%d iterations, %d moves accepted, staged %d, closest decision margin %.2e" % (its, acc, staged, got["min_margin"])
    return _finish(path, body, check)


def write_wtm_rrg(path, seed, N=64, K=3, beta=1.0, samples=40, step=60.0, replica=0):
    """wtmMC on GraphRRG{Int,(-1,1),K}: waiting times -tau log1p(-rand()), the earliest spin moves."""
    A = O.gen_rrg(N, K, seed)
    J = O.gen_couplings(A, seed)
    C0 = O.init_config(seed, replica, N)
    Es, ch, moves, t, Efin = O.wtm_mc_sparse(A, J.astype(np.int32), beta, samples, step, seed, C0, replica=replica)
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32)
    ndraw = N + (moves + 1) * (K + 1)
    us = []
    for n in range(ndraw):                       # WTM stream (oracle/rrrmc_oracle.c): the n-th uniform = 53-bit word (n & 1) of block n >> 1, tag 11, call 0
        blk = n >> 1
        w = O.philox([blk & 0xFFFFFFFF, blk >> 32, replica, 11], key)
        h = n & 1
        us.append(u53((int(w[2 * h]) << 32) | int(w[2 * h + 1])))
    body = ["# RRRMC tape v1 — wtmMC(X::GraphRRG{Int,(-1,1),%d}, beta, samples; step, C0) (src/RRRMC.jl:376-426, src/WaitingTimes.jl) with every" % K,
            "# rand() of gen_wt pre-drawn, in call order: the N initial times of THeap(X, C, beta), then per move the moved spin and its neighbours.",
            "# Written by tests/golden/make_tapes.py",
            "@kind wtmMC_rrg", "@N %d" % N, "@K %d" % K, "@beta %r" % beta, "@samples %d" % samples, "@step %r" % step,
            "@seed %d" % seed, "@replica %d" % replica,
            fmt_array("A", ("%d" % (v + 1) for v in A.reshape(-1))), fmt_array("J", ("%d" % v for v in J.reshape(-1))),
            fmt_array("C0", ("%016x" % int(c) for c in C0)), fmt_array("uniforms", (repr(u) for u in us)),
            fmt_array("expected_Es", ("%d" % int(e) for e in Es)), fmt_array("expected_chunks", ("%016x" % int(c) for c in ch)),
            "@expected_num_moves %d" % moves, "@expected_t %r" % t]

    def check(tp):
        got = TR.replay_wtm_rrg(tp)
        ok = (got["Es"] == [int(e) for e in Es] and got["chunks"] == [int(c) for c in ch] and got["num_moves"] == moves
              and abs(got["t"] - t) <= 1e-12 * t and got["min_margin"] >= 1e-9 and len(Es) == samples and moves > 5 * N)
        return ok, "%d samples, %d moves, global time %.4f, %d draws, closest decision margin %.2e" % (len(Es), moves, t, got["draws"], got["min_margin"])
    return _finish(path, body, check)


def write_eo_rrg(path, seed, N=64, K=3, tau=1.3, iters=3000, step=100, replica=0):
    """extremal_opt on GraphRRG{Int,(-1,1),K} with EOCache{Int,L}."""
    A = O.gen_rrg(N, K, seed)
    J = O.gen_couplings(A, seed)
    C0 = O.init_config(seed, replica, N)
    Es, ch, Emin, Cmin, itmin = O.extremal_opt_sparse(A, J.astype(np.int32), tau, iters, step, seed, C0, replica=replica)
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32)
    urank, umem = [], []
    for g in range(1, iters + 1):                # RRR stream sub 3 (oracle/rrrmc_oracle.c): words 0,1 -> rand() of r, words 2,3 -> the member index
        w = O.philox([g & 0xFFFFFFFF, g >> 32, replica, 8 | (3 << 8)], key)
        urank.append(u53((int(w[0]) << 32) | int(w[1])))
        umem.append((int(w[2]) << 32) | int(w[3]))
    body = ["# RRRMC tape v1 — extremal_opt(X::GraphRRG{Int,(-1,1),%d}, tau, iters; step, C0) (src/RRRMC.jl:474-521) with EOCache{Int,L}" % K,
            "# (src/DeltaE.jl:412-555), both draws of rand_move pre-drawn: u_rank = rand() of r = (1 - rand()) z (:487), u_member -> rand(1:t) as",
            "# floor(u * t / 2^64) + 1 (src/ArraySets.jl:83).  expected_Es = the E handed to the hook every `step` iterations.  Written by tests/golden/make_tapes.py",
            "@kind extremal_opt_rrg", "@N %d" % N, "@K %d" % K, "@tau %r" % tau, "@iters %d" % iters, "@step %d" % step,
            "@seed %d" % seed, "@replica %d" % replica,
            fmt_array("A", ("%d" % (v + 1) for v in A.reshape(-1))), fmt_array("J", ("%d" % v for v in J.reshape(-1))),
            fmt_array("C0", ("%016x" % int(c) for c in C0)),
            fmt_array("u_rank", (repr(u) for u in urank)), fmt_array("u_member", ("%d" % u for u in umem)),
            fmt_array("expected_Es", ("%d" % int(e) for e in Es)), fmt_array("expected_chunks", ("%016x" % int(c) for c in ch)),
            "@expected_Emin %d" % Emin, fmt_array("expected_Cmin", ("%016x" % int(c) for c in Cmin)), "@expected_itmin %d" % itmin]

    def check(t):
        got = TR.replay_eo_rrg(t)
        ok = (got["Es"] == [int(e) for e in Es] and got["chunks"] == [int(c) for c in ch] and got["Emin"] == Emin
              and got["Cmin"] == [int(c) for c in Cmin] and got["itmin"] == itmin and got["min_margin"] >= 1e-9 and 0 < itmin < iters)
        return ok, "%d iterations, Emin %d at iteration %d, closest decision margin %.2e" % (iters, Emin, itmin, got["min_margin"])
    return _finish(path, body, check)


if __name__ == "__main__":
    O.build()
    for seed in range(20261003, 20261003 + 50):
        if write_standard(os.path.join(HERE, "tape_rrg_n128.txt"), seed):
            break
    else:
        raise SystemExit("no seed gave a standardMC tape with a safe decision margin")
    for seed in range(20261003, 20261003 + 200):
        if write_quant(os.path.join(HERE, "tape_quant_nk16_m4.txt"), seed):
            break
    else:
        raise SystemExit("no seed gave an rrrMC tape with a safe decision margin")
    for seed in range(20261003, 20261003 + 200):      # staged_thr = 0: apply_move! and its undo on every rejected move
        if write_quant(os.path.join(HERE, "tape_quant_direct.txt"), seed, staged_thr=0.0):
            break
    else:
        raise SystemExit("no seed gave a direct-branch rrrMC tape with a safe decision margin")
    tries = lambda f, what: next((True for seed in range(20261003, 20261003 + 400) if f(seed)), None) or (_ for _ in ()).throw(SystemExit("no seed gave " + what))
    tries(lambda sd: write_standard_ea(os.path.join(HERE, "tape_ea_l2_d3.txt"), sd), "a GraphEA(2,3) tape with a safe margin")
    tries(lambda sd: write_standard_sk(os.path.join(HERE, "tape_skn_n24.txt"), sd, 24, False, 1.0, 6000, 200), "a GraphSKNormal(24) tape with a swap")
    tries(lambda sd: write_standard_sk(os.path.join(HERE, "tape_sk_n10.txt"), sd, 10, True, 1.0, 4000, 100), "a GraphSK(10) tape with a swap")
    tries(lambda sd: write_rrr_skn(os.path.join(HERE, "tape_rrr_skn_n10.txt"), sd), "an rrrMC(GraphSKNormal(10)) tape with a safe margin")
    tries(lambda sd: write_rrr_bkl_rrg(os.path.join(HERE, "tape_rrr_rrg_n64.txt"), sd, False), "an rrrMC(GraphRRG(64,3)) tape with both branches and a safe margin")
    tries(lambda sd: write_rrr_bkl_rrg(os.path.join(HERE, "tape_bkl_rrg_n64.txt"), sd, True, iters=20000, step=500), "a bklMC(GraphRRG(64,3)) tape with a safe margin")
    tries(lambda sd: write_wtm_rrg(os.path.join(HERE, "tape_wtm_rrg_n64.txt"), sd), "a wtmMC(GraphRRG(64,3)) tape with a safe margin")
    tries(lambda sd: write_eo_rrg(os.path.join(HERE, "tape_eo_rrg_n64.txt"), sd), "an extremal_opt(GraphRRG(64,3)) tape with a safe margin")
    tries(lambda sd: write_standard_rrgn(os.path.join(HERE, "tape_rrgn_n16.txt"), sd), "a GraphRRGNormal(16,3) tape with undo swaps and a safe margin")
    tries(lambda sd: write_rrr_rrgn(os.path.join(HERE, "tape_rrr_rrgn_n64.txt"), sd), "an rrrMC(GraphRRGNormal(64,3)) tape with both branches and a safe margin")
    tries(lambda sd: write_quant_f64(os.path.join(HERE, "tape_quant_qeat_l4_m8.txt"), sd), "an rrrMC(GraphQEAT(4, 2, 8)) tape with both branches and a safe margin")
