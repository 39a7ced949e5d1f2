"""Pure-Python model of the HIP sweep algorithm (bit-sliced replicas, dependency-level schedule,
bit-plane acceptance masks).  Test helper only: it lets the CPU test-suite check that the ALGORITHM the
kernels implement reproduces the sequential oracle bit for bit, without a GPU.
It mirrors rrrmc.jl_amd/csrc/sparse_kernels.hpp: plan_kernel, produce_chunk, consume_chunk.
"""
import math

import numpy as np

MASK32 = 0xFFFFFFFF
TAG_ACCEPT = 2


def plan_chunk(sites, A):
    """Level schedule of one chunk: returns (order, vecs): attempt indices sorted by level and the runs
    (start, n<=64) that stay inside one level."""
    N, K = A.shape
    W = np.zeros(N, np.int64)
    lvl = np.zeros(len(sites), np.int64)
    for t, i in enumerate(sites):
        l = max(W[i], W[A[i]].max()) + 1
        W[i] = l
        lvl[t] = l
    order = np.argsort(lvl, kind="stable")
    vecs = []
    pos = 0
    for l in range(1, int(lvl.max()) + 1 if len(sites) else 1):
        n = int((lvl == l).sum())
        q = pos
        while n > 0:
            m = min(n, 64)
            vecs.append((q, m))
            q += m
            n -= m
        pos = q
    return order, vecs


def accept_masks(philox, seed, g, group, T, always):
    """lt[n] = 32-bit mask over the group's replicas of [u(g, replica) < T[n]] (MSB-first bit planes)."""
    key = [seed & MASK32, seed >> 32]
    NT = len(T)
    lt = [MASK32 if always[n] else 0 for n in range(NT)]
    eq = [0 if always[n] else MASK32 for n in range(NT)]
    for pb in range(16):
        if not any(eq):
            break
        w4 = philox([g & MASK32, g >> 32, group, TAG_ACCEPT | (pb << 8)], key)
        for j in range(4):
            w = int(w4[j])
            sh = 63 - (pb * 4 + j)
            for n in range(NT):
                taum = MASK32 if (T[n] >> sh) & 1 else 0
                z = w ^ (~taum & MASK32)
                e2 = eq[n] & z
                lt[n] |= (eq[n] ^ e2) & taum
                eq[n] = e2
    return lt


def sweep(philox, site_of, threshold, A, J, beta, iters, step, seed, spins_bs, group=0, it0=0, C=960):
    """spins_bs: [N] python ints (32-replica words).  Returns (Es[nsamp][32], spins_bs, accepted[32])."""
    N, K = A.shape
    NT = (K + 1) // 2
    T, always = [], []
    for n in range(NT):
        t, a = threshold(math.exp(-beta * 2.0 * (K - 2 * n)))
        T.append(t)
        always.append(a)
    sp = [int(v) for v in spins_bs]

    def energy():
        U = [0] * 32
        for x in range(N):
            for k in range(K):
                u = sp[x] ^ sp[A[x, k]] ^ (MASK32 if J[x, k] < 0 else 0)
                for r in range(32):
                    U[r] += (u >> r) & 1
        return [U[r] - N * K // 2 for r in range(32)]

    E = energy()
    acc_tot = [0] * 32
    Es = []
    cur = 1
    while cur <= iters:
        if cur % step == 0:
            Es.append(list(E))
        nxt = (cur // step + 1) * step
        end = min(cur + C, nxt, iters + 1)
        gs = [it0 + it for it in range(cur, end)]
        sites = [site_of(seed, g, N) for g in gs]
        order, vecs = plan_chunk(sites, A)
        masks = {t: accept_masks(philox, seed, gs[t], group, T, always) for t in range(len(gs))}
        for (start, n) in vecs:
            reads = []
            for p in range(start, start + n):          # all lanes read before any lane writes
                t = int(order[p])
                i = sites[t]
                s = sp[i]
                cnt = [0] * 32
                nw = [0, 0, 0]
                for k in range(K):
                    u = s ^ sp[A[i, k]] ^ (MASK32 if J[i, k] < 0 else 0)
                    c0 = nw[0] & u
                    nw[0] ^= u
                    c1 = nw[1] & c0
                    nw[1] ^= c0
                    nw[2] ^= c1
                rej = 0
                for n_ in range(NT):
                    e = ~masks[t][n_] & MASK32
                    e &= nw[0] if n_ & 1 else ~nw[0] & MASK32
                    e &= nw[1] if n_ & 2 else ~nw[1] & MASK32
                    if K > 3:
                        e &= nw[2] if n_ & 4 else ~nw[2] & MASK32
                    rej |= e
                acc = ~rej & MASK32
                reads.append((i, s ^ acc, acc, nw))
            for (i, snew, acc, nw) in reads:
                sp[i] = snew
                for r in range(32):
                    if (acc >> r) & 1:
                        nn = ((nw[0] >> r) & 1) + 2 * ((nw[1] >> r) & 1) + 4 * ((nw[2] >> r) & 1)
                        E[r] += 2 * (K - 2 * nn)
                        acc_tot[r] += 1
        cur = end
    return Es, sp, acc_tot
