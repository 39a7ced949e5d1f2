"""Pure-Python model of the blocked dense-SK sweep (rrrmc.jl_amd/csrc/sk_block_kernel.hpp).  Test helper only.

The HIP kernel does not walk the chain one attempt at a time: it takes W = 64 attempts (a *block*; their sites and acceptance
uniforms do not depend on the state), gathers the W attempted sites' local fields into a *window* (lane m = attempt m of the
block), and lets one wavefront per replica run the block's decisions on the window alone:

  * every lane keeps its verdict `u_m < exp(-beta f_m)` up to date, so the next accepted attempt is the first set bit of a ballot
    — rejected attempts cost nothing;
  * an accepted attempt k updates the lanes m > k exactly as update_cache! (src/graphs/SK.jl:239-276) updates those sites'
    entries of lfields / lfields_last (the array swap of :247-250 included), and their verdicts are re-evaluated;

after which all threads apply the block's accepted moves, in order, to the full field arrays.  Every field receives the
reference's sequence of IEEE operations, whether it is tracked in the window or in the bulk arrays.  This model executes exactly
that schedule with numpy float64 scalars, so the CPU suite can check the ALGORITHM against the sequential oracle bit for bit.
"""
import numpy as np

W = 64


def run_chain(O, J, beta, iters, step, seed, chunks, it0=0, replica=0, state=None, hform=False):
    """One chain of standardMC on GraphSKNormal through the blocked schedule.
    Returns (Es, chunks_out, accepted, lfields, state) with state = (lfl, move_last) for a resumed call.

    hform = True models round 4's bulk phase: the registers hold H_j = sigma_j lfields[j] (the field without the site's own sign: a flip of i
    adds sigma_i' 4 J_ij to EVERY H_j with one wave-uniform sign, and leaves H_i alone), the bulk update of a block is one fused
    multiply-add per (attempt, site) with a multiplier in {+1, -1, 0} and no branch, lfields_last is copied only where it can be
    read (before the block's last accepted move and before a move that the next one undoes), and the spins flip once per block."""
    N = J.shape[0]
    ch = np.array(chunks, np.uint64, copy=True)
    sp = np.array([(int(ch[x >> 6]) >> (x & 63)) & 1 for x in range(N)], np.int64)
    E, lf = O.skn_energy(J, ch, want_fields=True)
    lf = lf.copy()
    if state is None:
        lfl, mlast = np.zeros(N), -1
    else:
        lf, lfl, mlast, E = state[0].copy(), state[1].copy(), state[2], state[3]
    J4 = 4.0 * J                                            # exact
    if hform:
        sg = lambda: np.where(sp == 1, 1.0, -1.0)
        H = sg() * lf                                                      # x * +-1.0 is exact
        sl = sg()
        if mlast >= 0:
            sl[mlast] = -sl[mlast]                                         # lfields_last[move_last] belongs to the spin before its last flip
        Hl = sl * lfl
    Es, acc_total = [], 0
    next_sample = step
    nblk = (iters + W - 1) // W
    for blk in range(nblk):
        nv = min(W, iters - blk * W)
        its = [blk * W + m + 1 for m in range(nv)]                       # iteration numbers of this call
        sites = [O.site_of(seed, it0 + t, N) for t in its]
        us = [O.rand53(seed, it0 + t, replica) for t in its]
        # ---- gather the window
        if hform:
            sgn = sg()
            f = np.array([sgn[s] * H[s] for s in sites])
            fl = np.array([(-sgn[s] if s == mlast else sgn[s]) * Hl[s] for s in sites])
        else:
            f = np.array([lf[s] for s in sites])
            fl = np.array([lfl[s] for s in sites])
        ws = np.array([sp[s] for s in sites], np.int64)

        def verdict(m):
            x = -beta * f[m]
            return x >= 0.0 or us[m] < O.det_exp(x)

        ok = [verdict(m) for m in range(nv)]
        ws0 = np.zeros(nv, np.int64)                                     # the moved spin before its flip, per accepted attempt
        moves = []                                                       # (k, swapped)
        pos = 0
        while True:
            k = next((m for m in range(pos, nv) if ok[m]), None)
            if k is None:
                break
            while next_sample <= its[k]:                                 # sample BEFORE the move (RRRMC.jl:104-108)
                Es.append(E)
                next_sample += step
            dE = f[k]
            ws0[k] = ws[k]
            E = E + dE
            acc_total += 1
            swapped = mlast == sites[k]
            if swapped:                                                  # SK.jl:247-250
                for m in range(k + 1, nv):
                    f[m], fl[m] = fl[m], f[m]
                    if sites[m] == sites[k]:
                        ws[m] ^= 1
            else:
                s_new = ws[k] ^ 1
                for m in range(k + 1, nv):
                    fl[m] = f[m]
                    if sites[m] == sites[k]:
                        f[m] = -f[m]                                      # lfields[move] = -lfm, SK.jl:263-264
                        ws[m] ^= 1
                    else:
                        d = J4[sites[k], sites[m]]
                        f[m] = f[m] + (-d if (ws[m] ^ s_new) else d)
                mlast = sites[k]
            for m in range(k + 1, nv):
                ok[m] = verdict(m)
            moves.append((k, swapped))
            pos = k + 1
        while next_sample <= its[-1]:
            Es.append(E)
            next_sample += step
        if hform:
            # what the deciding wavefront leaves behind: per attempt a multiplier, a copy flag, a swap flag; the spins flip once
            mult, cpy, swp = np.zeros(nv), np.zeros(nv, bool), np.zeros(nv, bool)
            for n, (k, swapped) in enumerate(moves):
                swp[k] = swapped
                if not swapped:
                    mult[k] = -1.0 if ws0[k] else 1.0                    # sigma_i' = the moved spin AFTER its flip
                    cpy[k] = n + 1 == len(moves) or moves[n + 1][1]      # last accepted move of the block, or the next one undoes it
            for k in range(nv):
                if cpy[k]:
                    Hl[:] = H
                if swp[k]:
                    H, Hl = Hl, H
                H[:] = H + J4[sites[k]] * mult[k]                        # one fused multiply-add: the product with +-1 / 0 is exact
            for k, _ in moves:
                sp[sites[k]] ^= 1
            continue
        # ---- bulk: the block's accepted moves on the full arrays
        for k, swapped in moves:
            i = sites[k]
            sp[i] ^= 1                                                   # spinflip!, Interface.jl:89-92
            if swapped:
                lf, lfl = lfl, lf
                continue
            lfm = lf[i]
            sig = np.where(sp ^ sp[i], -1.0, 1.0)
            lfl[:] = lf
            lf[:] = lf + sig * J4[i]                                     # lfj + 4 J sigma; sigma * 4J is exact
            lfl[i] = lfm
            lf[i] = -lfm
    if hform:
        lf = sg() * H
        sl = sg()
        if mlast >= 0:
            sl[mlast] = -sl[mlast]
        lfl = sl * Hl
    for x in range(N):
        w, b = x >> 6, x & 63
        ch[w] = np.uint64((int(ch[w]) & ~(1 << b)) | (int(sp[x]) << b))
    return np.array(Es), ch, acc_total, lf, (lf, lfl, mlast, E)

