"""RNG-free replays of the reference's samplers in plain Python — TEST INFRASTRUCTURE, independent of oracle/rrrmc_oracle.c.

A *tape* (tests/golden/tape_*.txt, written by tests/golden/make_tapes.py) holds a graph, a start configuration and every random
draw of a run, pre-drawn: sites and acceptance uniforms for standardMC, (class uniform, member word, acceptance uniform) for
rrrMC.  With the draws fixed the reference's loop is a deterministic function of its own graph code, so the same tape can be
replayed (a) here, statement by statement after the Julia sources cited below, (b) by tests/replay_tape.jl through the reference's
OWN functions wherever Julia and RRRMC.jl exist, (c) by the HIP library (tests/test_tapes.py).  The functions mirror
tests/replay_tape.jl line by line; indices are 1-based as in the tape."""
import math


def read_tape(path):
    """Sections: '@name n' followed by n whitespace-separated tokens (possibly over several lines); scalars: '@name value'."""
    out, name, want, buf = {}, None, 0, []
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        if line.startswith("@"):
            parts = line[1:].split()
            name = parts[0]
            if len(parts) == 3 and parts[1] == "array":
                want, buf = int(parts[2]), []
                out[name] = buf
                if want == 0:
                    name = None
            else:
                out[name] = parts[1] if len(parts) == 2 else parts[1:]
                name = None
            continue
        if name is not None:
            buf.extend(line.split())
            if len(buf) >= want:
                assert len(buf) == want, (name, len(buf), want)
                name = None
    return out


def bits_of_chunks(chunks, N):
    """Config.s as a list of 0/1 (index 0 = site 1): bit (i-1)&63 of chunk (i-1)>>6 (Base.BitArray layout, Common.jl:15-23)."""
    return [(int(chunks[i >> 6]) >> (i & 63)) & 1 for i in range(N)]


def chunks_of_bits(s):
    ch = [0] * ((len(s) + 63) // 64)
    for i, b in enumerate(s):
        ch[i >> 6] |= int(b) << (i & 63)
    return ch


# ---- GraphRRG{Int,(-1,1),K}: src/graphs/RRG.jl -------------------------------------------------------------------------
def rrg_energy(A, J, s):
    """energy (RRG.jl:164-189): lfields[x] = 2 * lf, lf = -sum_k J[x][k] sigma_x sigma_y; E = sum lf / 2."""
    N = len(A)
    lfields, n = [0] * N, 0
    for x in range(N):
        sx, lf = 2 * s[x] - 1, 0
        for y, Jxy in zip(A[x], J[x]):
            lf -= Jxy * sx * (2 * s[y - 1] - 1)
        n += lf
        lfields[x] = 2 * lf
    assert n % 2 == 0
    return n // 2, lfields


def rrg_spinflip(A, J, s, lfields, i):
    """spinflip! (Interface.jl:89-92) = flip the bit, then update_cache! (RRG.jl:191-234; the move_last fast path gives the same
    integers and is not restated)."""
    s[i - 1] ^= 1
    for y, Jxy in zip(A[i - 1], J[i - 1]):
        sxy = 1 - 2 * (s[i - 1] ^ s[y - 1])
        lfields[y - 1] -= 4 * sxy * Jxy
    lfields[i - 1] = -lfields[i - 1]


def replay_standard_mc(tape):
    """standardMC (RRRMC.jl:81-127) with rand(1:N) -> tape site, rand() -> tape uniform (consulted only when dE > 0, :39)."""
    N, K = int(tape["N"]), int(tape["K"])
    beta, iters, step = float(tape["beta"]), int(tape["iters"]), int(tape["step"])
    A = [[int(v) for v in tape["A"][x * K:(x + 1) * K]] for x in range(N)]
    J = [[int(v) for v in tape["J"][x * K:(x + 1) * K]] for x in range(N)]
    s = bits_of_chunks([int(c, 16) for c in tape["C0"]], N)
    sites = [int(v) for v in tape["sites"]]
    us = [float(v) for v in tape["uniforms"]]
    E, lfields = rrg_energy(A, J, s)
    Es, accepted, flips, min_margin = [], 0, [], float("inf")
    for it in range(1, iters + 1):
        if it % step == 0:
            Es.append(E)
        i = sites[it - 1]
        dE = -lfields[i - 1]                               # delta_energy, RRG.jl:236-244
        x = -beta * dE
        ok = x >= 0
        if not ok:
            p = math.exp(x)
            ok = us[it - 1] < p                            # accept, RRRMC.jl:39
            min_margin = min(min_margin, abs(us[it - 1] - p) / p)
        flips.append(1 if ok else 0)
        if not ok:
            continue
        rrg_spinflip(A, J, s, lfields, i)
        E += dE
        accepted += 1
    assert E == rrg_energy(A, J, s)[0]                     # the reference's own test: test/runtests.jl:12-20
    return {"Es": Es, "chunks": chunks_of_bits(s), "accepted": accepted, "flips": flips, "min_margin": min_margin}


# ---- GraphQuant over GraphRRG slices under rrrMC: src/graphs/QT.jl, src/DeltaE.jl, src/ArraySets.jl ------------------------
class ArraySet:                                                # ArraySets.jl:19-85 (1-based members, v[0] unused)
    def __init__(self, N):
        self.v, self.pos, self.t = [0] * (N + 1), [0] * (N + 1), 0

    def push(self, i):
        self.t += 1
        self.v[self.t] = i
        self.pos[i] = self.t

    def delete(self, i):
        p = self.pos[i]
        self.v[p] = self.v[self.t]
        self.pos[self.v[p]] = p
        self.pos[i] = 0
        self.t -= 1


def qt_neighbors(N, Nk, i):                                    # QT.jl:105-108
    return (i - Nk + N * (i <= Nk), i + Nk - N * (i + Nk > N))


def qt_delta(s, N, Nk, fourK, move):                           # QT.jl:86-103
    k1, k2 = qt_neighbors(N, Nk, move)
    sk, s1, s2 = s[move - 1], s[k1 - 1], s[k2 - 1]
    return ((1 if sk == s1 else 0) - (1 if sk != s2 else 0)) * fourK


def replay_rrr_quant(tape, exp=math.exp):
    """rrrMC(X::DoubleGraph) (RRRMC.jl:221-290) on GraphQuant(Nk, M, Gamma, beta, GraphRRG{Int,(-1,1),K}, A, J) with the three draws
    of an iteration taken from the tape: rand_move's rand() and rand(1:t) (DeltaE.jl:148,164; ArraySets.jl:83), accept's rand()
    (RRRMC.jl:43)."""
    Nk, K, M = int(tape["Nk"]), int(tape["K"]), int(tape["M"])
    N = Nk * M
    beta, Gamma = float(tape["beta"]), float(tape["Gamma"])
    iters, step = int(tape["iters"]), int(tape["step"])
    staged_thr, staged_thr_fact = float(tape["staged_thr"]), float(tape["staged_thr_fact"])
    fourK = round(2 / beta * math.log(1 / math.tanh(beta * Gamma / M)), 8)      # QT.jl:165
    assert fourK == float(tape["fourK"])
    A = [[int(v) for v in tape["A"][x * K:(x + 1) * K]] for x in range(Nk)]
    # slices: GraphRRG{Int,(-1,1),K} (the default), or — "@slices f64" — sparse Float64 graphs on the same table: GraphQEAT =
    # GraphQuant{fourK,GraphEANormal{twoD}} (src/QAliases.jl:50-83).  Without repeated neighbours in a row (L > 2) GraphEANormal's energy /
    # update_cache! (EA.jl:584-653) are the operations of GraphRRGNormal's (RRG.jl:546-617), restated above as rrgn_energy / rrgn_update_cache.
    f64 = tape.get("slices", "int") == "f64"
    J = [[(float(v) if f64 else int(v)) for v in tape["J"][x * K:(x + 1) * K]] for x in range(Nk)]
    assert not f64 or all(len(set(row)) == K for row in A)
    s = bits_of_chunks([int(c, 16) for c in tape["C0"]], N)
    u_cls = [float(v) for v in tape["u_class"]]
    u_mem = [int(v) for v in tape["u_member"]]
    u_acc = [float(v) for v in tape["u_accept"]]
    # energy(X::GraphQuant, C) (QT.jl:185-199): energy0 * fourK / 4 + sum_k energy(X1[k], C1[k]) / M; every slice keeps its own cache
    n0 = 0
    for i in range(1, Nk + 1):
        sj = s[i + (M - 1) * Nk - 1]
        for k in range(1, M + 1):
            sk = s[i + (k - 1) * Nk - 1]
            n0 -= 1 - 2 * (sk ^ sj)
            sj = sk
    E = n0 * fourK / 4
    slices = []
    for k in range(M):
        sl = s[k * Nk:(k + 1) * Nk]
        if f64:
            Ek, lf, lfl, ml = rrgn_energy(A, J, sl)
            slices.append([sl, lf, lfl, ml])
        else:
            Ek, lf = rrg_energy(A, J, sl)
            slices.append([sl, lf])
        E += Ek / M
    # DeltaECache{Float64,2}(X0 = GraphQT, C, (0.0, fourK), beta) (DeltaE.jl:74-103)
    dElist, L = (0.0, fourK), 2
    sets, pos = [ArraySet(N) for _ in range(2 * L)], [0] * (N + 1)
    for i in range(1, N + 1):
        dE = qt_delta(s, N, Nk, fourK, i)
        a = dElist.index(abs(dE)) + 1                          # findk
        up = dE > 0 or (dE == 0 and s[i - 1] == 1)
        k = a + L * (1 if up else 0)
        pos[i] = k
        sets[k - 1].push(i)
    ft = [exp(-beta * d) for d in dElist]
    fcls = lambda k: ft[k - L - 1] if k > L else 1.0           # get_class_f, DeltaE.jl:138-139
    T = [sets[k - 1].t * fcls(k) for k in range(1, 2 * L + 1)]
    z = 0.0
    for x in T:
        z += x
    lam = staged_thr_fact / N
    Es, accepted, staged_its, acc_rate, margin = [], 0, 0, 0.5, float("inf")

    def new_class(j):
        dE1 = qt_delta(s, N, Nk, fourK, j)
        return dElist.index(abs(dE1)) + 1 + L * (1 if (dE1 > 0 or (dE1 == 0 and s[j - 1] == 1)) else 0)

    def flip_all(move):                                        # spinflip!(X::GraphQuant, C, move): bit + update_cache! (QT.jl:172-183)
        s[move - 1] ^= 1
        k, i = (move - 1) // Nk, (move - 1) % Nk + 1
        if f64:                                                # C1[k] flipped, then update_cache!(X1[k], C1[k], i) with its undo path
            slices[k][0][i - 1] ^= 1
            slices[k][3] = rrgn_update_cache(A, J, slices[k][0], slices[k][1], slices[k][2], slices[k][3], i)
            return
        rrg_spinflip(A, J, slices[k][0], slices[k][1], i)

    def residual(move):                                        # delta_energy_residual (QT.jl:270-281)
        k, i = (move - 1) // Nk, (move - 1) % Nk + 1
        return (-slices[k][1][i - 1]) / M

    def apply_move(move):                                      # DeltaE.jl:232-295
        nonlocal z
        flip_all(move)
        zp = z
        for j in qt_neighbors(N, Nk, move):
            k0, k1 = pos[j], new_class(j)
            if k0 == k1:
                continue
            f0, f1 = fcls(k0), fcls(k1)
            T[k0 - 1] -= f0
            T[k1 - 1] += f1
            zp += f1 - f0
            sets[k0 - 1].delete(j)
            sets[k1 - 1].push(j)
            pos[j] = k1
        k0 = pos[move]
        k1 = k0 - L * (2 * (1 if k0 > L else 0) - 1)
        f0, f1 = fcls(k0), fcls(k1)
        T[k0 - 1] -= f0
        T[k1 - 1] += f1
        zp += f1 - f0
        sets[k0 - 1].delete(move)
        sets[k1 - 1].push(move)
        pos[move] = k1
        c = z / zp
        z = zp
        return c

    for it in range(1, iters + 1):
        if it % step == 0:
            Es.append(E)
        # rand_move (DeltaE.jl:146-167)
        r = u_cls[it - 1] * z
        k, cT = 0, 0.0
        for k in range(1, 2 * L + 1):
            cT += T[k - 1]
            margin = min(margin, abs(r - cT) / z)
            if r < cT:
                break
        if not (r < cT):
            while T[k - 1] == 0:
                k -= 1
        dE0 = -dElist[k - 1] if k <= L else dElist[k - L - 1]
        t = sets[k - 1].t
        move = sets[k - 1].v[((u_mem[it - 1] * t) >> 64) + 1]   # rand(1:t) as floor(u64 * t / 2^64) + 1
        acc = False

        def accept(c, x):                                      # RRRMC.jl:40-44
            nonlocal margin
            if c >= 1 and x >= 0:
                return True
            a = c * exp(x)
            margin = min(margin, abs(a - 1) / a)
            if a >= 1:
                return True
            margin = min(margin, abs(u_acc[it - 1] - a) / a)
            return u_acc[it - 1] < a

        if acc_rate < staged_thr:
            staged_its += 1
            # step_rrr (RRRMC.jl:131-138): compute_staged! + compute_reverse_probabilities! (DeltaE.jl:184-230)
            flip_all(move)
            staged = []
            for j in qt_neighbors(N, Nk, move):
                k0, k1 = pos[j], new_class(j)
                if k0 != k1:
                    staged.append((j, k0, k1))
            k0 = pos[move]
            staged.append((move, k0, k0 - L * (2 * (1 if k0 > L else 0) - 1)))
            flip_all(move)
            Tp, zp = list(T), z
            for _, k0, k1 in staged:
                f0, f1 = fcls(k0), fcls(k1)
                Tp[k0 - 1] -= f0
                Tp[k1 - 1] += f1
                zp += f1 - f0
            c = z / zp
            dE1 = residual(move)
            if accept(c, -beta * dE1):
                flip_all(move)
                for j, k0, k1 in staged:                       # apply_staged! (DeltaE.jl:169-182)
                    sets[k0 - 1].delete(j)
                    sets[k1 - 1].push(j)
                    pos[j] = k1
                T[:] = Tp
                z = zp
                E += dE0 + dE1
                accepted += 1
                acc = True
        else:
            dE1 = residual(move)
            c = apply_move(move)
            if accept(c, -beta * dE1):
                E += dE0 + dE1
                accepted += 1
                acc = True
            else:
                apply_move(move)
        acc_rate = acc_rate * (1 - lam) + (1.0 if acc else 0.0) * lam
    return {"Es": Es, "chunks": chunks_of_bits(s), "accepted": accepted, "staged_its": staged_its, "min_margin": margin,
            "sizes": [st.t for st in sets], "pos": pos[1:]}


# ---- GraphEA{Int,(-1,1),2D}: src/graphs/EA.jl (a site may list a neighbour twice: L = 2) ------------------------------------
def ea_unique_neighbours(A):
    """uA (EA.jl:155-158): every second entry of the sorted neighbour tuple when the lattice has L = 2 (A[1][1] == A[1][2])."""
    isL2 = A[0][0] == A[0][1]
    return [list(a[::2]) if isL2 else list(a) for a in A]


def ea_energy(A, J, s):
    """energy (EA.jl:195-222): as GraphRRG's; resets the undo state (move_last = 0, lfields_last = 0)."""
    E, lfields = rrg_energy(A, J, s)
    return E, lfields, [0] * len(A), 0


def ea_update_cache(A, uA, J, s, lfields, lfields_last, move_last, move):
    """update_cache! (EA.jl:224-264), called AFTER the bit flip.  Returns the new move_last.  Both branches are restated: the undo
    fast path (:231-240) swaps the two arrays over the unique neighbours and negates both entries of the moved spin; the normal
    path (:243-259) walks ALL 2D entries of A[move], so a doubled bond (L = 2) is applied twice."""
    if move_last == move:
        for y in uA[move - 1]:
            lfields[y - 1], lfields_last[y - 1] = lfields_last[y - 1], lfields[y - 1]
        lfields[move - 1] = -lfields[move - 1]
        lfields_last[move - 1] = -lfields_last[move - 1]
        return move_last
    for y in uA[move - 1]:
        lfields_last[y - 1] = lfields[y - 1]
    sx = s[move - 1]
    for y, Jxy in zip(A[move - 1], J[move - 1]):
        sxy = 1 - 2 * (sx ^ s[y - 1])
        lfields[y - 1] = lfields[y - 1] - 4 * sxy * Jxy
    lfm = lfields[move - 1]
    lfields_last[move - 1] = lfm
    lfields[move - 1] = -lfm
    return move


def replay_standard_mc_ea(tape):
    """standardMC (RRRMC.jl:81-127) on GraphEA{Int,(-1,1),2D}(A, J) with the draws of the tape."""
    N, K = int(tape["N"]), int(tape["K"])
    beta, iters, step = float(tape["beta"]), int(tape["iters"]), int(tape["step"])
    A = [[int(v) for v in tape["A"][x * K:(x + 1) * K]] for x in range(N)]
    J = [[int(v) for v in tape["J"][x * K:(x + 1) * K]] for x in range(N)]
    uA = ea_unique_neighbours(A)
    s = bits_of_chunks([int(c, 16) for c in tape["C0"]], N)
    sites = [int(v) for v in tape["sites"]]
    us = [float(v) for v in tape["uniforms"]]
    E, lfields, lfields_last, move_last = ea_energy(A, J, s)
    Es, accepted, flips, undos, min_margin = [], 0, [], 0, float("inf")
    for it in range(1, iters + 1):
        if it % step == 0:
            Es.append(E)
        i = sites[it - 1]
        dE = -lfields[i - 1]                               # delta_energy, EA.jl:266-275
        x = -beta * dE
        ok = x >= 0
        if not ok:
            p = math.exp(x)
            ok = us[it - 1] < p
            min_margin = min(min_margin, abs(us[it - 1] - p) / p)
        flips.append(1 if ok else 0)
        if not ok:
            continue
        s[i - 1] ^= 1                                      # spinflip!, Interface.jl:89-92
        undos += move_last == i
        move_last = ea_update_cache(A, uA, J, s, lfields, lfields_last, move_last, i)
        E += dE
        accepted += 1
    E1, lf1 = rrg_energy(A, J, s)
    assert E == E1 and lfields == lf1                      # tracked E == energy(X, C) (test/runtests.jl:12-20), cache == recomputation
    return {"Es": Es, "chunks": chunks_of_bits(s), "accepted": accepted, "flips": flips, "undos": undos, "min_margin": min_margin}


# ---- GraphRRGNormal: src/graphs/RRG.jl:503-627 (SimpleGraph{Float64}: Gaussian couplings, Float64 local fields) --------------------------
def rrgn_energy(A, J, s):
    """energy (RRG.jl:531-558): lf -= Jxy * sx * sy per neighbour in the order of A[x]; E1 += lf; lfields[x] = 2 lf; E1 /= 2; the undo
    state is reset (move_last = 0, lfields_last = 0)."""
    E1, lfields = 0.0, [0.0] * len(A)
    for x in range(len(A)):
        sx = 2 * s[x] - 1
        lf = 0.0
        for y, Jxy in zip(A[x], J[x]):
            lf -= Jxy * sx * (2 * s[y - 1] - 1)
        E1 += lf
        lfields[x] = 2 * lf
    return E1 / 2, lfields, [0.0] * len(A), 0


def rrgn_update_cache(A, J, s, lfields, lfields_last, move_last, move):
    """update_cache! (RRG.jl:560-600), called AFTER the bit flip; returns the new move_last.  Undo branch (:566-577): the neighbours' entries of
    lfields and lfields_last are swapped, both entries of the moved spin negated, move_last kept.  Normal branch (:579-596): every
    neighbour's field is saved and changed by -4 sxy Jxy (sxy from the NEW spin of `move`), the moved spin's field saved and negated."""
    if move_last == move:
        for y in A[move - 1]:
            lfields[y - 1], lfields_last[y - 1] = lfields_last[y - 1], lfields[y - 1]
        lfields[move - 1] = -lfields[move - 1]
        lfields_last[move - 1] = -lfields_last[move - 1]
        return move_last
    sx = s[move - 1]
    for y, Jxy in zip(A[move - 1], J[move - 1]):
        sxy = 1 - 2 * (sx ^ s[y - 1])
        lfy = lfields[y - 1]
        lfields_last[y - 1] = lfy
        lfields[y - 1] = lfy - 4 * sxy * Jxy
    lfm = lfields[move - 1]
    lfields_last[move - 1] = lfm
    lfields[move - 1] = -lfm
    return move


def replay_standard_mc_rrgn(tape):
    """standardMC (RRRMC.jl:81-127) on GraphRRGNormal with the graph and the draws of the tape.  Every quantity is the same sequence of IEEE
    double operations the reference executes, so the results are compared bit for bit."""
    N, K = int(tape["N"]), int(tape["K"])
    beta, iters, step = float(tape["beta"]), int(tape["iters"]), int(tape["step"])
    A = [[int(v) for v in tape["A"][x * K:(x + 1) * K]] for x in range(N)]
    J = [[float(v) for v in tape["J"][x * K:(x + 1) * K]] for x in range(N)]
    s = bits_of_chunks([int(c, 16) for c in tape["C0"]], N)
    sites = [int(v) for v in tape["sites"]]
    us = [float(v) for v in tape["uniforms"]]
    E, lfields, lfields_last, move_last = rrgn_energy(A, J, s)
    Es, accepted, undos, min_margin = [], 0, 0, float("inf")
    for it in range(1, iters + 1):
        if it % step == 0:
            Es.append(E)
        i = sites[it - 1]
        dE = -lfields[i - 1]                               # delta_energy, RRG.jl:602-609
        x = -beta * dE
        ok = x >= 0                                        # accept, RRRMC.jl:39
        if not ok:
            p = math.exp(x)
            ok = us[it - 1] < p
            min_margin = min(min_margin, abs(us[it - 1] - p) / p)
        if not ok:
            continue
        s[i - 1] ^= 1                                      # spinflip!, Interface.jl:89-92
        undos += move_last == i
        move_last = rrgn_update_cache(A, J, s, lfields, lfields_last, move_last, i)
        E += dE
        accepted += 1
    E1, lf1, _, _ = rrgn_energy(A, J, s)
    assert abs(E - E1) <= 1e-9 * max(1.0, abs(E1))         # tracked E == energy(X, C) up to rounding (test/runtests.jl:12-20)
    return {"Es": Es, "chunks": chunks_of_bits(s), "accepted": accepted, "undos": undos, "min_margin": min_margin, "lfields": lfields}


class RRGNormal:
    """GraphRRGNormal as an object with the interface of SKNormal below (for the samplers that work through delta_energy / spinflip!):
    delta_energy(i) = -lfields[i] (RRG.jl:602-609), neighbors(i) = A[i] (:625)."""

    def __init__(self, A, J):
        self.N, self.A, self.J = len(A), A, J
        self.lfields, self.lfields_last, self.move_last, self.swaps = [0.0] * self.N, [0.0] * self.N, 0, 0

    def energy(self, s):
        E, self.lfields, self.lfields_last, self.move_last = rrgn_energy(self.A, self.J, s)
        return E

    def delta_energy(self, s, move):
        return -self.lfields[move - 1]

    def spinflip(self, s, move):                           # Interface.jl:89-92
        s[move - 1] ^= 1
        self.swaps += self.move_last == move
        self.move_last = rrgn_update_cache(self.A, self.J, s, self.lfields, self.lfields_last, self.move_last, move)

    def neighbors(self, move):
        return list(self.A[move - 1])


# ---- GraphSKNormal: src/graphs/SK.jl:170-297 ----------------------------------------------------------------------------------
class SKNormal:
    """J = N rows of Float64; cache = (lfields, lfields_last, move_last).  delta_energy(i) = +lfields[i] (SK.jl:278-284)."""

    def __init__(self, J):
        self.N, self.J = len(J), J
        self.lfields, self.lfields_last, self.move_last, self.swaps = [0.0] * self.N, [0.0] * self.N, 0, 0

    def energy(self, s):                                   # SK.jl:212-237
        n = 0.0
        for i in range(self.N):
            Ji, si, lf = self.J[i], s[i], 0.0
            for j in range(self.N):
                lf += (1 - 2 * (si ^ s[j])) * Ji[j]
            self.lfields[i] = 2 * lf
            n -= lf
        n /= 2
        self.move_last = 0
        self.lfields_last = [0.0] * self.N
        return n

    def update_cache(self, s, move):                       # SK.jl:239-276, after the flip
        if self.move_last == move:
            self.lfields, self.lfields_last = self.lfields_last, self.lfields      # the whole arrays swap; move_last stays
            self.swaps += 1
            return
        Ji, si = self.J[move - 1], s[move - 1]
        lfm = self.lfields[move - 1]
        for j in range(self.N):
            Jsij = (1 - 2 * (si ^ s[j])) * Ji[j]
            lfj = self.lfields[j]
            self.lfields_last[j] = lfj
            self.lfields[j] = lfj + 4 * Jsij
        self.lfields_last[move - 1] = lfm
        self.lfields[move - 1] = -lfm
        self.move_last = move

    def delta_energy(self, s, move):
        return self.lfields[move - 1]

    def spinflip(self, s, move):                           # Interface.jl:89-92
        s[move - 1] ^= 1
        self.update_cache(s, move)


class SKBinary:
    """GraphSK (SK.jl:28-165): J = N rows of bits (coupling = (2 J_ij - 1) / sqrt(N)), integer cache, delta_energy = lfields / sN."""

    def __init__(self, Jbits):
        self.N, self.J, self.sN = len(Jbits), Jbits, math.sqrt(len(Jbits))
        self.lfields, self.lfields_last, self.move_last, self.swaps = [0] * self.N, [0] * self.N, 0, 0

    def energy(self, s):                                   # SK.jl:62-96
        N = self.N
        n = -2 * sum(s)
        for i in range(N):
            sc = sum(a ^ b for a, b in zip(self.J[i], s))
            si = s[i]
            lf = -(2 * si - 1) * (N - 1 - 2 * sc)
            self.lfields[i] = 2 * (-lf + 2 * si)
            n += lf
        assert n % 2 == 0
        n //= 2
        self.move_last = 0
        self.lfields_last = [0] * N
        return n / self.sN

    def update_cache(self, s, move):                       # SK.jl:98-135
        if self.move_last == move:
            self.lfields, self.lfields_last = self.lfields_last, self.lfields
            self.swaps += 1
            return
        Ji, si = self.J[move - 1], s[move - 1]
        lfm = self.lfields[move - 1]
        for j in range(self.N):
            Jsij = si ^ s[j] ^ Ji[j]
            lfj = self.lfields[j]
            self.lfields_last[j] = lfj
            self.lfields[j] = lfj + 8 * Jsij - 4
        self.lfields_last[move - 1] = lfm
        self.lfields[move - 1] = -lfm
        self.move_last = move

    def delta_energy(self, s, move):                       # SK.jl:137-140
        return self.lfields[move - 1] / self.sN

    def spinflip(self, s, move):
        s[move - 1] ^= 1
        self.update_cache(s, move)


def _sk_graph(tape):
    N = int(tape["N"])
    if tape["kind"].endswith("skb"):
        rows = [int(c, 16) for c in tape["J_chunks"]]
        nch = (N + 63) // 64
        return SKBinary([bits_of_chunks(rows[i * nch:(i + 1) * nch], N) for i in range(N)])
    Jf = [float(v) for v in tape["J"]]
    return SKNormal([Jf[i * N:(i + 1) * N] for i in range(N)])


def replay_standard_mc_sk(tape):
    """standardMC (RRRMC.jl:81-127) on GraphSKNormal(J) / GraphSK(J) with the draws of the tape (uniform consulted only when dE > 0)."""
    N = int(tape["N"])
    beta, iters, step = float(tape["beta"]), int(tape["iters"]), int(tape["step"])
    X = _sk_graph(tape)
    s = bits_of_chunks([int(c, 16) for c in tape["C0"]], N)
    sites = [int(v) for v in tape["sites"]]
    us = [float(v) for v in tape["uniforms"]]
    E = X.energy(s)
    Es, accepted, flips, min_margin = [], 0, [], float("inf")
    for it in range(1, iters + 1):
        if it % step == 0:
            Es.append(E)
        i = sites[it - 1]
        dE = X.delta_energy(s, i)
        x = -beta * dE
        ok = x >= 0
        if not ok:
            p = math.exp(x)
            ok = us[it - 1] < p
            min_margin = min(min_margin, abs(us[it - 1] - p) / p)
        flips.append(1 if ok else 0)
        if not ok:
            continue
        X.spinflip(s, i)
        E += dE
        accepted += 1
    lf_live = list(X.lfields)
    E1 = X.energy(s)                                        # the reference's own test: abs(E - energy(X, C)) < 1e-11 scaled (test/runtests.jl:12-20)
    assert abs(E - E1) < 1e-9 and all(abs(a - b) < 1e-9 for a, b in zip(lf_live, X.lfields))
    return {"Es": Es, "chunks": chunks_of_bits(s), "accepted": accepted, "flips": flips, "swaps": X.swaps, "min_margin": min_margin,
            "lfields": lf_live}


# ---- DynamicSampler (src/DynamicSamplers.jl) + DeltaECacheCont (src/DeltaE.jl:297-410) + rrrMC(SingleGraph) (src/RRRMC.jl:149-219) ----
class DynamicSampler:
    def __init__(self, v):                                  # DynamicSamplers.jl:34-51
        self.N = len(v)
        self.levs = max(0, math.ceil(math.log2(self.N)))
        N2 = 2 ** self.levs
        self.v = list(v) + [0.0] * (N2 - self.N)
        self.ps = [0.0] * (N2 - 1)
        self.tinds, self.tpos = self.buildtable(self.levs)
        self.z, self.trefresh, self.refreshes, self.margin = 0.0, 0, 0, float("inf")
        self.refresh()
        self.refreshes = 0

    @staticmethod
    def buildtable(levs):                                   # :54-82 (1-based tables kept 1-based: index 0 unused)
        N2 = 2 ** levs
        tinds, tpos = [0] * (N2 + 2), [0]
        j0 = 1
        for i in range(1, N2 + 1):
            tinds[i] = j0
            j1, k, u, off = j0, 0, 1 << (levs - 1) if levs > 0 else 0, 1
            for _ in range(levs):
                if (i - 1) & u == 0:
                    tpos.append(off + k)
                    j1 += 1
                    k *= 2
                else:
                    k = 2 * k + 1
                u >>= 1
                off *= 2
            j0 = j1
        tinds[N2 + 1] = j0
        return tinds, tpos

    def refresh(self):                                      # :84-98
        z = 0.0
        for x in self.v:                                    # sum(v) (Julia sums pairwise: last-bit differences, inside the tapes' margins)
            z += x
        self.z = z
        self.ps = [0.0] * len(self.ps)
        for i in range(1, self.N + 1):
            for j in range(self.tinds[i], self.tinds[i + 1]):
                self.ps[self.tpos[j] - 1] += self.v[i - 1]
        self.trefresh = 0
        self.refreshes += 1

    def getel(self, x):                                     # :130-152 (the precision-loss branch must not be reached by a tape)
        x *= self.z
        k, off = 0, 1
        for _ in range(self.levs):
            p = self.ps[off + k - 1]
            self.margin = min(self.margin, abs(x - p) / self.z)          # how close the walk came to taking the other branch
            k *= 2
            if x > p:
                x -= p
                k += 1
            off *= 2
        assert not (k >= self.N or self.v[k] == 0), "precision-loss branch of getel"
        return k + 1

    def set(self, i, x):                                    # setindex!, :159-176
        if self.trefresh >= max(self.N, 100):
            self.refresh()
        self.trefresh += 1
        d = x - self.v[i - 1]
        self.v[i - 1] = x
        self.z += d
        for j in range(self.tinds[i], self.tinds[i + 1]):
            self.ps[self.tpos[j] - 1] += d


def replay_rrr_single_sk(tape, exp=math.exp):
    """rrrMC(X::SingleGraph) (RRRMC.jl:149-219) on GraphSKNormal(J) with DeltaECacheCont (DeltaE.jl:299-410): rand(dynsmp) = getel(dynsmp, u)
    with u from the tape, `rand() < c` with the tape's acceptance uniform (drawn at every iteration, :192,202)."""
    N = int(tape["N"])
    beta, iters, step = float(tape["beta"]), int(tape["iters"]), int(tape["step"])
    staged_thr, staged_thr_fact = float(tape["staged_thr"]), float(tape["staged_thr_fact"])
    if tape["kind"] == "rrrMC_rrgn":                        # the same sampler over GraphRRGNormal (round 4): sparse neighbourhoods
        K = int(tape["K"])
        X = RRGNormal([[int(v) for v in tape["A"][x * K:(x + 1) * K]] for x in range(N)],
                      [[float(v) for v in tape["J"][x * K:(x + 1) * K]] for x in range(N)])
    else:
        X = _sk_graph(tape)
    s = bits_of_chunks([int(c, 16) for c in tape["C0"]], N)
    u_move = [float(v) for v in tape["u_move"]]
    u_acc = [float(v) for v in tape["u_accept"]]
    prior = lambda x: exp(-x) if x > 0 else 1.0             # DeltaE.jl:297
    E = X.energy(s)
    dEs = [X.delta_energy(s, i) for i in range(1, N + 1)]   # DeltaECacheCont, DeltaE.jl:304-313
    ds = DynamicSampler([prior(beta * d) for d in dEs])
    neighbors = X.neighbors if hasattr(X, "neighbors") else (lambda i: [j for j in range(1, N + 1) if j != i])     # AllButOne(N, i), SK.jl:297
    lam = staged_thr_fact / N
    Es, accepted, staged_its, acc_rate, margin = [], 0, 0, 0.5, float("inf")

    def apply_move(move):                                   # DeltaE.jl:379-410 (inner graph of a SingleGraph = itself)
        X.spinflip(s, move)
        z = ds.z
        d = X.delta_energy(s, move)
        dEs[move - 1] = d
        ds.set(move, prior(beta * d))
        for j in neighbors(move):
            d = X.delta_energy(s, j)
            dEs[j - 1] = d
            ds.set(j, prior(beta * d))
        return z / ds.z

    for it in range(1, iters + 1):
        if it % step == 0:
            Es.append(E)
        acc = False
        if acc_rate < staged_thr:
            staged_its += 1
            z = ds.z                                        # step_rrr, RRRMC.jl:131-138
            move = ds.getel(u_move[it - 1])
            dE = dEs[move - 1]
            X.spinflip(s, move)                             # compute_staged!, DeltaE.jl:357-374
            staged = []
            d = X.delta_energy(s, move)
            staged.append((move, d, prior(beta * d)))
            for j in neighbors(move):
                d = X.delta_energy(s, j)
                staged.append((j, d, prior(beta * d)))
            X.spinflip(s, move)
            zp = ds.z                                       # compute_reverse_probabilities!, :345-355
            for j, _, p in staged:
                zp += p - ds.v[j - 1]
            zp = min(max(zp, 2.2250738585072014e-308), float(N))
            c = z / zp
            margin = min(margin, abs(u_acc[it - 1] - c) / c)
            if u_acc[it - 1] < c:
                X.spinflip(s, move)
                for j, d, p in staged:                      # apply_staged!, :335-343
                    dEs[j - 1] = d
                    ds.set(j, p)
                E += dE
                accepted += 1
                acc = True
        else:
            move = ds.getel(u_move[it - 1])
            dE = dEs[move - 1]
            c = apply_move(move)
            margin = min(margin, abs(u_acc[it - 1] - c) / c)
            if u_acc[it - 1] < c:
                E += dE
                accepted += 1
                acc = True
            else:
                apply_move(move)
        acc_rate = acc_rate * (1 - lam) + (1.0 if acc else 0.0) * lam
    return {"Es": Es, "chunks": chunks_of_bits(s), "accepted": accepted, "staged_its": staged_its, "min_margin": min(margin, ds.margin),
            "dEs": dEs, "z": ds.z, "refreshes": ds.refreshes, "swaps": X.swaps}


# ---- rrrMC(X::SingleGraph) and bklMC on GraphRRG{Int,(-1,1),K}: src/RRRMC.jl:131-219, 311-359; src/DeltaE.jl:62-295 (round 3) -------
def replay_rrr_bkl_rrg(tape, exp=math.exp, log1p=math.log1p):
    """kind rrrMC_rrg: rrrMC(X::SingleGraph) with DeltaECache{Int,L} — rand_move's rand() and rand(1:t) (DeltaE.jl:148,164; ArraySets.jl:83)
    and the `rand() < c` of RRRMC.jl:192,202 taken from the tape.  kind bklMC_rrg: bklMC (RRRMC.jl:311-359) — rand_skip's rand()
    (DeltaE.jl:141-144) and rand_move's two draws per MOVE from the tape."""
    bkl = tape["kind"] == "bklMC_rrg"
    N, K = int(tape["N"]), int(tape["K"])
    beta, iters, step = float(tape["beta"]), int(tape["iters"]), int(tape["step"])
    A = [[int(v) for v in tape["A"][x * K:(x + 1) * K]] for x in range(N)]
    J = [[int(v) for v in tape["J"][x * K:(x + 1) * K]] for x in range(N)]
    uA = [[y for y, Jxy in zip(a, j) if Jxy != 0] for a, j in zip(A, J)]      # neighbors(X, i) = X.uA[i] (RRG.jl:133,261): the neighbours with a non-zero coupling
    s = bits_of_chunks([int(c, 16) for c in tape["C0"]], N)
    u_cls = [float(v) for v in tape["u_class"]]
    u_mem = [int(v) for v in tape["u_member"]]
    E, lfields = rrg_energy(A, J, s)
    # allDeltaE(GraphRRG{Int,(-1,1),K}) (RRG.jl:262-264)
    dElist = tuple(4 * (d - 1) for d in range(1, K // 2 + 2)) if K % 2 == 0 else tuple(2 * (2 * d - 1) for d in range(1, (K + 1) // 2 + 1))
    L = len(dElist)
    delta = lambda i: -lfields[i - 1]                          # delta_energy, RRG.jl:236-244

    def cls(i):                                                # DeltaE.jl:79-84 / :246-250
        dE = delta(i)
        return dElist.index(abs(dE)) + 1 + L * (1 if (dE > 0 or (dE == 0 and s[i - 1] == 1)) else 0)

    sets, pos = [ArraySet(N) for _ in range(2 * L)], [0] * (N + 1)
    for i in range(1, N + 1):
        pos[i] = cls(i)
        sets[pos[i] - 1].push(i)
    ft = [exp(-beta * d) for d in dElist]
    fcls = lambda k: ft[k - L - 1] if k > L else 1.0           # get_class_f, DeltaE.jl:138-139
    T = [sets[k - 1].t * fcls(k) for k in range(1, 2 * L + 1)]
    z = 0.0
    for x in T:
        z += x
    margin = float("inf")

    def rand_move(g):                                          # DeltaE.jl:146-167
        nonlocal margin
        r = u_cls[g - 1] * z
        k, cT = 0, 0.0
        for k in range(1, 2 * L + 1):
            cT += T[k - 1]
            margin = min(margin, abs(r - cT) / z)
            if r < cT:
                break
        if not (r < cT):
            while T[k - 1] == 0:
                k -= 1
        dE = -dElist[k - 1] if k <= L else dElist[k - L - 1]
        t = sets[k - 1].t
        return sets[k - 1].v[((u_mem[g - 1] * t) >> 64) + 1], dE

    def flip(i):
        rrg_spinflip(A, J, s, lfields, i)

    def staged_of(move):                                       # compute_staged! (DeltaE.jl:198-230)
        flip(move)
        st = []
        for j in uA[move - 1]:
            k0, k1 = pos[j], cls(j)
            if k0 != k1:
                st.append((j, k0, k1))
        k0 = pos[move]
        st.append((move, k0, k0 - L * (2 * (1 if k0 > L else 0) - 1)))
        flip(move)
        return st

    def apply_move(move):                                      # DeltaE.jl:232-295
        nonlocal z
        flip(move)
        zp = z
        for j in uA[move - 1]:
            k0, k1 = pos[j], cls(j)
            if k0 == k1:
                continue
            f0, f1 = fcls(k0), fcls(k1)
            T[k0 - 1] -= f0
            T[k1 - 1] += f1
            zp += f1 - f0
            sets[k0 - 1].delete(j)
            sets[k1 - 1].push(j)
            pos[j] = k1
        k0 = pos[move]
        k1 = k0 - L * (2 * (1 if k0 > L else 0) - 1)
        f0, f1 = fcls(k0), fcls(k1)
        T[k0 - 1] -= f0
        T[k1 - 1] += f1
        zp += f1 - f0
        sets[k0 - 1].delete(move)
        sets[k1 - 1].push(move)
        pos[move] = k1
        c = z / zp
        z = zp
        return c

    Es, accepted, staged_its = [], 0, 0
    if not bkl:
        u_acc = [float(v) for v in tape["u_accept"]]
        staged_thr, lam, acc_rate = float(tape["staged_thr"]), float(tape["staged_thr_fact"]) / N, 0.5
        for it in range(1, iters + 1):
            if it % step == 0:
                Es.append(E)
            acc = False
            if acc_rate < staged_thr:
                staged_its += 1
                z0 = z                                         # step_rrr, RRRMC.jl:131-138
                move, dE = rand_move(it)
                st = staged_of(move)
                Tp, zp = list(T), z                            # compute_reverse_probabilities!, DeltaE.jl:184-196
                for _, k0, k1 in st:
                    f0, f1 = fcls(k0), fcls(k1)
                    Tp[k0 - 1] -= f0
                    Tp[k1 - 1] += f1
                    zp += f1 - f0
                c = z0 / zp
                margin = min(margin, abs(u_acc[it - 1] - c) / c)
                if u_acc[it - 1] < c:
                    flip(move)
                    for j, k0, k1 in st:                       # apply_staged!, DeltaE.jl:169-182
                        sets[k0 - 1].delete(j)
                        sets[k1 - 1].push(j)
                        pos[j] = k1
                    T[:] = Tp
                    z = zp
                    E += dE
                    accepted += 1
                    acc = True
            else:
                move, dE = rand_move(it)
                c = apply_move(move)
                margin = min(margin, abs(u_acc[it - 1] - c) / c)
                if u_acc[it - 1] < c:
                    E += dE
                    accepted += 1
                    acc = True
                else:
                    apply_move(move)
            acc_rate = acc_rate * (1 - lam) + (1.0 if acc else 0.0) * lam
        its_done = iters
    else:
        u_skip = [float(v) for v in tape["u_skip"]]
        it, nextstep, m, out = 0, step, 0, False
        while it < iters and not out:
            m += 1
            q = log1p(-u_skip[m - 1]) / log1p(-z / N)          # rand_skip, DeltaE.jl:141-144
            skip = int(math.floor(q))
            margin = min(margin, min(q - skip, skip + 1 - q) / max(q, 1.0))
            move, dE = rand_move(m)
            while it + skip + 1 >= nextstep:
                Es.append(E)
                nextstep += step
                if nextstep > iters:
                    out = True
                    break
            if out:
                break
            apply_move(move)                                   # apply_step_bkl!, RRRMC.jl:297-298
            it += skip + 1
            E += dE
            accepted += 1
        staged_its, its_done = accepted, it
    assert E == rrg_energy(A, J, s)[0]
    assert all(pos[i] == cls(i) for i in range(1, N + 1))      # check_consistency's class test (DeltaE.jl:120-136)
    return {"Es": Es, "chunks": chunks_of_bits(s), "accepted": accepted, "staged_its": staged_its, "iters_done": its_done,
            "min_margin": margin, "sizes": [st.t for st in sets], "pos": pos[1:]}


# ---- wtmMC on GraphRRG{Int,(-1,1),K}: src/RRRMC.jl:376-426, src/WaitingTimes.jl (round 3) ---------------------------------------
def replay_wtm_rrg(tape, exp=math.exp, log1p=math.log1p):
    """wtmMC with every rand() of gen_wt (WaitingTimes.jl:18-22) taken from the tape, in call order: the N initial times of THeap(X, C, beta)
    (:26-36), then per move the moved spin and its neighbours (update_heap!, :40-52).  The heap is DataStructures' MutableBinaryMinHeap:
    only "top = smallest time" matters while the times are distinct, which the margin below asserts."""
    N, K = int(tape["N"]), int(tape["K"])
    beta, samples, step = float(tape["beta"]), int(tape["samples"]), float(tape["step"])
    A = [[int(v) for v in tape["A"][x * K:(x + 1) * K]] for x in range(N)]
    J = [[int(v) for v in tape["J"][x * K:(x + 1) * K]] for x in range(N)]
    nbrs = [[y for y, Jxy in zip(a, j) if Jxy != 0] for a, j in zip(A, J)]          # neighbors(X, i) = X.uA[i] (RRG.jl:133,261)
    s = bits_of_chunks([int(c, 16) for c in tape["C0"]], N)
    us = [float(v) for v in tape["uniforms"]]
    E, lfields = rrg_energy(A, J, s)
    nd = 0

    def gen_wt(tau):
        nonlocal nd
        u = us[nd]
        nd += 1
        return -tau * log1p(-u)

    tau_of = lambda dE: max(1.0, exp(beta * dE))               # WaitingTimes.jl:16
    tm = [0.0] * (N + 1)
    for i in range(1, N + 1):
        tm[i] = gen_wt(tau_of(-lfields[i - 1]))
    step /= N
    tmax = step * samples
    t, nextstep, num_moves, Es, margin, out = 0.0, step, 0, [], float("inf"), False
    while t < tmax and not out:
        order = sorted(range(1, N + 1), key=lambda i: tm[i])
        move, tp = order[0], tm[order[0]]
        margin = min(margin, (tm[order[1]] - tp) / max(tp, 1e-300))            # pick_next: the top of the heap is unambiguous
        while tp >= nextstep:
            Es.append(E)
            nextstep += step
            if nextstep > tmax + 1e-10:
                out = True
                break
        margin = min(margin, abs(tp - nextstep) / max(tp, 1e-300), abs(tp - (nextstep - step)) / max(tp, 1e-300))
        if out:
            break
        t = tp
        dE = -lfields[move - 1]                                # update_heap!, WaitingTimes.jl:40-52
        rrg_spinflip(A, J, s, lfields, move)
        tm[move] = t + gen_wt(tau_of(-dE))
        for j in nbrs[move - 1]:
            tm[j] = t + gen_wt(tau_of(-lfields[j - 1]))
        E += dE
        num_moves += 1
    assert E == rrg_energy(A, J, s)[0]
    return {"Es": Es, "chunks": chunks_of_bits(s), "num_moves": num_moves, "t": t, "draws": nd, "min_margin": margin}


# ---- extremal_opt on GraphRRG{Int,(-1,1),K} with EOCache{Int,L}: src/RRRMC.jl:474-521, src/DeltaE.jl:412-555 (round 3) -----------
def replay_eo_rrg(tape):
    """tau-EO with rand_move's two draws from the tape: u_rank = rand() of r = (1 - rand()) z (DeltaE.jl:487), u_member -> rand(1:t)
    (ArraySets.jl:83).  ftau = cumsum(j^-tau) is recomputed here (sequential sum, libm pow)."""
    N, K = int(tape["N"]), int(tape["K"])
    tau, iters, step = float(tape["tau"]), int(tape["iters"]), int(tape["step"])
    A = [[int(v) for v in tape["A"][x * K:(x + 1) * K]] for x in range(N)]
    J = [[int(v) for v in tape["J"][x * K:(x + 1) * K]] for x in range(N)]
    nbrs = [[y for y, Jxy in zip(a, j) if Jxy != 0] for a, j in zip(A, J)]          # neighbors(X, i) = X.uA[i] (RRG.jl:133,261)
    s = bits_of_chunks([int(c, 16) for c in tape["C0"]], N)
    u_rank = [float(v) for v in tape["u_rank"]]
    u_mem = [int(v) for v in tape["u_member"]]
    E, lfields = rrg_energy(A, J, s)
    dElist = tuple(4 * (d - 1) for d in range(1, K // 2 + 2)) if K % 2 == 0 else tuple(2 * (2 * d - 1) for d in range(1, (K + 1) // 2 + 1))
    L, has_zero = len(dElist), 1 if 0 in dElist else 0
    KK = 2 * L - has_zero

    def findks(dE):                                            # DeltaE.jl:412-421
        ak = dElist.index(abs(dE)) + 1
        return ak + L - has_zero if dE >= 0 else L + 1 - ak

    sets, pos = [ArraySet(N) for _ in range(KK)], [0] * (N + 1)
    for i in range(1, N + 1):
        pos[i] = findks(-lfields[i - 1])
        sets[pos[i] - 1].push(i)
    ftau, c = [], 0.0
    for j in range(1, N + 1):                                  # cumsum([j^(-tau) for j = 1:N]), DeltaE.jl:444
        c += j ** (-tau)
        ftau.append(c)
    z = ftau[-1]
    Emin, Cmin, itmin, Es, margin = E, list(s), 0, [], float("inf")
    for it in range(1, iters + 1):
        if it % step == 0:
            Es.append(E)                                       # the hook's E (RRRMC.jl:499-502)
        r = (1 - u_rank[it - 1]) * z                           # rand_move, DeltaE.jl:484-516
        i = next(q for q in range(1, N + 1) if ftau[q - 1] >= r)          # searchsortedfirst(ftau, r)
        margin = min(margin, abs(ftau[i - 1] - r) / z, abs(r - (ftau[i - 2] if i > 1 else 0.0)) / z)
        k, t = 0, 0
        while i > t:
            k += 1
            t += sets[k - 1].t
        dE = -dElist[L - k] if k <= L else dElist[k - L + has_zero - 1]
        tk = sets[k - 1].t
        move = sets[k - 1].v[((u_mem[it - 1] * tk) >> 64) + 1]
        rrg_spinflip(A, J, s, lfields, move)                   # apply_move!, DeltaE.jl:518-552
        for j in nbrs[move - 1]:
            k0, k1 = pos[j], findks(-lfields[j - 1])
            if k0 != k1:
                sets[k0 - 1].delete(j)
                sets[k1 - 1].push(j)
                pos[j] = k1
        k0, k1 = pos[move], findks(-lfields[move - 1])
        if k0 != k1:
            sets[k0 - 1].delete(move)
            sets[k1 - 1].push(move)
            pos[move] = k1
        E += dE
        if E < Emin:
            Emin, Cmin, itmin = E, list(s), it
    assert E == rrg_energy(A, J, s)[0]
    return {"Es": Es, "chunks": chunks_of_bits(s), "Emin": Emin, "Cmin": chunks_of_bits(Cmin), "itmin": itmin, "min_margin": margin,
            "sizes": [st.t for st in sets], "pos": pos[1:]}
