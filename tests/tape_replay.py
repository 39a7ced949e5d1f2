"""RNG-free replays of the reference's samplers in plain Python — TEST INFRASTRUCTURE, independent of oracle/rrrmc_oracle.c.

A *tape* (tests/golden/tape_*.txt, written by tests/golden/make_tapes.py) holds a graph, a start configuration and every random
draw of a run, pre-drawn: sites and acceptance uniforms for standardMC, (class uniform, member word, acceptance uniform) for
rrrMC.  With the draws fixed the reference's loop is a deterministic function of its own graph code, so the same tape can be
replayed (a) here, statement by statement after the Julia sources cited below, (b) by julia/replay_tape.jl through the reference's
OWN functions wherever Julia and RRRMC.jl exist, (c) by the HIP library (tests/test_tapes.py).  The functions mirror
julia/replay_tape.jl line by line; indices are 1-based as in the tape."""
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
    J = [[int(v) for v in tape["J"][x * K:(x + 1) * K]] for x in range(Nk)]
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
        Ek, lf = rrg_energy(A, J, sl)
        E += Ek / M
        slices.append([sl, lf])
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
