"""Graph and configuration objects: the data the drop-in boundary marshals.

Indices are 0-based here; the Julia glue converts from the reference's 1-based tuples.
"""
import math

import numpy as np

from ._lib import check, lib

DEFAULT_SEED = 167432777111  # src/RRRMC.jl:82


def nchunks(N):
    return (int(N) + 63) // 64


class Config:
    """R configurations of N spins in Julia BitVector chunk layout (src/Interface.jl:21-29).

    ``s[r, c]`` is chunk c of replica r; spin x of replica r is bit (x & 63) of ``s[r, x >> 6]``,
    sigma = 2*bit - 1.  ``Config(N, R)`` alone is all-zeros; random initialisation is done on the
    device by ``Engine.init_spins_random`` (the reference draws rand!(BitVector), Interface.jl:26).
    """

    def __init__(self, N, R=1, s=None):
        self.N = int(N)
        self.R = int(R)
        if s is None:
            s = np.zeros((self.R, nchunks(N)), np.uint64)
        s = np.ascontiguousarray(s, np.uint64).reshape(self.R, nchunks(N))
        self.s = s

    def __len__(self):
        return self.N

    def copy(self):
        return Config(self.N, self.R, self.s.copy())

    def __eq__(self, other):
        return isinstance(other, Config) and self.N == other.N and self.R == other.R and bool((self.s == other.s).all())

    def bits(self):
        """[R, N] array of 0/1."""
        x = np.arange(self.N)
        return ((self.s[:, x >> 6] >> (x & 63).astype(np.uint64)) & np.uint64(1)).astype(np.uint8)

    @staticmethod
    def from_bits(bits):
        bits = np.atleast_2d(np.asarray(bits)).astype(np.uint64)
        R, N = bits.shape
        s = np.zeros((R, nchunks(N)), np.uint64)
        x = np.arange(N)
        np.bitwise_or.at(s, (np.arange(R)[:, None], (x >> 6)[None, :]), bits << (x & 63).astype(np.uint64))
        return Config(N, R, s)


def level_units(LEV):
    """LEV as given to the reference's constructors -> (units, mul, div): value = units * mul / div.
    Int levels: (LEV, 1, 1.0).  Float64 levels become DFloat64 (RRG.jl:162,324; EA.jl:193,357): the Int64 t = round(x * 10^5)
    (src/DFloats.jl:23, ties to even); units = t / g with g = gcd of the |t|, (mul, div) = (g, 1e5).  Rational levels
    (fractions.Fraction; runtests.jl:40) over the common denominator d: units = numerators / g, (mul, div) = (g, d)."""
    from fractions import Fraction
    LEV = tuple(LEV)
    if all(isinstance(l, (int, np.integer)) for l in LEV):
        return tuple(int(l) for l in LEV), 1, 1.0
    if all(isinstance(l, (int, np.integer, Fraction)) for l in LEV):
        d = 1
        for l in LEV:
            d = d * Fraction(l).denominator // math.gcd(d, Fraction(l).denominator)
        t = [int(Fraction(l) * d) for l in LEV]
        div = float(d)
    else:
        t = [int(np.rint(float(l) * 100000.0)) for l in LEV]
        div = 100000.0
    g = 0
    for x in t:
        g = math.gcd(g, abs(x))
    g = max(g, 1)
    return tuple(x // g for x in t), g, div


class _SparseLevelsGraph:
    """GraphRRG{ET,LEV,K} / GraphEA{ET,LEV,2D} with levels other than (-1, 1) (RRG.jl:116-162, EA.jl:138-193): couplings in level
    units, energies come back from the device in units and are returned as ``units * lev_mul / lev_div`` (exact integers for Int
    levels, the Float64 value of the reference's DFloat64 / Rational otherwise)."""
    model_kind = 7          # RRRMC_MODEL_SPARSE_LEVELS

    def _init_levels(self, A, J, LEV, ea_form):
        self.levels = tuple(LEV)
        self.LEV, self.lev_mul, self.lev_div = level_units(LEV)
        if len(set(self.LEV)) != len(self.LEV):
            raise ValueError("repeated levels in LEV: %r" % (LEV,))                   # RRG.jl:100
        if max(abs(u) for u in self.LEV) > 127:
            raise NotImplementedError("levels %r need more than 8 bits per coupling after reduction by their gcd" % (LEV,))
        A = np.ascontiguousarray(A, np.int32)
        self.N, self.K = A.shape
        self.A = A
        if J is None:
            J = np.zeros(A.shape, np.int8)
            check(lib().rrrmc_gen_couplings_lev(self.N, self.K, A, self._seed, np.asarray(self.LEV, np.int32), len(self.LEV), J))
        J = np.ascontiguousarray(J, np.int8)
        if J.shape != A.shape:
            raise ValueError("incompatible shapes of A and J: %r, %r" % (A.shape, J.shape))
        if not np.isin(J, self.LEV).all():
            raise ValueError("the given J is incompatible with levels %r" % (LEV,))  # RRG.jl:130
        self.J = J
        self.ea_form = ea_form
        self.energy_dtype = np.int64 if self.lev_div == 1.0 else np.float64

    def energy_value(self, units):
        """Level units -> the energy in the caller's terms (Int, or Float64(::DFloat64) = t / 10^5)."""
        u = np.asarray(units, np.int64) * self.lev_mul
        return u if self.lev_div == 1.0 else u / self.lev_div


class _SparsePM1Graph:
    """Common part of GraphRRG / GraphEA with LEV = (-1, 1): neighbour table A[N, K], couplings J[N, K]."""
    model_kind = 1          # RRRMC_MODEL_SPARSE_PM1
    energy_dtype = np.int64

    def __init__(self, A, J):
        A = np.ascontiguousarray(A, np.int32)
        J = np.ascontiguousarray(J, np.int8)
        if A.ndim != 2 or A.shape != J.shape:
            raise ValueError("incompatible shapes of A and J: %r, %r" % (A.shape, J.shape))
        if not np.isin(J, (-1, 1)).all():
            raise ValueError("the given J is incompatible with levels (-1, 1)")   # RRG.jl:130
        self.N, self.K = A.shape
        self.A, self.J = A, J


class GraphRRG(_SparsePM1Graph):
    """``GraphRRG(N, K)`` — random regular graph, +-1 couplings (src/graphs/RRG.jl:140-162).

    ``GraphRRG.from_AJ(A, J)`` is the inner constructor ``GraphRRG{Int,(-1,1),K}(A, J)`` (RRG.jl:122).
    The disorder comes from the GRAPH / COUPLING Philox streams of ``seed`` (the reference uses the
    global RNG, RRG.jl:45,155).
    """

    def __new__(cls, N=None, K=None, LEV=(-1, 1), seed=DEFAULT_SEED):
        if cls is GraphRRG and (not _is_pm1(LEV) or (K is not None and int(K) > PM1_MAX_K)):
            return object.__new__(GraphRRGLevels)        # GraphRRG{ET,LEV,K} with general levels: its own device context
        return object.__new__(cls)

    def __init__(self, N, K, LEV=(-1, 1), seed=DEFAULT_SEED):
        A = np.zeros((int(N), int(K)), np.int32)
        check(lib().rrrmc_gen_rrg(N, K, seed, A))
        J = np.zeros((int(N), int(K)), np.int8)
        check(lib().rrrmc_gen_couplings_pm1(N, K, A, seed, J))
        super().__init__(A, J)

    @classmethod
    def from_AJ(cls, A, J, LEV=(-1, 1)):
        """``GraphRRG{ET,LEV,K}(A, J)`` (RRG.jl:122); J in level units when LEV is not (-1, 1)."""
        if not _is_pm1(LEV) or np.shape(A)[1] > PM1_MAX_K:
            self = object.__new__(GraphRRGLevels)
            self._init_levels(A, J, LEV, 0)
            return self
        self = object.__new__(cls)
        _SparsePM1Graph.__init__(self, A, J)
        return self


# the bit-sliced +-J kernels count unsatisfied bonds in 3 bit planes (K <= 7); beyond that a +-J graph is a general-level graph with
# LEV = (-1, 1): same couplings (same COUPLING draws), same chains, integer energies, one thread per replica
PM1_MAX_K = 7


def _is_pm1(LEV):
    LEV = tuple(LEV)
    return LEV == (-1, 1) and all(isinstance(l, (int, np.integer)) for l in LEV)


class GraphRRGLevels(_SparseLevelsGraph, GraphRRG):
    """``GraphRRG(N, K, LEV)`` with LEV other than (-1, 1): Int, Float64 (-> DFloat64) or Fraction levels (test/runtests.jl:37-40)."""

    def __init__(self, N, K, LEV=(-1, 1), seed=DEFAULT_SEED):
        A = np.zeros((int(N), int(K)), np.int32)
        check(lib().rrrmc_gen_rrg(N, K, seed, A))
        self._seed = seed
        self._init_levels(A, None, LEV, 0)


class GraphEA(_SparsePM1Graph):
    """``GraphEA(L, D)`` — Edwards-Anderson lattice, +-1 couplings (src/graphs/EA.jl:171-193)."""

    def __new__(cls, L=None, D=None, LEV=(-1, 1), seed=DEFAULT_SEED):
        if cls is GraphEA and (not _is_pm1(LEV) or (D is not None and 2 * int(D) > PM1_MAX_K)):
            return object.__new__(GraphEALevels)
        return object.__new__(cls)

    def __init__(self, L, D, LEV=(-1, 1), seed=DEFAULT_SEED):
        N = int(L) ** int(D)
        A = np.zeros((N, 2 * int(D)), np.int32)
        check(lib().rrrmc_gen_ea(L, D, A))
        J = np.zeros((N, 2 * int(D)), np.int8)
        check(lib().rrrmc_gen_couplings_pm1(N, 2 * int(D), A, seed, J))
        super().__init__(A, J)
        self.L, self.D = int(L), int(D)

    @classmethod
    def from_AJ(cls, A, J, LEV=(-1, 1)):
        if not _is_pm1(LEV) or np.shape(A)[1] > PM1_MAX_K:
            self = object.__new__(GraphEALevels)
            self._init_levels(A, J, LEV, 1)
            return self
        self = object.__new__(cls)
        _SparsePM1Graph.__init__(self, A, J)
        return self


class GraphEALevels(_SparseLevelsGraph, GraphEA):
    """``GraphEA(L, D, LEV)`` with LEV other than (-1, 1) (test/runtests.jl:47-50, 57-60)."""

    def __init__(self, L, D, LEV=(-1, 1), seed=DEFAULT_SEED):
        N = int(L) ** int(D)
        A = np.zeros((N, 2 * int(D)), np.int32)
        check(lib().rrrmc_gen_ea(L, D, A))
        self._seed = seed
        self._init_levels(A, None, LEV, 1)
        self.L, self.D = int(L), int(D)


class _SparseF64Graph:
    """Common part of GraphRRGNormal / GraphEANormal: neighbour table A[N, K], Float64 couplings J[N, K]."""
    model_kind = 5          # RRRMC_MODEL_SPARSE_F64
    energy_dtype = np.float64

    def __init__(self, A, J):
        A = np.ascontiguousarray(A, np.int32)
        J = np.ascontiguousarray(J, np.float64)
        if A.ndim != 2 or A.shape != J.shape:
            raise ValueError("incompatible shapes of A and J: %r, %r" % (A.shape, J.shape))
        self.N, self.K = A.shape
        self.A, self.J = A, J

    @classmethod
    def from_AJ(cls, A, J):
        self = cls.__new__(cls)
        _SparseF64Graph.__init__(self, A, J)
        return self


class GraphRRGNormal(_SparseF64Graph):
    """``GraphRRGNormal(N, K)`` — random regular graph, couplings ~ Normal(0, 1) (src/graphs/RRG.jl:503-531).
    GRAPH stream for the pairing, one GAUSS-stream normal per bond where the reference calls ``randn()`` (RRG.jl:510-512)."""

    def __init__(self, N, K, seed=DEFAULT_SEED):
        A = np.zeros((int(N), int(K)), np.int32)
        check(lib().rrrmc_gen_rrg(N, K, seed, A))
        J = np.zeros((int(N), int(K)), np.float64)
        check(lib().rrrmc_gen_couplings_gauss(N, K, A, seed, J.reshape(-1)))
        super().__init__(A, J)


class GraphEANormal(_SparseF64Graph):
    """``GraphEANormal(L, D)`` — Edwards-Anderson lattice, couplings ~ Normal(0, 1) (src/graphs/EA.jl:534-574), or
    ``GraphEANormal(fname)`` from the text format of ``gen_AJ`` (EA.jl:73-118, D = 2):

        type: <anything>
        size: L
        name: <anything>
        x y Jxy          (one line per bond, 1-based sites)
    """

    def __init__(self, L, D=None, seed=DEFAULT_SEED):
        if isinstance(L, str):
            L, D, A, J = self._gen_AJ(L)
        else:
            if D < 1:
                raise ValueError("D must be >= 0, given: %d" % D)                  # EA.jl:566
            N = int(L) ** int(D)
            A = np.zeros((N, 2 * int(D)), np.int32)
            check(lib().rrrmc_gen_ea(L, D, A))
            J = np.zeros((N, 2 * int(D)), np.float64)
            check(lib().rrrmc_gen_couplings_gauss(N, 2 * int(D), A, seed, J.reshape(-1)))
        super().__init__(A, J)
        self.L, self.D = int(L), int(D)

    @staticmethod
    def _gen_AJ(fname):
        D = 2
        with open(fname) as f:
            if not f.readline().strip().startswith("type:"):
                raise ValueError("%s: expected a 'type:' line" % fname)
            ls = f.readline().split()
            if len(ls) != 2 or ls[0] != "size:":
                raise ValueError("%s: expected 'size: L'" % fname)
            L = int(ls[1])
            if not f.readline().strip().startswith("name:"):
                raise ValueError("%s: expected a 'name:' line" % fname)
            N = L ** D
            A = np.zeros((N, 2 * D), np.int32)
            check(lib().rrrmc_gen_ea(L, D, A))
            J = np.full((N, 2 * D), np.nan)
            for line in f:
                ls = line.split()
                if not ls:
                    continue
                if len(ls) != 3:
                    raise ValueError("%s: expected 'x y J', got %r" % (fname, line))
                x, y, Jxy = int(ls[0]) - 1, int(ls[1]) - 1, float(ls[2])
                for a, b in ((x, y), (y, x)):
                    ks = np.nonzero(A[a] == b)[0]
                    if ks.size == 0:
                        raise ValueError("%s: sites %d and %d are not neighbours" % (fname, x + 1, y + 1))
                    k = ks[0]                                                   # findfirst(Ax, y): EA.jl:99
                    if not np.isnan(J[a, k]):
                        raise ValueError("%s: bond (%d,%d) given twice" % (fname, x + 1, y + 1))
                    J[a, k] = Jxy
            if np.isnan(J).any():
                raise ValueError("%s: some bonds are missing" % fname)
        return L, D, A, J


class _DiscretizedGraph:
    """Common part of Graph{RRG,EA}NormalDiscretized: ``A``, the Gaussian couplings ``cJ`` and their split
    ``dJ`` (levels, the inner DiscrGraph ``X0``) + ``rJ`` (residuals) by ``discretize`` (src/Common.jl:38-72)."""
    model_kind = 6          # RRRMC_MODEL_SPARSE_DISCRETIZED
    energy_dtype = np.float64

    def _split(self, A, cJ, LEV):
        LEV = tuple(LEV)
        # Int levels: GraphRRGNormalDiscretized{Int,LEV,K}; Float64 levels -> DFloat64 (RRG.jl:324, EA.jl:357): see level_units
        self.LEV, self.lev_mul, self.lev_div = level_units(LEV)
        if max(abs(u) for u in self.LEV) > 127:
            raise NotImplementedError("levels %r need more than 8 bits per coupling after reduction by their gcd" % (LEV,))
        self.levels = LEV                                                             # as given by the caller
        if len(set(self.LEV)) != len(self.LEV):
            raise ValueError("repeated levels in LEV: %r" % (LEV,))                   # RRG.jl:100
        self.A = np.ascontiguousarray(A, np.int32)
        self.cJ = np.ascontiguousarray(cJ, np.float64)
        self.N, self.K = self.A.shape
        self.dJ = np.zeros(self.A.shape, np.int8)
        self.rJ = np.zeros(self.A.shape, np.float64)
        check(lib().rrrmc_discretize_scaled(self.cJ.reshape(-1), self.cJ.size, np.asarray(self.LEV, np.int32), len(self.LEV),
                                            self.lev_mul, self.lev_div, self.dJ.reshape(-1), self.rJ.reshape(-1)))


class GraphRRGNormalDiscretized(_DiscretizedGraph):
    """``GraphRRGNormalDiscretized(N, K, LEV)`` — DoubleGraph{DiscrGraph,Float64} (src/graphs/RRG.jl:285-324)."""
    ea_form = 0

    def __init__(self, N, K, LEV, seed=DEFAULT_SEED):
        A = np.zeros((int(N), int(K)), np.int32)
        check(lib().rrrmc_gen_rrg(N, K, seed, A))
        cJ = np.zeros((int(N), int(K)), np.float64)
        check(lib().rrrmc_gen_couplings_gauss(N, K, A, seed, cJ.reshape(-1)))
        self._split(A, cJ, LEV)


class GraphEANormalDiscretized(_DiscretizedGraph):
    """``GraphEANormalDiscretized(L, D, LEV)`` — DoubleGraph{DiscrGraph,Float64} (src/graphs/EA.jl:311-357)."""
    ea_form = 1

    def __init__(self, L, D, LEV, seed=DEFAULT_SEED):
        N = int(L) ** int(D)
        A = np.zeros((N, 2 * int(D)), np.int32)
        check(lib().rrrmc_gen_ea(L, D, A))
        cJ = np.zeros((N, 2 * int(D)), np.float64)
        check(lib().rrrmc_gen_couplings_gauss(N, 2 * int(D), A, seed, cJ.reshape(-1)))
        self._split(A, cJ, LEV)
        self.L, self.D = int(L), int(D)


class GraphSKNormal:
    """``GraphSKNormal(N)`` — Sherrington-Kirkpatrick model, couplings ~ Normal(0, 1/N) (src/graphs/SK.jl:181-210).

    ``GraphSKNormal.from_J(J)`` is ``GraphSKNormal(J; check=true)`` (SK.jl:184-197): J must be symmetric with a
    zero diagonal.  ``ET = Float64``: energies are float64.
    """
    model_kind = 2          # RRRMC_MODEL_SK_NORMAL
    energy_dtype = np.float64
    K = 0

    def __init__(self, N, seed=DEFAULT_SEED):
        J = np.zeros((int(N), int(N)), np.float64)
        check(lib().rrrmc_gen_sk_gauss(N, seed, J.reshape(-1)))
        self.N, self.J = int(N), J

    @classmethod
    def from_J(cls, J):
        J = np.ascontiguousarray(J, np.float64)
        if J.ndim != 2 or J.shape[0] != J.shape[1]:
            raise ValueError("invalid J inner length, expected %d" % J.shape[0])
        if (np.diag(J) != 0).any():
            raise ValueError("diagonal entries of J must be 0")
        if not (J == J.T).all():
            raise ValueError("J must be symmetric")
        self = cls.__new__(cls)
        self.N, self.J = J.shape[0], J
        return self


class GraphSK:
    """``GraphSK(N)`` — Sherrington-Kirkpatrick model with binary couplings J in {-1/sqrt(N), 1/sqrt(N)}
    (src/graphs/SK.jl:28-60).  ``J`` holds N BitVector rows ([N, ceil(N/64)] chunks, bit = 1 means +1/sqrt(N)).
    ``ET = Float64``; the cache is integer (lfields = sqrt(N) * delta_energy, SK.jl:137-140)."""
    model_kind = 4          # RRRMC_MODEL_SK_BINARY
    energy_dtype = np.float64
    K = 0

    def __init__(self, N, seed=DEFAULT_SEED):
        J = np.zeros((int(N), nchunks(N)), np.uint64)
        check(lib().rrrmc_gen_sk_binary(N, seed, J.reshape(-1)))
        self.N, self.J = int(N), J


class GraphQuant:
    """``GraphQuant(Nk, M, Γ, β, GraphRRG, Nk, K)`` — quantum Ising model in a transverse field Γ via the Suzuki-Trotter
    transformation: M coupled copies ("slices") of a classical graph (src/graphs/QT.jl:126-170).

    As the in-tree aliases do (src/QAliases.jl:43-67) the disorder is generated ONCE and shared by all slices:
    ``GraphQuant(X1, M, Γ, β)`` with ``X1`` a ``GraphRRG`` / ``GraphEA`` (±J), or a binary ``GraphSK`` — the reference's
    ``GraphQSKT(Nk, M, Γ, β)`` (src/QAliases.jl:34-43), the graph of ``scripts.jl:test_QIsing``.  ``N = Nk * M`` spins, slice-major.
    ``ET = Float64``.
    """
    model_kind = 3          # RRRMC_MODEL_QUANT_RRG
    energy_dtype = np.float64

    def __init__(self, X1, M, Gamma, beta):
        import math
        if M <= 2:
            raise ValueError("M must be greater than 2, given: %d" % M)          # QT.jl:47
        if Gamma < 0:
            raise ValueError("Γ must be >= 0")                                   # QT.jl:164
        self.X1, self.M, self.Gamma, self.beta = X1, int(M), float(Gamma), float(beta)
        self.sk_slices = isinstance(X1, GraphSK)
        self.skn_slices = isinstance(X1, GraphSKNormal)                    # GraphQSKNormalT (QAliases.jl:45-46; test/runtests.jl:80)
        self.f64_slices = isinstance(X1, _SparseF64Graph)                  # GraphQEAT = GraphQuant{fourK,GraphEANormal{twoD}} (QAliases.jl:50-83)
        dense = self.sk_slices or self.skn_slices
        self.Nk, self.K = X1.N, (0 if dense else X1.K)
        self.N = self.Nk * self.M
        self.A, self.J = (None if dense else X1.A), X1.J                   # GraphSK slices: J = the bit-packed rows (SK.jl:32); GraphSKNormal: N x N Float64
        # fourK = round(2/β * log(coth(β Γ / M)), digits = MAXDIGITS): QT.jl:165
        self.fourK = round(2.0 / beta * math.log(1.0 / math.tanh(beta * Gamma / M)), 8)


def GraphQSKNormalT(Nk, M, Gamma, beta, seed=DEFAULT_SEED):
    """``GraphQSKNormalT(Nk, M, Γ, β)`` = ``GraphQuant(Nk, M, Γ, β, GraphSKNormal, SK.gen_J_gauss(Nk))`` (src/QAliases.jl:45-46)."""
    return GraphQuant(GraphSKNormal(Nk, seed=seed), M, Gamma, beta)


def _gen_J_uniform(A, seed):
    """``EA.gen_J(Float64, N, A) do 4 * rand() - 2 end`` (src/QAliases.jl:60-62 over src/graphs/EA.jl:45-71): one draw per bond x < y in (x, k) order,
    mirrored into the first free slot of row y (so that two bonds to the same neighbour — L = 2 — get two draws).  The draws come from a
    Philox generator keyed by ``seed`` (Julia's stream is not reproducible outside Julia: SURVEY.md appendix B)."""
    N, K = A.shape
    rng = np.random.Generator(np.random.Philox(key=int(seed) & (2 ** 64 - 1)))
    J = np.full((N, K), np.nan)
    for x in range(N):
        for k in range(K):
            y = int(A[x, k])
            if x < y:
                Jxy = 4.0 * rng.random() - 2.0
                assert np.isnan(J[x, k])
                J[x, k] = Jxy
                free = np.nonzero(np.isnan(J[y]))[0]
                J[y, free[0]] = Jxy
    assert not np.isnan(J).any()
    return J


def GraphQEAT(L, D_or_M, M=None, Gamma=None, beta=None, seed=DEFAULT_SEED):
    """The reference's three constructors (src/QAliases.jl:50-83), all ``GraphQuant{fourK,GraphEANormal{2D}}`` over ONE shared ``(A, J)``:

    * ``GraphQEAT(L, D, M, Γ, β)`` — couplings uniform in [-2, 2) (``4 rand() - 2``, :60-62);
    * ``GraphQEAT(fname, M, Γ, β)`` — from the text format of ``gen_AJ`` (EA.jl:73-118);
    * ``GraphQEAT(X::GraphEANormal, M, Γ, β)``.
    """
    if isinstance(L, (str, GraphEANormal)):
        X1 = GraphEANormal(L) if isinstance(L, str) else L
        return GraphQuant(X1, D_or_M, M, Gamma)                 # (arguments shifted by one: fname | X, M, Γ, β)
    D = D_or_M
    if D < 1:
        raise ValueError("D must be ≥ 0, given: %d" % D)         # QAliases.jl:58 (the reference's message)
    N = int(L) ** int(D)
    A = np.zeros((N, 2 * int(D)), np.int32)
    check(lib().rrrmc_gen_ea(L, D, A))
    X1 = GraphEANormal.from_AJ(A, _gen_J_uniform(A, seed))
    X1.L, X1.D = int(L), int(D)
    return GraphQuant(X1, M, Gamma, beta)


def GraphQSKT(Nk, M, Gamma, beta, seed=DEFAULT_SEED):
    """``GraphQSKT(Nk, M, Γ, β)`` = ``GraphQuant(Nk, M, Γ, β, GraphSK, SK.gen_J(Nk))`` (src/QAliases.jl:34-43)."""
    return GraphQuant(GraphSK(Nk, seed=seed), M, Gamma, beta)


def checkerboard_coloring(L, D):
    """Two-colouring (parity of the coordinate sum) of the periodic L^D lattice of ``GraphEA``; L must be even."""
    if L % 2:
        raise ValueError("the periodic lattice is two-colourable only for even L, given: %d" % L)
    x = np.arange(int(L) ** int(D))
    par = np.zeros_like(x)
    for d in range(int(D)):
        par += (x // int(L) ** d) % int(L)
    return (par % 2).astype(np.int32)


def getN(X):
    """src/Interface.jl:145"""
    return X.N


def neighbors(X, i):
    """src/Interface.jl:158; RRG.jl:261 (uA = neighbours with non-zero coupling), EA.jl:292 (de-duplicated)."""
    if getattr(X, "model_kind", 0) == 7 and not X.ea_form:
        return X.A[i][X.J[i] != 0]
    return np.unique(X.A[i])


def all_delta_e(X):
    """allΔE (src/Interface.jl:200-201): RRG.jl:262-281, EA.jl:293-309 — sorted values of |dE|."""
    K = X.K
    if getattr(X, "model_kind", 0) == 7:
        es = {0}
        for _ in range(K):
            es = {e + s * l for e in es for l in X.LEV for s in (-1, 1)}
        return tuple(X.energy_value(sorted({2 * abs(e) for e in es})).tolist())
    return tuple(2 * m for m in range(K & 1, K + 1, 2))
