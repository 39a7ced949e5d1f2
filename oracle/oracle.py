"""ctypes front-end of the CPU oracle (oracle/rrrmc_oracle.c).

ORACLE — TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg, never by the product package.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librrrmc_oracle.so")

u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the oracle with gcc (no-op when the .so is newer than its sources)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


def build_native():
    """The -O3 -march=native build for bench.py's cpu_baseline leg.  Built where it is timed; falls back to the portable
    build when the host compiler refuses.  Select it with RRRMC_ORACLE_NATIVE=1 BEFORE the first call into this module."""
    import hashlib
    try:          # the file name carries the host CPU's identity: a build made on another machine (the tree travels) is never loaded here
        cpu = "".join(l for l in open("/proc/cpuinfo") if l.startswith(("model name", "flags")))[:20000]
    except OSError:
        cpu = "unknown"
    tag = hashlib.sha1(cpu.encode()).hexdigest()[:10]
    so = os.path.join(_HERE, "_build", "librrrmc_oracle_native_%s.so" % tag)
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    tmp = "%s.%d.tmp" % (so, os.getpid())          # built beside its final name and renamed: nobody ever loads a half-written file
    try:
        subprocess.check_call(["make", "-s", "-C", _HERE, "native", "NATIVE_SO=%s" % tmp])
        os.replace(tmp, so)
    except (subprocess.CalledProcessError, OSError):
        return build()
    return so


_lib = None
_lib_lock = threading.Lock()
flavour = "x86-64-v2"


def lib():
    global _lib, flavour
    if _lib is not None:
        return _lib
    with _lib_lock:          # (bench.py's verification calls in from several threads at once)
        return _lib_locked()


def _lib_locked():
    global _lib, flavour
    if _lib is None:
        so = build()
        if os.environ.get("RRRMC_ORACLE_LIB"):          # another build of the same sources (tests/test_sanitizers.py: the ASan/UBSan one)
            so, flavour = os.environ["RRRMC_ORACLE_LIB"], "override"
        elif os.environ.get("RRRMC_ORACLE_NATIVE") == "1":
            so = build_native()
            flavour = "native" if "_native_" in os.path.basename(so) else flavour
        L = C.CDLL(so)
        L.orc_philox.argtypes = [u32p, u32p, u32p]
        L.orc_set_hook.restype = None
        L.orc_set_hook.argtypes = [HOOK_FN, C.c_void_p]
        L.orc_site_of.restype = C.c_int64
        L.orc_site_of.argtypes = [C.c_uint64, C.c_uint64, C.c_int64]
        L.orc_accept_uniform.restype = C.c_uint64
        L.orc_accept_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_accept_less.restype = C.c_int
        L.orc_accept_less.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64]
        L.orc_threshold.restype = C.c_uint64
        L.orc_threshold.argtypes = [C.c_double, C.POINTER(C.c_int)]
        L.orc_init_config.argtypes = [C.c_uint64, C.c_uint32, C.c_int64, u64p]
        L.orc_gen_rrg.restype = C.c_int
        L.orc_gen_rrg.argtypes = [C.c_int64, C.c_int64, C.c_uint64, i32p]
        L.orc_gen_ea.restype = C.c_int
        L.orc_gen_ea.argtypes = [C.c_int64, C.c_int64, i32p]
        L.orc_gen_couplings.restype = C.c_int
        L.orc_gen_couplings.argtypes = [C.c_int64, C.c_int64, i32p, C.c_uint64, C.c_int64, i32p, i32p]
        L.orc_sparse_energy.restype = C.c_int64
        L.orc_sparse_energy.argtypes = [C.c_int64, C.c_int64, i32p, i32p, u64p, C.c_void_p]
        L.orc_standard_mc_sparse.restype = C.c_int64
        L.orc_standard_mc_sparse.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, i32p, C.c_double, C.c_int64, C.c_int64,
                                             C.c_uint64, C.c_uint64, C.c_uint32, u64p, i64p,
                                             C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_standard_mc_sparse_batch.restype = C.c_int64
        L.orc_standard_mc_sparse_batch.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, i32p, C.c_double, C.c_int64,
                                                   C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int64,
                                                   u64p, i64p, i64p]
        L.orc_gauss.restype = C.c_double
        L.orc_gauss.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_exp.restype = C.c_double
        L.orc_exp.argtypes = [C.c_double]
        L.orc_rand53_of.restype = C.c_double
        L.orc_rand53_of.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_gen_sk_gauss.argtypes = [C.c_int64, C.c_uint64, f64p]
        L.orc_skn_energy.restype = C.c_double
        L.orc_skn_energy.argtypes = [C.c_int64, f64p, u64p, C.c_void_p]
        L.orc_standard_mc_skn.restype = C.c_int64
        L.orc_standard_mc_skn.argtypes = [C.c_int64, f64p, C.c_double, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32,
                                          u64p, f64p, C.POINTER(C.c_int64), C.c_void_p]
        L.orc_rrr_mc_quant.restype = C.c_int64
        L.orc_rrr_mc_quant.argtypes = [C.c_int64, C.c_int64, C.c_int64, i32p, i32p, C.c_double, C.c_double, C.c_int64, C.c_int64,
                                       C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, i64p, C.c_void_p]
        L.orc_quant_energy.restype = C.c_double
        L.orc_quant_energy.argtypes = [C.c_int64, C.c_int64, C.c_int64, i32p, i32p, C.c_double, u64p, C.POINTER(C.c_double)]
        L.orc_colored_sweeps_sparse.restype = C.c_int64
        L.orc_colored_sweeps_sparse.argtypes = [C.c_int64, C.c_int64, i32p, i32p, i32p, C.c_int32, C.c_double, C.c_int64, C.c_int64,
                                                C.c_uint64, C.c_uint64, C.c_uint32, u64p, i64p, C.POINTER(C.c_int64)]
        L.orc_gen_sk_binary.argtypes = [C.c_int64, C.c_uint64, u64p]
        L.orc_skb_energy.restype = C.c_double
        L.orc_skb_energy.argtypes = [C.c_int64, u64p, u64p, C.c_void_p]
        L.orc_standard_mc_skb.restype = C.c_int64
        L.orc_standard_mc_skb.argtypes = [C.c_int64, u64p, C.c_double, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32,
                                          u64p, f64p, C.POINTER(C.c_int64), C.c_void_p]
        L.orc_rrr_mc_skn.restype = C.c_int64
        L.orc_rrr_mc_skn.argtypes = [C.c_int64, f64p, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_uint64, C.c_uint64,
                                     C.c_uint32, u64p, f64p, i64p, C.c_void_p, C.POINTER(C.c_double)]
        L.orc_dyns_test.restype = C.c_int64
        L.orc_dyns_test.argtypes = [C.c_int64, f64p, C.c_int64, i64p, f64p, C.c_int64, f64p, i64p, C.POINTER(C.c_double), C.c_void_p]
        L.orc_rrr_bkl_sparse.restype = C.c_int64
        L.orc_rrr_bkl_sparse.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, i32p, i32p, C.c_double, C.c_int64, C.c_int64,
                                         C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, u64p, i64p, i64p, C.c_void_p]
        L.orc_all_delta_e_pm1.restype = C.c_int64
        L.orc_all_delta_e_pm1.argtypes = [C.c_int64, i64p]
        _lib = L
    return _lib


# ------------------------------------------------------------------------------------------------
HOOK_FN = C.CFUNCTYPE(C.c_int, C.c_double, C.POINTER(C.c_uint64), C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_void_p)


class hooked:
    """``with hooked(fn) as calls:`` — every sampler of the oracle called inside the block calls ``fn(it, chunks, accepted, E, Emin) -> bool``
    at its samples, where the reference calls its ``hook`` (src/RRRMC.jl:107,187,256,341,404,501); False ends the chain there.  ``fn=None``
    records only.  ``calls`` collects ``(it, chunks.copy(), accepted, E, Emin)`` of every hook call of the thread."""

    def __init__(self, fn=None):
        self.fn, self.calls = fn, []

        def cb(it, chunks, nch, accepted, E, Emin, _user):
            ch = np.ctypeslib.as_array(chunks, shape=(nch,)).copy()
            self.calls.append((it, ch, int(accepted), E, Emin))
            return 1 if (self.fn is None or self.fn(it, ch, int(accepted), E, Emin)) else 0

        self._cb = HOOK_FN(cb)          # keep the trampoline alive for the duration of the block

    def __enter__(self):
        lib().orc_set_hook(self._cb, None)
        return self.calls

    def __exit__(self, *a):
        lib().orc_set_hook(C.cast(None, HOOK_FN), None)


def philox(ctr, key):
    out = np.zeros(4, np.uint32)
    lib().orc_philox(np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), out)
    return out


def site_of(seed, g, N):
    return int(lib().orc_site_of(seed, g, N))


def accept_uniform(seed, g, replica):
    return int(lib().orc_accept_uniform(seed, g, replica))


def accept_less(seed, g, replica, T):
    return bool(lib().orc_accept_less(seed, g, replica, T))


def threshold(p):
    a = C.c_int(0)
    t = lib().orc_threshold(p, C.byref(a))
    return int(t), bool(a.value)


def nchunks(N):
    return (N + 63) // 64


def init_config(seed, replica, N):
    ch = np.zeros(nchunks(N), np.uint64)
    lib().orc_init_config(seed, replica, N, ch)
    return ch


def init_configs(seed, replica0, R, N):
    return np.stack([init_config(seed, replica0 + r, N) for r in range(R)])


def gen_rrg(N, K, seed):
    A = np.zeros((N, K), np.int32)
    rc = lib().orc_gen_rrg(N, K, seed, A)
    if rc < 0:
        raise ValueError("gen_rrg failed rc=%d" % rc)
    return A


def gen_ea(L, D):
    A = np.zeros((L ** D, 2 * D), np.int32)
    rc = lib().orc_gen_ea(L, D, A)
    if rc < 0:
        raise ValueError("gen_ea failed rc=%d" % rc)
    return A


def gen_couplings(A, seed, lev=(-1, 1)):
    N, K = A.shape
    J = np.zeros((N, K), np.int32)
    rc = lib().orc_gen_couplings(N, K, np.ascontiguousarray(A), seed, len(lev), np.asarray(lev, np.int32), J)
    if rc < 0:
        raise ValueError("gen_couplings failed")
    return J


def sparse_energy(A, J, chunks, want_fields=False):
    N, K = A.shape
    lf = np.zeros(N, np.int64)
    E = lib().orc_sparse_energy(N, K, A, J, np.ascontiguousarray(chunks), lf.ctypes.data)
    return (int(E), lf) if want_fields else int(E)


FORM = {"rrg": 0, "ea": 1}


def standard_mc_sparse(A, J, beta, iters, step, seed, chunks, it0=0, replica=0, trace=False, form="rrg"):
    """One chain (form: which graph type's update_cache! is followed, 'rrg' = GraphRRG, 'ea' = GraphEA).  Returns (Es, chunks_out, accepted, lfields[, sites, flips])."""
    N, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.int64)
    acc = C.c_int64(0)
    lf = np.zeros(N, np.int64)
    sites = np.zeros(iters if trace else 1, np.int32)
    flips = np.zeros(iters if trace else 1, np.uint8)
    n = lib().orc_standard_mc_sparse(FORM[form], N, K, A, J, beta, iters, step, seed, it0, replica, ch, Es, C.byref(acc),
                                     lf.ctypes.data, sites.ctypes.data if trace else None,
                                     flips.ctypes.data if trace else None)
    out = (Es[:n], ch, int(acc.value), lf)
    return out + (sites, flips) if trace else out


def standard_mc_sparse_batch(A, J, beta, iters, step, seed, chunks, it0=0, replica0=0, form="rrg"):
    """R chains, chunks shaped [R, nchunks].  Returns (Es[R, nsamp], chunks_out, accepted[R])."""
    N, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    R = ch.shape[0]
    nsamp = iters // step
    Es = np.zeros((R, max(nsamp, 1)), np.int64)
    acc = np.zeros(R, np.int64)
    lib().orc_standard_mc_sparse_batch(FORM[form], N, K, A, J, beta, iters, step, seed, it0, replica0, R, ch,
                                       Es.reshape(-1)[: R * nsamp] if nsamp else Es.reshape(-1), acc)
    return Es[:, :nsamp] if nsamp else Es[:, :0], ch, acc


def all_delta_e_pm1(K):
    out = np.zeros(K + 1, np.int64)
    n = lib().orc_all_delta_e_pm1(K, out)
    return tuple(int(v) for v in out[:n])


# ---- dense Gaussian SK (GraphSKNormal) ------------------------------------------------------------
def det_exp(x):
    return float(lib().orc_exp(x))


def rand53(seed, g, replica):
    return float(lib().orc_rand53_of(seed, g, replica))


def gen_sk_gauss(N, seed):
    J = np.zeros((N, N), np.float64)
    lib().orc_gen_sk_gauss(N, seed, J.reshape(-1))
    return J


def skn_energy(J, chunks, want_fields=False):
    N = J.shape[0]
    lf = np.zeros(N, np.float64)
    E = lib().orc_skn_energy(N, np.ascontiguousarray(J).reshape(-1), np.ascontiguousarray(chunks), lf.ctypes.data)
    return (float(E), lf) if want_fields else float(E)


def standard_mc_skn(J, beta, iters, step, seed, chunks, it0=0, replica=0):
    """One chain on GraphSKNormal.  Returns (Es, chunks_out, accepted, lfields)."""
    N = J.shape[0]
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    acc = C.c_int64(0)
    lf = np.zeros(N, np.float64)
    n = lib().orc_standard_mc_skn(N, np.ascontiguousarray(J).reshape(-1), beta, iters, step, seed, it0, replica, ch, Es,
                                  C.byref(acc), lf.ctypes.data)
    return Es[:n], ch, int(acc.value), lf


def standard_mc_skn_batch(J, beta, iters, step, seed, chunks, it0=0, replica0=0):
    outs = [standard_mc_skn(J, beta, iters, step, seed, chunks[r], it0=it0, replica=replica0 + r) for r in range(len(chunks))]
    return (np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs]), np.array([o[2] for o in outs], np.int64),
            np.stack([o[3] for o in outs]))


# ---- GraphQuant (Suzuki-Trotter slices of a GraphRRG disorder) under rrrMC ------------------------------
def quant_fourK(beta, Gamma, M):
    """fourK = round(2/beta * log(coth(beta*Gamma/M)), digits=8): src/graphs/QT.jl:165"""
    import math
    x = beta * Gamma / M
    return round(2.0 / beta * math.log(1.0 / math.tanh(x)), 8)


def quant_energy(A, J, M, fourK, chunks):
    Nk, K = A.shape
    qt = C.c_double(0)
    E = lib().orc_quant_energy(Nk, M, K, A, J, fourK, np.ascontiguousarray(chunks), C.byref(qt))
    return float(E), float(qt.value)


def standard_mc_quant(A, J, M, fourK, beta, iters, step, seed, chunks, it0=0, replica=0):
    """standardMC on GraphQuant(GraphRRG slices).  Returns (Es, chunks_out, accepted)."""
    L = lib()
    L.orc_standard_mc_quant.restype = C.c_int64
    L.orc_standard_mc_quant.argtypes = [C.c_int64, C.c_int64, C.c_int64, i32p, i32p, C.c_double, C.c_double, C.c_int64, C.c_int64,
                                        C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, i64p]
    A = np.ascontiguousarray(A, np.int32)
    Nk, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    acc = np.zeros(1, np.int64)
    n = L.orc_standard_mc_quant(Nk, M, K, A, np.ascontiguousarray(J, np.int32), float(fourK), float(beta), int(iters), int(step), seed, it0,
                                replica, ch, Es, acc)
    return Es[:n], ch, int(acc[0])


def rrr_mc_quant(A, J, M, fourK, beta, iters, step, seed, chunks, it0=0, replica=0, staged_thr=0.5, staged_thr_fact=5.0,
                 want_cache=False):
    """One chain of rrrMC on GraphQuant.  Returns (Es, chunks_out, accepted, staged_its[, pos, set_sizes])."""
    Nk, K = A.shape
    N = Nk * M
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    stats = np.zeros(2, np.int64)
    cache = np.zeros(N + 4, np.int32)
    n = lib().orc_rrr_mc_quant(Nk, M, K, A, J, fourK, beta, iters, step, staged_thr, staged_thr_fact, seed, it0, replica, ch, Es,
                               stats, cache.ctypes.data if want_cache else None)
    if n < 0:
        raise AssertionError("DeltaECache / ArraySet consistency check failed")
    out = (Es[:n], ch, int(stats[0]), int(stats[1]))
    return out + (cache[:N].copy(), cache[N:].copy()) if want_cache else out


# ---- GraphQuant over binary GraphSK slices (GraphQSKT, src/QAliases.jl:34-43; scripts/scripts.jl test_QIsing) ----
def _jb(Jb):
    return np.ascontiguousarray(Jb, np.uint64).reshape(-1)


def quant_sk_energy(Jb, Nk, M, fourK, chunks):
    L = lib()
    L.orc_quant_energy_sk.restype = C.c_double
    L.orc_quant_energy_sk.argtypes = [C.c_int64, C.c_int64, u64p, C.c_double, u64p, C.POINTER(C.c_double)]
    qt = C.c_double(0)
    E = L.orc_quant_energy_sk(int(Nk), int(M), _jb(Jb), float(fourK), np.ascontiguousarray(chunks, np.uint64), C.byref(qt))
    return float(E), float(qt.value)


def standard_mc_quant_sk(Jb, Nk, M, fourK, beta, iters, step, seed, chunks, it0=0, replica=0):
    """standardMC on GraphQuant(GraphSK slices).  Returns (Es, chunks_out, accepted)."""
    L = lib()
    L.orc_standard_mc_quant_sk.restype = C.c_int64
    L.orc_standard_mc_quant_sk.argtypes = [C.c_int64, C.c_int64, u64p, C.c_double, C.c_double, C.c_int64, C.c_int64,
                                           C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, i64p]
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    acc = np.zeros(1, np.int64)
    n = L.orc_standard_mc_quant_sk(int(Nk), int(M), _jb(Jb), float(fourK), float(beta), int(iters), int(step), seed, it0, replica, ch, Es, acc)
    return Es[:n], ch, int(acc[0])


def rrr_mc_quant_sk(Jb, Nk, M, fourK, beta, iters, step, seed, chunks, it0=0, replica=0, staged_thr=0.5, staged_thr_fact=5.0,
                    want_cache=False):
    """One chain of rrrMC on GraphQuant(GraphSK slices).  Returns (Es, chunks_out, accepted, staged_its[, pos, set_sizes])."""
    L = lib()
    L.orc_rrr_mc_quant_sk.restype = C.c_int64
    L.orc_rrr_mc_quant_sk.argtypes = [C.c_int64, C.c_int64, u64p, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                      C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, i64p, C.c_void_p]
    N = int(Nk) * int(M)
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    stats = np.zeros(2, np.int64)
    cache = np.zeros(N + 4, np.int32)
    n = L.orc_rrr_mc_quant_sk(int(Nk), int(M), _jb(Jb), float(fourK), float(beta), int(iters), int(step), float(staged_thr),
                              float(staged_thr_fact), seed, it0, replica, ch, Es, stats, cache.ctypes.data if want_cache else None)
    if n < 0:
        raise AssertionError("DeltaECache / ArraySet consistency check failed")
    out = (Es[:n], ch, int(stats[0]), int(stats[1]))
    return out + (cache[:N].copy(), cache[N:].copy()) if want_cache else out


def rrr_mc_quant_skn(Jd, Nk, M, fourK, beta, iters, step, seed, chunks, it0=0, replica=0, staged_thr=0.5, staged_thr_fact=5.0,
                     want_cache=False):
    """One chain of rrrMC on GraphQuant over GraphSKNormal slices (GraphQSKNormalT; test/runtests.jl:80).
    Returns (Es, chunks_out, accepted, staged_its[, pos, set_sizes])."""
    L = lib()
    L.orc_rrr_mc_quant_skn.restype = C.c_int64
    L.orc_rrr_mc_quant_skn.argtypes = [C.c_int64, C.c_int64, f64p, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                       C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, i64p, C.c_void_p]
    N = int(Nk) * int(M)
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    stats = np.zeros(2, np.int64)
    cache = np.zeros(N + 4, np.int32)
    n = L.orc_rrr_mc_quant_skn(int(Nk), int(M), np.ascontiguousarray(Jd, np.float64).reshape(-1), float(fourK), float(beta), int(iters), int(step),
                               float(staged_thr), float(staged_thr_fact), seed, it0, replica, ch, Es, stats, cache.ctypes.data if want_cache else None)
    if n < 0:
        raise AssertionError("DeltaECache / ArraySet consistency check failed")
    out = (Es[:n], ch, int(stats[0]), int(stats[1]))
    return out + (cache[:N].copy(), cache[N:].copy()) if want_cache else out


def standard_mc_quant_skn(Jd, Nk, M, fourK, beta, iters, step, seed, chunks, it0=0, replica=0):
    L = lib()
    L.orc_standard_mc_quant_skn.restype = C.c_int64
    L.orc_standard_mc_quant_skn.argtypes = [C.c_int64, C.c_int64, f64p, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64,
                                            C.c_uint32, u64p, f64p, C.POINTER(C.c_int64)]
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    acc = C.c_int64(0)
    n = L.orc_standard_mc_quant_skn(int(Nk), int(M), np.ascontiguousarray(Jd, np.float64).reshape(-1), float(fourK), float(beta), int(iters),
                                    int(step), seed, it0, replica, ch, Es, C.byref(acc))
    return Es[:n], ch, int(acc.value)


def quant_skn_energy(Jd, Nk, M, fourK, chunks):
    L = lib()
    L.orc_quant_energy_skn.restype = C.c_double
    L.orc_quant_energy_skn.argtypes = [C.c_int64, C.c_int64, f64p, C.c_double, u64p]
    return float(L.orc_quant_energy_skn(int(Nk), int(M), np.ascontiguousarray(Jd, np.float64).reshape(-1), float(fourK),
                                        np.ascontiguousarray(chunks, np.uint64)))


def quant_sk_observables(Jb, Nk, M, fourK, beta, Gamma, chunks):
    """(Qenergy, transverse_mag, overlaps[M//2], energy0, Eslice[M] (the integers n_k: E_k = n_k / sqrt(Nk)), ovs_raw[M//2])."""
    L = lib()
    L.orc_quant_observables_sk.restype = C.c_int
    L.orc_quant_observables_sk.argtypes = [C.c_int64, C.c_int64, u64p, C.c_double, C.c_double, C.c_double, u64p,
                                           C.POINTER(C.c_double), C.POINTER(C.c_double), f64p, C.POINTER(C.c_int64), i64p, i64p]
    Q, tm, e0 = C.c_double(0), C.c_double(0), C.c_int64(0)
    ovs = np.zeros(max(M // 2, 1))
    Es = np.zeros(M, np.int64)
    raw = np.zeros(max(M // 2, 1), np.int64)
    L.orc_quant_observables_sk(int(Nk), int(M), _jb(Jb), float(fourK), float(beta), float(Gamma), np.ascontiguousarray(chunks, np.uint64),
                               C.byref(Q), C.byref(tm), ovs, C.byref(e0), Es, raw)
    return Q.value, tm.value, ovs[:M // 2], e0.value, Es, raw[:M // 2]


# ---- colour-parallel sweeps (build-defined "checkerboard" sampler) -------------------------------------
def checkerboard_coloring(L, D):
    """Parity colouring of the periodic L^D lattice in the reference's column-major site order (EA.jl:24-43); L even."""
    x = np.arange(L ** D)
    par = np.zeros_like(x)
    for d in range(D):
        par += (x // L ** d) % L
    return (par % 2).astype(np.int32)


def greedy_coloring(A):
    N = A.shape[0]
    col = -np.ones(N, np.int32)
    for x in range(N):
        used = {int(col[y]) for y in A[x] if col[y] >= 0}
        c = 0
        while c in used:
            c += 1
        col[x] = c
    return col


def colored_sweeps_sparse(A, J, color, beta, sweeps, step, seed, chunks, sweep0=0, replica=0):
    """One chain.  Returns (Es, chunks_out, accepted)."""
    N, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(sweeps // step, 1), np.int64)
    acc = C.c_int64(0)
    color = np.ascontiguousarray(color, np.int32)
    n = lib().orc_colored_sweeps_sparse(N, K, A, J, color, int(color.max()) + 1, beta, sweeps, step, seed, sweep0, replica, ch, Es,
                                        C.byref(acc))
    return Es[:n], ch, int(acc.value)


# ---- binary SK (GraphSK) ----------------------------------------------------------------------------
def gen_sk_binary(N, seed):
    """J as [N, ceil(N/64)] BitVector chunk rows."""
    J = np.zeros((N, nchunks(N)), np.uint64)
    lib().orc_gen_sk_binary(N, seed, J.reshape(-1))
    return J


def skb_energy(J, chunks, want_fields=False):
    N = J.shape[0]
    lf = np.zeros(N, np.int64)
    E = lib().orc_skb_energy(N, np.ascontiguousarray(J).reshape(-1), np.ascontiguousarray(chunks), lf.ctypes.data)
    return (float(E), lf) if want_fields else float(E)


def standard_mc_skb(J, beta, iters, step, seed, chunks, it0=0, replica=0):
    N = J.shape[0]
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    acc = C.c_int64(0)
    lf = np.zeros(N, np.int64)
    n = lib().orc_standard_mc_skb(N, np.ascontiguousarray(J).reshape(-1), beta, iters, step, seed, it0, replica, ch, Es,
                                  C.byref(acc), lf.ctypes.data)
    return Es[:n], ch, int(acc.value), lf


# ---- continuous-energy RRR path: DynamicSampler + DeltaECacheCont + rrrMC(SingleGraph) on GraphSKNormal ----
def dyns_test(v, upd_i, upd_x, xs):
    """Build a DynamicSampler from v, apply setindex!(i, x) updates, then getel for each uniform in xs.
    Returns (elements (0-based), z, ps)."""
    v = np.ascontiguousarray(v, np.float64)
    N = len(v)
    upd_i = np.ascontiguousarray(upd_i, np.int64); upd_x = np.ascontiguousarray(upd_x, np.float64)
    xs = np.ascontiguousarray(xs, np.float64)
    out = np.zeros(max(len(xs), 1), np.int64)
    z = C.c_double(0)
    levs = max(int(np.ceil(np.log2(N))), 0) if N > 1 else 0
    ps = np.zeros(max((1 << levs) - 1, 1), np.float64)
    lib().orc_dyns_test(N, v, len(upd_i), upd_i if len(upd_i) else np.zeros(1, np.int64), upd_x if len(upd_x) else np.zeros(1),
                        len(xs), xs if len(xs) else np.zeros(1), out, C.byref(z), ps.ctypes.data)
    return out[: len(xs)], float(z.value), ps


def rrr_mc_skn(J, beta, iters, step, seed, chunks, it0=0, replica=0, staged_thr=0.8, staged_thr_fact=5.0, want_cache=False):
    """One chain of rrrMC on GraphSKNormal.  Returns (Es, chunks_out, accepted, staged_its[, dEs, z])."""
    N = J.shape[0]
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    stats = np.zeros(2, np.int64)
    dEs = np.zeros(N, np.float64)
    z = C.c_double(0)
    n = lib().orc_rrr_mc_skn(N, np.ascontiguousarray(J).reshape(-1), beta, iters, step, staged_thr, staged_thr_fact, seed, it0, replica,
                             ch, Es, stats, dEs.ctypes.data, C.byref(z))
    if n < 0:
        raise AssertionError("Unrecoverable loss of precision detected in the dynamic sampler")
    out = (Es[:n], ch, int(stats[0]), int(stats[1]))
    return out + (dEs, float(z.value)) if want_cache else out


# ---- rrrMC(SingleGraph) / bklMC on the DiscrGraphs GraphRRG / GraphEA --------------------------------
def rrr_sparse(A, J, beta, iters, step, seed, chunks, it0=0, replica=0, staged_thr=0.5, staged_thr_fact=5.0, form="rrg", bkl=False,
               want_cache=False, lev=None, mul=1, div=1.0):
    """One chain of rrrMC (or bklMC with bkl=True).  Returns (Es, chunks_out, accepted, staged_its_or_moves, iters_done[, pos, sizes]).
    lev = None: the +-J graphs; otherwise a stand-alone GraphRRG / GraphEA with levels lev (units, value = units * mul / div)."""
    N, K = A.shape
    A = np.ascontiguousarray(A, np.int32)
    J = np.ascontiguousarray(J, np.int32)
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1) + 1, np.int64)
    stats = np.zeros(3, np.int64)
    L = len(all_delta_e_pm1(K)) if lev is None else len(all_delta_e(K, lev))
    cache = np.zeros(N + 2 * L, np.int32)
    if lev is None:
        n = lib().orc_rrr_bkl_sparse(1 if bkl else 0, FORM[form], N, K, A, J, beta, iters, step, staged_thr, staged_thr_fact, seed, it0,
                                     replica, ch, Es, stats, cache.ctypes.data if want_cache else None)
    else:
        F = lib().orc_rrr_bkl_sparse_lev
        F.restype = C.c_int64
        F.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, i32p, i32p, i32p, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_int64,
                      C.c_int64, C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, u64p, i64p, i64p, C.c_void_p]
        n = F(1 if bkl else 0, FORM[form], N, K, A, J, np.asarray(lev, np.int32), len(lev), int(mul), float(div), beta, iters, step,
              staged_thr, staged_thr_fact, seed, it0, replica, ch, Es, stats, cache.ctypes.data if want_cache else None)
    if n < 0:
        raise AssertionError("DeltaECache / ArraySet consistency check failed")
    out = (Es[:n], ch, int(stats[0]), int(stats[1]), int(stats[2]))
    return out + (cache[:N].copy(), cache[N:].copy()) if want_cache else out


def pm1dot(a, b, N):
    """scripts/scripts.jl:283-295"""
    L = lib()
    L.orc_pm1dot.restype = C.c_int64
    L.orc_pm1dot.argtypes = [u64p, u64p, C.c_int64]
    return int(L.orc_pm1dot(np.ascontiguousarray(a, np.uint64), np.ascontiguousarray(b, np.uint64), int(N)))


def q2_window(Cs, N, i, j):
    """parseovs window statistic (scripts/scripts.jl:380-401) over samples [i, j) of one chain; Cs[samples, nch]."""
    L = lib()
    L.orc_q2_window.restype = C.c_int
    L.orc_q2_window.argtypes = [u64p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    m, s = C.c_double(0), C.c_double(0)
    rc = L.orc_q2_window(np.ascontiguousarray(Cs, np.uint64), int(N), int(i), int(j), C.byref(m), C.byref(s))
    return (m.value, s.value) if rc == 0 else (float("nan"), float("nan"))


def quant_observables(A, J, M, fourK, beta, Gamma, chunks):
    """(Qenergy, transverse_mag, overlaps[M//2], energy0, Eslice[M], ovs_raw[M//2]) — QT.jl:113-122, 213-268."""
    L = lib()
    L.orc_quant_observables.restype = C.c_int
    L.orc_quant_observables.argtypes = [C.c_int64, C.c_int64, C.c_int64, i32p, i32p, C.c_double, C.c_double, C.c_double, u64p,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double), f64p, C.POINTER(C.c_int64), i64p, i64p]
    A = np.ascontiguousarray(A, np.int32)
    Nk, K = A.shape
    Q, tm, e0 = C.c_double(0), C.c_double(0), C.c_int64(0)
    ovs = np.zeros(max(M // 2, 1))
    Es = np.zeros(M, np.int64)
    raw = np.zeros(max(M // 2, 1), np.int64)
    L.orc_quant_observables(Nk, int(M), K, A, np.ascontiguousarray(J, np.int32), float(fourK), float(beta), float(Gamma),
                            np.ascontiguousarray(chunks, np.uint64), C.byref(Q), C.byref(tm), ovs, C.byref(e0), Es, raw)
    return Q.value, tm.value, ovs[:M // 2], e0.value, Es, raw[:M // 2]


def gen_couplings_gauss(A, seed):
    """gen_J(Float64, N, A) do randn() end — RRG.jl:71-96 / EA.jl:45-71"""
    L = lib()
    L.orc_gen_couplings_gauss.restype = C.c_int
    L.orc_gen_couplings_gauss.argtypes = [C.c_int64, C.c_int64, i32p, C.c_uint64, f64p]
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    J = np.zeros((N, K))
    if L.orc_gen_couplings_gauss(N, K, A, seed, J.reshape(-1)) < 0:
        raise ValueError("gen_couplings_gauss failed")
    return J


def spf_energy(A, J, chunks, want_fields=False, form="rrg"):
    """energy of GraphRRGNormal / GraphEANormal (RRG.jl:546-574, EA.jl:584-611)"""
    L = lib()
    L.orc_spf_energy.restype = C.c_double
    L.orc_spf_energy.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, f64p, u64p, C.c_void_p]
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    lf = np.zeros(N)
    E = L.orc_spf_energy(1 if form == "ea" else 0, N, K, A, np.ascontiguousarray(J, np.float64).reshape(-1),
                         np.ascontiguousarray(chunks, np.uint64), lf.ctypes.data if want_fields else None)
    return (float(E), lf) if want_fields else float(E)


def standard_mc_spf(A, J, beta, iters, step, seed, chunks, it0=0, replica=0, form="rrg"):
    """standardMC on GraphRRGNormal / GraphEANormal; returns (Es, final chunks, accepted, final lfields)."""
    L = lib()
    L.orc_standard_mc_spf.restype = C.c_int64
    L.orc_standard_mc_spf.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, f64p, C.c_double, C.c_int64, C.c_int64, C.c_uint64,
                                      C.c_uint64, C.c_uint32, u64p, f64p, C.POINTER(C.c_int64), f64p]
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1))
    acc = C.c_int64(0)
    lf = np.zeros(N)
    n = L.orc_standard_mc_spf(1 if form == "ea" else 0, N, K, A, np.ascontiguousarray(J, np.float64).reshape(-1), float(beta), int(iters),
                              int(step), seed, it0, replica, ch, Es, C.byref(acc), lf)
    return Es[:n], ch, acc.value, lf


def standard_mc_spf_fast(A, J, beta, iters, step, seed, chunks, it0=0, replica=0, form="rrg"):
    """standardMC on GraphRRGNormal / GraphEANormal in the library's FAST mode (pattern delta_energy, ACCEPT bit-plane stream);
    returns (Es, chunks_out, accepted)."""
    L = lib()
    L.orc_standard_mc_spf_fast.restype = C.c_int64
    L.orc_standard_mc_spf_fast.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, f64p, C.c_double, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64,
                                           C.c_uint32, u64p, f64p, C.POINTER(C.c_int64)]
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    acc = C.c_int64(0)
    n = L.orc_standard_mc_spf_fast(1 if form == "ea" else 0, N, K, A, np.ascontiguousarray(J, np.float64).reshape(-1), float(beta), int(iters),
                                   int(step), seed, it0, replica, ch, Es, C.byref(acc))
    return Es[:n], ch, int(acc.value)


def all_delta_e(K, lev):
    """allΔE for integer levels (RRG.jl:268-281, EA.jl:295-309)"""
    L = lib()
    L.orc_all_delta_e.restype = C.c_int64
    L.orc_all_delta_e.argtypes = [C.c_int64, i32p, C.c_int64, i64p, C.c_int64]
    out = np.zeros(64, np.int64)
    n = L.orc_all_delta_e(int(K), np.asarray(lev, np.int32), len(lev), out, 64)
    if n < 0:
        raise ValueError("too many levels")
    return tuple(int(v) for v in out[:n])


def dfloat_units(LEV):
    """Float64 levels -> (units, mul, div): DFloat64(x) is the Int64 t = round(x * 10^5) (src/DFloats.jl:11-24, ties to even);
    units = t / g with g = gcd of the |t| so that they fit narrow tables; value = units * mul / div with (mul, div) = (g, 1e5).
    Integer levels -> (LEV, 1, 1.0)."""
    if all(isinstance(l, (int, np.integer)) for l in LEV):
        return tuple(int(l) for l in LEV), 1, 1.0
    import math
    t = [int(np.rint(float(l) * 100000.0)) for l in LEV]
    g = 0
    for x in t:
        g = math.gcd(g, abs(x))
    g = max(g, 1)
    return tuple(x // g for x in t), g, 100000.0


def discretize(x, lev, mul=1, div=1.0):
    """discretize (Common.jl:38-72): (levels [units], residuals) with the shape of x"""
    L = lib()
    L.orc_discretize_scaled.restype = None
    L.orc_discretize_scaled.argtypes = [f64p, C.c_int64, i32p, C.c_int64, C.c_int64, C.c_double, i32p, f64p]
    x = np.ascontiguousarray(x, np.float64)
    d = np.zeros(x.shape, np.int32)
    r = np.zeros(x.shape, np.float64)
    L.orc_discretize_scaled(x.reshape(-1), x.size, np.asarray(lev, np.int32), len(lev), int(mul), float(div), d.reshape(-1), r.reshape(-1))
    return d, r


def dbl_energy(A, dJ, rJ, chunks, form="rrg", mul=1, div=1.0):
    L = lib()
    L.orc_dbl_energy_scaled.restype = C.c_double
    L.orc_dbl_energy_scaled.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, i32p, f64p, u64p, C.c_int64, C.c_double]
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    return float(L.orc_dbl_energy_scaled(1 if form == "ea" else 0, N, K, A, np.ascontiguousarray(dJ, np.int32).reshape(-1),
                                         np.ascontiguousarray(rJ, np.float64).reshape(-1), np.ascontiguousarray(chunks, np.uint64),
                                         int(mul), float(div)))


def rrr_double_sparse(A, dJ, rJ, lev, beta, iters, step, seed, chunks, it0=0, replica=0, staged_thr=0.5, staged_thr_fact=5.0,
                      form="rrg", mul=1, div=1.0):
    """rrrMC(X::DoubleGraph) on Graph{RRG,EA}NormalDiscretized; returns (Es, chunks, accepted, staged_its, pos[N], sizes[2L])."""
    L = lib()
    L.orc_rrr_double_sparse_scaled.restype = C.c_int64
    L.orc_rrr_double_sparse_scaled.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, i32p, f64p, i32p, C.c_int64, C.c_int64, C.c_double,
                                               C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_uint64, C.c_uint64,
                                               C.c_uint32, u64p, f64p, i64p, i32p]
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    nl = len(all_delta_e(K, lev))
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1))
    stats = np.zeros(2, np.int64)
    cache = np.zeros(N + 2 * nl, np.int32)
    n = L.orc_rrr_double_sparse_scaled(1 if form == "ea" else 0, N, K, A, np.ascontiguousarray(dJ, np.int32).reshape(-1),
                                       np.ascontiguousarray(rJ, np.float64).reshape(-1), np.asarray(lev, np.int32), len(lev),
                                       int(mul), float(div), float(beta), int(iters), int(step), float(staged_thr),
                                       float(staged_thr_fact), seed, it0, replica, ch, Es, stats, cache)
    if n < 0:
        raise RuntimeError("rrr_double_sparse: cache inconsistent (rc=%d)" % n)
    return Es[:n], ch, int(stats[0]), int(stats[1]), cache[:N].copy(), cache[N:].copy()


def standard_mc_dbl(A, dJ, rJ, beta, iters, step, seed, chunks, it0=0, replica=0, form="rrg", mul=1, div=1.0):
    """standardMC on Graph{RRG,EA}NormalDiscretized; returns (Es, chunks, accepted)."""
    L = lib()
    L.orc_standard_mc_dbl.restype = C.c_int64
    L.orc_standard_mc_dbl.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, i32p, f64p, C.c_int64, C.c_double, C.c_double, C.c_int64,
                                      C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, i64p]
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1))
    acc = np.zeros(1, np.int64)
    n = L.orc_standard_mc_dbl(1 if form == "ea" else 0, N, K, A, np.ascontiguousarray(dJ, np.int32).reshape(-1),
                              np.ascontiguousarray(rJ, np.float64).reshape(-1), int(mul), float(div), float(beta), int(iters), int(step),
                              seed, it0, replica, ch, Es, acc)
    return Es[:n], ch, int(acc[0])


def standard_mc_lev(A, J, beta, iters, step, seed, chunks, it0=0, replica=0, form="rrg", mul=1, div=1.0):
    """standardMC on a stand-alone GraphRRG / GraphEA with general levels (J in units); returns (Es [units], chunks, accepted)."""
    L = lib()
    L.orc_standard_mc_lev.restype = C.c_int64
    L.orc_standard_mc_lev.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, i32p, C.c_int64, C.c_double, C.c_double, C.c_int64, C.c_int64,
                                      C.c_uint64, C.c_uint64, C.c_uint32, u64p, i64p, i64p]
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.int64)
    acc = np.zeros(1, np.int64)
    n = L.orc_standard_mc_lev(1 if form == "ea" else 0, N, K, A, np.ascontiguousarray(J, np.int32), int(mul), float(div), float(beta),
                              int(iters), int(step), seed, it0, replica, ch, Es, acc)
    if n < 0:
        raise RuntimeError("standard_mc_lev: tracked energy != energy(X, C)")
    return Es[:n], ch, int(acc[0])


def wtm_mc_sparse(A, J, beta, samples, step, seed, chunks, call=0, replica=0, form="rrg", mul=1, div=1.0):
    """wtmMC (RRRMC.jl:376-426) on GraphRRG / GraphEA (any levels: J in units, value = units * mul / div);
    returns (Es, chunks, num_moves, t, E_final)."""
    L = lib()
    L.orc_wtm_mc_sparse_lev.restype = C.c_int64
    L.orc_wtm_mc_sparse_lev.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, i32p, C.c_int64, C.c_double, C.c_double, C.c_int64, C.c_double,
                                        C.c_uint64, C.c_uint32, C.c_uint32, u64p, i64p, i64p, C.POINTER(C.c_double)]
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(samples, 1), np.int64)
    stats = np.zeros(3, np.int64)
    t = C.c_double(0)
    n = L.orc_wtm_mc_sparse_lev(1 if form == "ea" else 0, N, K, A, np.ascontiguousarray(J, np.int32), int(mul), float(div), float(beta),
                                int(samples), float(step), seed, call, replica, ch, Es, stats, C.byref(t))
    return Es[:n], ch, int(stats[0]), t.value, int(stats[2])


def eo_ftau(N, tau):
    """cumsum([j^(-tau) for j = 1:N]) (DeltaE.jl:444): the rank-probability table, an INPUT of both oracle and device."""
    return np.cumsum(np.arange(1, int(N) + 1, dtype=np.float64) ** (-float(tau)))


def extremal_opt_sparse(A, J, tau, iters, step, seed, chunks, it0=0, replica=0, form="rrg", lev=None):
    """extremal_opt (RRRMC.jl:474-521) on GraphRRG / GraphEA (lev = None: +-J); returns (Es, final chunks, Emin, Cmin chunks, itmin)."""
    L = lib()
    L.orc_extremal_opt_sparse_lev.restype = C.c_int64
    L.orc_extremal_opt_sparse_lev.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, i32p, C.c_void_p, C.c_int64, f64p, C.c_int64, C.c_int64,
                                              C.c_uint64, C.c_uint64, C.c_uint32, u64p, i64p, C.POINTER(C.c_int64), u64p,
                                              C.POINTER(C.c_int64)]
    levp = None if lev is None else np.ascontiguousarray(lev, np.int32)
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.int64)
    Cmin = np.zeros_like(ch)
    Emin, itmin = C.c_int64(0), C.c_int64(0)
    n = L.orc_extremal_opt_sparse_lev(1 if form == "ea" else 0, N, K, A, np.ascontiguousarray(J, np.int32),
                                      None if levp is None else levp.ctypes.data, 0 if levp is None else len(levp), eo_ftau(N, tau),
                                      int(iters), int(step), seed, it0, replica, ch, Es, C.byref(Emin), Cmin, C.byref(itmin))
    if n < 0:
        raise RuntimeError("extremal_opt_sparse: inconsistent cache / energy")
    return Es[:n], ch, Emin.value, Cmin, itmin.value


def cont_sparse(mode, A, J, beta, iters, step, seed, chunks, it0=0, call=0, replica=0, staged_thr=0.8, staged_thr_fact=5.0,
                stepf=1.0, form="rrg"):
    """rrrMC (mode "rrr"), bklMC ("bkl"), wtmMC ("wtm") on GraphRRGNormal / GraphEANormal with the continuous-energy caches.
    Returns (Es, chunks, stats[3], t)."""
    L = lib()
    L.orc_cont_sparse.restype = C.c_int64
    L.orc_cont_sparse.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, i32p, f64p, C.c_double, C.c_int64, C.c_int64, C.c_double,
                                  C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u64p, f64p, i64p,
                                  C.POINTER(C.c_double)]
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    m = {"rrr": 0, "bkl": 1, "wtm": 2}[mode]
    nmax = iters if m == 2 else max(iters // step, 1)
    Es = np.zeros(max(nmax, 1))
    stats = np.zeros(3, np.int64)
    t = C.c_double(0)
    n = L.orc_cont_sparse(m, 1 if form == "ea" else 0, N, K, A, np.ascontiguousarray(J, np.float64).reshape(-1), float(beta), int(iters),
                          int(step), float(stepf), float(staged_thr), float(staged_thr_fact), seed, it0, call, replica, ch, Es, stats,
                          C.byref(t))
    if n < 0:
        raise RuntimeError("cont_sparse: DynamicSampler lost precision")
    return Es[:n], ch, stats, t.value


def cont_double(mode, A, dJ, rJ, beta, iters, step, seed, chunks, it0=0, call=0, replica=0, stepf=1.0, form="rrg", mul=1, div=1.0):
    """bklMC ("bkl") / wtmMC ("wtm") on Graph{RRG,EA}NormalDiscretized: the continuous-energy caches over the whole DoubleGraph.
    Returns (Es, chunks, stats[3], t)."""
    L = lib()
    L.orc_cont_double.restype = C.c_int64
    L.orc_cont_double.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, i32p, i32p, f64p, C.c_int64, C.c_double, C.c_double, C.c_int64,
                                  C.c_int64, C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u64p, f64p, i64p,
                                  C.POINTER(C.c_double)]
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    m = {"bkl": 1, "wtm": 2}[mode]
    nmax = iters if m == 2 else max(iters // step, 1)
    Es = np.zeros(max(nmax, 1))
    stats = np.zeros(3, np.int64)
    t = C.c_double(0)
    n = L.orc_cont_double(m, 1 if form == "ea" else 0, N, K, A, np.ascontiguousarray(dJ, np.int32).reshape(-1),
                          np.ascontiguousarray(rJ, np.float64).reshape(-1), int(mul), float(div), float(beta), int(iters), int(step),
                          float(stepf), seed, it0, call, replica, ch, Es, stats, C.byref(t))
    if n < 0:
        raise RuntimeError("cont_double: DynamicSampler lost precision")
    return Es[:n], ch, stats, t.value


def extremal_opt_cont(A, J, tau, iters, step, seed, chunks, it0=0, replica=0, form="rrg", dJ=None, mul=1, div=1.0):
    """extremal_opt (RRRMC.jl:474-521) with EOCacheCont (DeltaE.jl:557-635) on GraphRRGNormal / GraphEANormal (J Float64) or, with dJ
    (level units) and J = the residuals, on the discretised DoubleGraphs.  Returns (Es, final chunks, Emin, Cmin chunks, itmin)."""
    L = lib()
    L.orc_extremal_opt_cont.restype = C.c_int64
    L.orc_extremal_opt_cont.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, f64p, C.c_void_p, C.c_int64, C.c_double, f64p, C.c_int64, C.c_int64,
                                        C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, C.POINTER(C.c_double), u64p, C.POINTER(C.c_int64)]
    A = np.ascontiguousarray(A, np.int32)
    N, K = A.shape
    dJp = None if dJ is None else np.ascontiguousarray(dJ, np.int32).reshape(-1)
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1))
    Cmin = np.zeros_like(ch)
    Emin, itmin = C.c_double(0), C.c_int64(0)
    n = L.orc_extremal_opt_cont(1 if form == "ea" else 0, N, K, A, np.ascontiguousarray(J, np.float64).reshape(-1),
                                None if dJp is None else dJp.ctypes.data, int(mul), float(div), eo_ftau(N, tau), int(iters), int(step),
                                seed, it0, replica, ch, Es, C.byref(Emin), Cmin, C.byref(itmin))
    if n < 0:
        raise RuntimeError("extremal_opt_cont: inconsistent cache / energy (%d)" % n)
    return Es[:n], ch, Emin.value, Cmin, itmin.value


def extremal_opt_quant(A, J, M, fourK, tau, iters, step, seed, chunks, it0=0, replica=0, form="rrg"):
    """extremal_opt on a GraphQuant over GraphRRG / GraphEA slices (EOCacheCont over all Nk M spins).
    Returns (Es, final chunks, Emin, Cmin chunks, itmin)."""
    L = lib()
    L.orc_extremal_opt_quant.restype = C.c_int64
    L.orc_extremal_opt_quant.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, i32p, i32p, C.c_double, f64p, C.c_int64, C.c_int64,
                                         C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, C.POINTER(C.c_double), u64p, C.POINTER(C.c_int64)]
    A = np.ascontiguousarray(A, np.int32)
    Nk, K = A.shape
    N = Nk * int(M)
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1))
    Cmin = np.zeros_like(ch)
    Emin, itmin = C.c_double(0), C.c_int64(0)
    n = L.orc_extremal_opt_quant(1 if form == "ea" else 0, Nk, int(M), K, A, np.ascontiguousarray(J, np.int32), float(fourK), eo_ftau(N, tau),
                                 int(iters), int(step), seed, it0, replica, ch, Es, C.byref(Emin), Cmin, C.byref(itmin))
    if n < 0:
        raise RuntimeError("extremal_opt_quant: inconsistent cache / energy (%d)" % n)
    return Es[:n], ch, Emin.value, Cmin, itmin.value


def wtm_mc_skn(J, beta, samples, step, seed, chunks, call=0, replica=0):
    """wtmMC on GraphSKNormal; returns (Es, chunks, num_moves, t)."""
    L = lib()
    L.orc_wtm_mc_skn.restype = C.c_int64
    L.orc_wtm_mc_skn.argtypes = [C.c_int64, f64p, C.c_double, C.c_int64, C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, u64p, f64p, i64p,
                                 C.POINTER(C.c_double)]
    J = np.ascontiguousarray(J, np.float64)
    N = J.shape[0]
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(samples, 1))
    stats = np.zeros(2, np.int64)
    t = C.c_double(0)
    n = L.orc_wtm_mc_skn(N, J.reshape(-1), float(beta), int(samples), float(step), seed, call, replica, ch, Es, stats, C.byref(t))
    if n < 0:
        raise RuntimeError("wtm_mc_skn: tracked energy != energy(X, C)")
    return Es[:n], ch, int(stats[0]), t.value


def bkl_mc_skn(J, beta, iters, step, seed, chunks, it0=0, replica=0):
    """bklMC on GraphSKNormal (continuous-energy cache); returns (Es, chunks, moves, iterations done)."""
    L = lib()
    L.orc_bkl_mc_skn.restype = C.c_int64
    L.orc_bkl_mc_skn.argtypes = [C.c_int64, f64p, C.c_double, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, i64p]
    J = np.ascontiguousarray(J, np.float64)
    N = J.shape[0]
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1))
    stats = np.zeros(3, np.int64)
    n = L.orc_bkl_mc_skn(N, J.reshape(-1), float(beta), int(iters), int(step), seed, it0, replica, ch, Es, stats)
    if n < 0:
        raise RuntimeError("bkl_mc_skn: DynamicSampler lost precision")
    return Es[:n], ch, int(stats[0]), int(stats[2])


# ---- the continuous-energy samplers on the binary GraphSK (a SimpleGraph{Float64} too: SK.jl:28) ---------------------------------
def rrr_mc_skb(Jb, beta, iters, step, seed, chunks, it0=0, replica=0, staged_thr=0.8, staged_thr_fact=5.0):
    """One chain of rrrMC on the binary GraphSK.  Returns (Es, chunks_out, accepted, staged_its)."""
    L = lib()
    L.orc_rrr_mc_skb.restype = C.c_int64
    L.orc_rrr_mc_skb.argtypes = [C.c_int64, u64p, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_uint64, C.c_uint64,
                                 C.c_uint32, u64p, f64p, i64p, C.c_void_p, C.c_void_p]
    N = Jb.shape[0]
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    stats = np.zeros(2, np.int64)
    n = L.orc_rrr_mc_skb(N, np.ascontiguousarray(Jb, np.uint64).reshape(-1), float(beta), int(iters), int(step), float(staged_thr),
                         float(staged_thr_fact), seed, it0, replica, ch, Es, stats, None, None)
    if n < 0:
        raise AssertionError("Unrecoverable loss of precision detected in the dynamic sampler")
    return Es[:n], ch, int(stats[0]), int(stats[1])


def bkl_mc_skb(Jb, beta, iters, step, seed, chunks, it0=0, replica=0):
    """bklMC on the binary GraphSK; returns (Es, chunks, moves, iterations done)."""
    L = lib()
    L.orc_bkl_mc_skb.restype = C.c_int64
    L.orc_bkl_mc_skb.argtypes = [C.c_int64, u64p, C.c_double, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, i64p]
    N = Jb.shape[0]
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1))
    stats = np.zeros(3, np.int64)
    n = L.orc_bkl_mc_skb(N, np.ascontiguousarray(Jb, np.uint64).reshape(-1), float(beta), int(iters), int(step), seed, it0, replica, ch, Es, stats)
    if n < 0:
        raise RuntimeError("bkl_mc_skb: DynamicSampler lost precision")
    return Es[:n], ch, int(stats[0]), int(stats[2])


def wtm_mc_skb(Jb, beta, samples, step, seed, chunks, call=0, replica=0):
    """wtmMC on the binary GraphSK; returns (Es, chunks, num_moves, t)."""
    L = lib()
    L.orc_wtm_mc_skb.restype = C.c_int64
    L.orc_wtm_mc_skb.argtypes = [C.c_int64, u64p, C.c_double, C.c_int64, C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, u64p, f64p, i64p,
                                 C.POINTER(C.c_double)]
    N = Jb.shape[0]
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(samples, 1))
    stats = np.zeros(2, np.int64)
    t = C.c_double(0)
    n = L.orc_wtm_mc_skb(N, np.ascontiguousarray(Jb, np.uint64).reshape(-1), float(beta), int(samples), float(step), seed, call, replica,
                         ch, Es, stats, C.byref(t))
    if n < 0:
        raise RuntimeError("wtm_mc_skb: tracked energy != energy(X, C)")
    return Es[:n], ch, int(stats[0]), t.value


def extremal_opt_sk(J, tau, iters, step, seed, chunks, it0=0, replica=0, binary=False):
    """extremal_opt with EOCacheCont on GraphSKNormal (J dense Float64) or the binary GraphSK (J bit rows, binary=True).
    Returns (Es, final chunks, Emin, Cmin chunks, itmin)."""
    L = lib()
    fn = L.orc_extremal_opt_skb if binary else L.orc_extremal_opt_skn
    fn.restype = C.c_int64
    fn.argtypes = [C.c_int64, u64p if binary else f64p, f64p, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p,
                   C.POINTER(C.c_double), u64p, C.POINTER(C.c_int64)]
    N = J.shape[0]
    Jf = np.ascontiguousarray(J, np.uint64 if binary else np.float64).reshape(-1)
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1))
    Cmin = np.zeros_like(ch)
    Emin, itmin = C.c_double(0), C.c_int64(0)
    n = fn(N, Jf, eo_ftau(N, tau), int(iters), int(step), seed, it0, replica, ch, Es, C.byref(Emin), Cmin, C.byref(itmin))
    if n < 0:
        raise RuntimeError("extremal_opt_sk: inconsistent ranking / energy (%d)" % n)
    return Es[:n], ch, Emin.value, Cmin, itmin.value


def cont_quant(mode, A, J, M, fourK, beta, iters, step, seed, chunks, it0=0, call=0, replica=0, stepf=1.0, form="rrg"):
    """bklMC ("bkl") / wtmMC ("wtm") on GraphQuant over GraphRRG / GraphEA slices: the continuous-energy caches over the whole graph.
    Returns (Es, chunks, stats[3], t)."""
    L = lib()
    L.orc_cont_quant.restype = C.c_int64
    L.orc_cont_quant.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, i32p, i32p, C.c_double, C.c_double, C.c_int64, C.c_int64,
                                 C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u64p, f64p, i64p, C.POINTER(C.c_double)]
    A = np.ascontiguousarray(A, np.int32)
    Nk, K = A.shape
    ch = np.array(chunks, np.uint64, copy=True)
    m = {"bkl": 1, "wtm": 2}[mode]
    Es = np.zeros(max(iters if m == 2 else iters // step, 1))
    stats = np.zeros(3, np.int64)
    t = C.c_double(0)
    n = L.orc_cont_quant(m, 1 if form == "ea" else 0, Nk, int(M), K, A, np.ascontiguousarray(J, np.int32).reshape(-1), float(fourK), float(beta), int(iters), int(step),
                         float(stepf), seed, it0, call, replica, ch, Es, stats, C.byref(t))
    if n < 0:
        raise RuntimeError("cont_quant: DynamicSampler lost precision / unsupported (%d)" % n)
    return Es[:n], ch, stats, t.value


def _dense_quant_args(Jb, Jd):
    """(kind, Jb pointer or None, Jd pointer or None): 2 = binary GraphSK slices (bit-packed rows), 3 = GraphSKNormal (Nk x Nk Float64)."""
    if Jd is not None:
        a = np.ascontiguousarray(Jd, np.float64).reshape(-1)
        return 3, None, a.ctypes.data, a
    a = _jb(Jb)
    return 2, a.ctypes.data, None, a


def cont_quant_dense(mode, Nk, M, fourK, beta, iters, step, seed, chunks, Jb=None, Jd=None, it0=0, call=0, replica=0, stepf=1.0):
    """bklMC / wtmMC on a GraphQuant over dense slices (GraphQSKT: Jb; GraphQSKNormalT: Jd): every spin has the Trotter pair and the
    Nk - 1 other spins of its slice as neighbours.  Returns (Es, chunks, stats[3], t)."""
    L = lib()
    L.orc_cont_quant_dense.restype = C.c_int64
    L.orc_cont_quant_dense.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int64, C.c_int64,
                                       C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u64p, f64p, i64p, C.POINTER(C.c_double)]
    kind, pb, pd, keep = _dense_quant_args(Jb, Jd)
    ch = np.array(chunks, np.uint64, copy=True)
    m = {"bkl": 1, "wtm": 2}[mode]
    Es = np.zeros(max(iters if m == 2 else iters // step, 1))
    stats = np.zeros(3, np.int64)
    t = C.c_double(0)
    n = L.orc_cont_quant_dense(m, kind, int(Nk), int(M), pb, pd, float(fourK), float(beta), int(iters), int(step), float(stepf), seed, it0, call,
                               replica, ch, Es, stats, C.byref(t))
    if n < 0:
        raise RuntimeError("cont_quant_dense: DynamicSampler lost precision / unsupported (%d)" % n)
    return Es[:n], ch, stats, t.value


def extremal_opt_quant_dense(Nk, M, fourK, tau, iters, step, seed, chunks, Jb=None, Jd=None, it0=0, replica=0):
    """extremal_opt on a GraphQuant over dense slices.  Returns (Es, final chunks, Emin, Cmin chunks, itmin)."""
    L = lib()
    L.orc_extremal_opt_quant_dense.restype = C.c_int64
    L.orc_extremal_opt_quant_dense.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, f64p, C.c_int64, C.c_int64,
                                               C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, C.POINTER(C.c_double), u64p, C.POINTER(C.c_int64)]
    kind, pb, pd, keep = _dense_quant_args(Jb, Jd)
    N = int(Nk) * int(M)
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1))
    Cmin = np.zeros_like(ch)
    Emin, itmin = C.c_double(0), C.c_int64(0)
    n = L.orc_extremal_opt_quant_dense(kind, int(Nk), int(M), pb, pd, float(fourK), eo_ftau(N, tau), int(iters), int(step), seed, it0, replica,
                                       ch, Es, C.byref(Emin), Cmin, C.byref(itmin))
    if n < 0:
        raise RuntimeError("extremal_opt_quant_dense: inconsistent cache / energy (%d)" % n)
    return Es[:n], ch, Emin.value, Cmin, itmin.value


# ---- GraphQEAT = GraphQuant over sparse Float64 slices (src/QAliases.jl:50-83) ---------------------------------------------------------
def _spf_args(A, Jf):
    A = np.ascontiguousarray(A, np.int32)
    return A, np.ascontiguousarray(Jf, np.float64).reshape(-1), A.shape[0], A.shape[1]


def quant_spf_energy(A, Jf, M, fourK, chunks, form="ea"):
    L = lib()
    L.orc_quant_energy_spf.restype = C.c_double
    L.orc_quant_energy_spf.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, i32p, f64p, C.c_double, u64p]
    A, J, Nk, K = _spf_args(A, Jf)
    return float(L.orc_quant_energy_spf(1 if form == "ea" else 0, Nk, int(M), K, A, J, float(fourK), np.ascontiguousarray(chunks, np.uint64)))


def rrr_mc_quant_spf(A, Jf, M, fourK, beta, iters, step, seed, chunks, it0=0, replica=0, staged_thr=0.5, staged_thr_fact=5.0, want_cache=False, form="ea"):
    """One chain of rrrMC(X::DoubleGraph) on a GraphQEAT.  Returns (Es, chunks_out, accepted, staged_its[, pos, set_sizes])."""
    L = lib()
    L.orc_rrr_mc_quant_spf.restype = C.c_int64
    L.orc_rrr_mc_quant_spf.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, i32p, f64p, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_double,
                                       C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, i64p, C.c_void_p]
    A, J, Nk, K = _spf_args(A, Jf)
    N = Nk * int(M)
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    stats = np.zeros(2, np.int64)
    cache = np.zeros(N + 4, np.int32)
    n = L.orc_rrr_mc_quant_spf(1 if form == "ea" else 0, Nk, int(M), K, A, J, float(fourK), float(beta), int(iters), int(step), float(staged_thr),
                               float(staged_thr_fact), seed, it0, replica, ch, Es, stats, cache.ctypes.data if want_cache else None)
    if n < 0:
        raise AssertionError("DeltaECache / ArraySet consistency check failed")
    out = (Es[:n], ch, int(stats[0]), int(stats[1]))
    return out + (cache[:N].copy(), cache[N:].copy()) if want_cache else out


def standard_mc_quant_spf(A, Jf, M, fourK, beta, iters, step, seed, chunks, it0=0, replica=0, form="ea"):
    L = lib()
    L.orc_standard_mc_quant_spf.restype = C.c_int64
    L.orc_standard_mc_quant_spf.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, i32p, f64p, C.c_double, C.c_double, C.c_int64, C.c_int64,
                                            C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, i64p]
    A, J, Nk, K = _spf_args(A, Jf)
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1), np.float64)
    acc = np.zeros(1, np.int64)
    n = L.orc_standard_mc_quant_spf(1 if form == "ea" else 0, Nk, int(M), K, A, J, float(fourK), float(beta), int(iters), int(step), seed, it0, replica, ch, Es, acc)
    return Es[:n], ch, int(acc[0])


def cont_quant_spf(mode, A, Jf, M, fourK, beta, iters, step, seed, chunks, it0=0, call=0, replica=0, stepf=1.0, form="ea"):
    """bklMC ("bkl") / wtmMC ("wtm") on a GraphQEAT: the continuous-energy caches over the whole graph.  Returns (Es, chunks, stats[3], t)."""
    L = lib()
    L.orc_cont_quant_spf.restype = C.c_int64
    L.orc_cont_quant_spf.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, i32p, f64p, C.c_double, C.c_double, C.c_int64, C.c_int64,
                                     C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u64p, f64p, i64p, C.POINTER(C.c_double)]
    A, J, Nk, K = _spf_args(A, Jf)
    ch = np.array(chunks, np.uint64, copy=True)
    m = {"bkl": 1, "wtm": 2}[mode]
    Es = np.zeros(max(iters if m == 2 else iters // step, 1))
    stats = np.zeros(3, np.int64)
    t = C.c_double(0)
    n = L.orc_cont_quant_spf(m, 1 if form == "ea" else 0, Nk, int(M), K, A, J, float(fourK), float(beta), int(iters), int(step), float(stepf), seed, it0, call,
                             replica, ch, Es, stats, C.byref(t))
    if n < 0:
        raise RuntimeError("cont_quant_spf: DynamicSampler lost precision / unsupported (%d)" % n)
    return Es[:n], ch, stats, t.value


def extremal_opt_quant_spf(A, Jf, M, fourK, tau, iters, step, seed, chunks, it0=0, replica=0, form="ea"):
    """extremal_opt on a GraphQEAT (EOCacheCont over all Nk M spins).  Returns (Es, final chunks, Emin, Cmin chunks, itmin)."""
    L = lib()
    L.orc_extremal_opt_quant_spf.restype = C.c_int64
    L.orc_extremal_opt_quant_spf.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, i32p, f64p, C.c_double, f64p, C.c_int64, C.c_int64,
                                             C.c_uint64, C.c_uint64, C.c_uint32, u64p, f64p, C.POINTER(C.c_double), u64p, C.POINTER(C.c_int64)]
    A, J, Nk, K = _spf_args(A, Jf)
    N = Nk * int(M)
    ch = np.array(chunks, np.uint64, copy=True)
    Es = np.zeros(max(iters // step, 1))
    Cmin = np.zeros_like(ch)
    Emin, itmin = C.c_double(0), C.c_int64(0)
    n = L.orc_extremal_opt_quant_spf(1 if form == "ea" else 0, Nk, int(M), K, A, J, float(fourK), eo_ftau(N, tau), int(iters), int(step), seed, it0, replica,
                                     ch, Es, C.byref(Emin), Cmin, C.byref(itmin))
    if n < 0:
        raise RuntimeError("extremal_opt_quant_spf: inconsistent cache / energy (%d)" % n)
    return Es[:n], ch, Emin.value, Cmin, itmin.value
