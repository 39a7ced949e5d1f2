"""Sampler front-end: ``Engine`` owns a device context; ``standardMC`` mirrors src/RRRMC.jl:81-127."""
import ctypes as C

import numpy as np

from ._lib import RRRMCError, check, lib
from .graphs import DEFAULT_SEED, Config, GraphEA, nchunks

MODEL_SPARSE_PM1 = 1
MODEL_SK_NORMAL = 2
MODEL_QUANT_RRG = 3
MODEL_QUANT_SK, MODEL_QUANT_SKN, MODEL_QUANT_F64 = 8, 9, 10          # selectors of rrrmc_ctx_create_multi (GraphQuant over other slice families)
MODEL_SK_BINARY = 4
MODEL_SPARSE_F64 = 5
MODEL_SPARSE_DISCRETIZED = 6
MODEL_SPARSE_LEVELS = 7


class Engine:
    """R replicas of one graph on one MI355X — or, with ``devices=[...]``, on several from this one process: the library shards the
    replicas by global id over the listed devices (rrrmc_ctx_create_multi; one stream + host thread per device)
    and every call below is the same call, with gathered results."""

    def __init__(self, X, R=1, device=0, replica0=0, devices=None):
        self.X, self.R = X, int(R)
        self._ctx = C.c_void_p()
        self._f64 = X.model_kind not in (MODEL_SPARSE_PM1, MODEL_SPARSE_LEVELS)
        self._units = X.model_kind == MODEL_SPARSE_LEVELS        # device energies are int64 level units: X.energy_value converts
        if devices is not None:
            ids = np.asarray(list(devices), np.int32)
            quant = X.model_kind == MODEL_QUANT_RRG
            # a GraphQuant over dense slices is made per device by rrrmc_ctx_create_quant_skn / _sk: the selectors 9 / 8 of the header
            kind = (X.model_kind if not quant else MODEL_QUANT_SKN if getattr(X, "skn_slices", False) else MODEL_QUANT_SK if X.sk_slices
                    else MODEL_QUANT_F64 if getattr(X, "f64_slices", False) else X.model_kind)
            check(lib().rrrmc_ctx_create_multi(C.byref(self._ctx), kind, X.Nk if quant else X.N, X.K, X.M if quant else 0, self.R,
                                               ids, len(ids), replica0))
        elif X.model_kind == MODEL_QUANT_RRG and getattr(X, "skn_slices", False):
            check(lib().rrrmc_ctx_create_quant_skn(C.byref(self._ctx), X.Nk, X.M, self.R, device, replica0))
        elif X.model_kind == MODEL_QUANT_RRG and X.sk_slices:
            check(lib().rrrmc_ctx_create_quant_sk(C.byref(self._ctx), X.Nk, X.M, self.R, device, replica0))
        elif X.model_kind == MODEL_QUANT_RRG and getattr(X, "f64_slices", False):
            check(lib().rrrmc_ctx_create_quant_f64(C.byref(self._ctx), X.Nk, X.K, X.M, self.R, device, replica0))
        elif X.model_kind == MODEL_QUANT_RRG:
            check(lib().rrrmc_ctx_create_quant(C.byref(self._ctx), X.Nk, X.K, X.M, self.R, device, replica0))
        else:
            check(lib().rrrmc_ctx_create(C.byref(self._ctx), X.model_kind, X.N, X.K, self.R, device, replica0))
        try:
            if X.model_kind == MODEL_QUANT_RRG:
                if getattr(X, "skn_slices", False):
                    check(lib().rrrmc_set_couplings_dense(self._ctx, X.J.reshape(-1)), self._ctx)
                elif X.sk_slices:
                    check(lib().rrrmc_set_couplings_bits(self._ctx, X.J.reshape(-1)), self._ctx)
                elif getattr(X, "f64_slices", False):
                    check(lib().rrrmc_set_graph_f64(self._ctx, X.A, X.J.reshape(-1)), self._ctx)
                else:
                    check(lib().rrrmc_quant_slice_form(self._ctx, 1 if isinstance(X.X1, GraphEA) else 0), self._ctx)
                    check(lib().rrrmc_set_graph(self._ctx, X.A, X.J), self._ctx)
                check(lib().rrrmc_quant_set_field(self._ctx, X.beta, X.fourK), self._ctx)
            elif X.model_kind == MODEL_SPARSE_DISCRETIZED:
                check(lib().rrrmc_set_graph_discretized(self._ctx, X.A, X.dJ, X.rJ.reshape(-1), np.asarray(X.LEV, np.int32), len(X.LEV),
                                                        X.ea_form), self._ctx)
                check(lib().rrrmc_set_level_scale(self._ctx, X.lev_mul, X.lev_div), self._ctx)
            elif X.model_kind == MODEL_SPARSE_LEVELS:
                check(lib().rrrmc_set_graph_levels(self._ctx, X.A, X.J, np.asarray(X.LEV, np.int32), len(X.LEV), X.ea_form), self._ctx)
                check(lib().rrrmc_set_level_scale(self._ctx, X.lev_mul, X.lev_div), self._ctx)
            elif X.model_kind == MODEL_SPARSE_F64:
                check(lib().rrrmc_set_graph_f64(self._ctx, X.A, X.J.reshape(-1)), self._ctx)
            elif X.model_kind == MODEL_SK_BINARY:
                check(lib().rrrmc_set_couplings_bits(self._ctx, X.J.reshape(-1)), self._ctx)
            elif self._f64:
                check(lib().rrrmc_set_couplings_dense(self._ctx, X.J.reshape(-1)), self._ctx)
            else:
                check(lib().rrrmc_set_graph(self._ctx, X.A, X.J), self._ctx)
        except RRRMCError:
            self.close()
            raise

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            lib().rrrmc_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- state ------------------------------------------------------------------------------------
    def seed(self, seed):
        check(lib().rrrmc_seed(self._ctx, int(seed) & (2 ** 64 - 1)), self._ctx)

    def iterations_done(self):
        return int(lib().rrrmc_iterations_done(self._ctx))

    def init_spins_random(self):
        check(lib().rrrmc_init_spins_random(self._ctx), self._ctx)

    def set_config(self, Cfg):
        if Cfg.N != self.X.N:
            raise ValueError("Invalid C0, wrong N, expected %d, given: %d" % (self.X.N, Cfg.N))   # RRRMC.jl:94
        if Cfg.R != self.R:
            raise ValueError("Invalid C0, wrong number of replicas, expected %d, given: %d" % (self.R, Cfg.R))
        check(lib().rrrmc_set_spins(self._ctx, Cfg.s), self._ctx)

    def get_config(self, out=None):
        if out is None:
            out = Config(self.X.N, self.R)
        check(lib().rrrmc_get_spins(self._ctx, out.s), self._ctx)
        return out

    def energy(self):
        E = np.zeros(self.R, np.float64 if self._f64 else np.int64)
        check((lib().rrrmc_energy_f64 if self._f64 else lib().rrrmc_energy)(self._ctx, E), self._ctx)
        return self.X.energy_value(E) if self._units else E.astype(self.X.energy_dtype, copy=False)

    def fields(self):
        f64 = self.X.model_kind in (MODEL_SK_NORMAL, MODEL_SPARSE_F64)        # GraphSK's cache is integer (SK.jl:33)
        lf = np.zeros((self.R, self.X.N), np.float64 if f64 else np.int64)
        check((lib().rrrmc_get_fields_f64 if f64 else lib().rrrmc_get_fields)(self._ctx, lf.reshape(-1)), self._ctx)
        return lf

    # -- sampling ---------------------------------------------------------------------------------
    def standard_mc(self, beta, iters, step=1, want_energies=True, out=None):
        """Returns (Es[R, iters // step], accepted[R]).  ``out=(Es, accepted)``: caller-owned result arrays of those shapes (e.g. from
        ``pinned_empty``: page-locked memory is filled at the bus rate), reused across calls like the buffers of a C or Julia caller."""
        nsamp = int(iters) // int(step)
        if out is not None:
            Es, acc = out
            if Es.shape != (self.R, nsamp) or Es.dtype != (np.float64 if self._f64 else np.int64) or acc.shape != (self.R,) or acc.dtype != np.int64:
                raise ValueError("out must be (Es[R, iters // step] of the model's energy type, accepted[R] int64)")
            # the library writes R * nsamp contiguous values behind the pointer: a strided view of a larger buffer would be filled in the wrong
            # layout (and its neighbours overwritten), a read-only array must not be written at all
            if not (Es.flags.c_contiguous and acc.flags.c_contiguous and Es.flags.writeable and acc.flags.writeable):
                raise ValueError("out arrays must be C-contiguous and writeable (slice a pinned buffer along its first axis, not its second)")
        else:
            Es = np.zeros((self.R, nsamp), np.float64 if self._f64 else np.int64)
            acc = np.zeros(self.R, np.int64)
        fn = lib().rrrmc_standard_mc_f64 if self._f64 else lib().rrrmc_standard_mc
        check(fn(self._ctx, float(beta), int(iters), int(step), Es.ctypes.data if (want_energies and nsamp) else None,
                 acc.ctypes.data), self._ctx)
        return (self.X.energy_value(Es) if self._units else Es), acc

    def set_resume(self, on=True):
        """Make the following standardMC calls continue the previous one (tracked energy, cache) instead of starting like a fresh
        reference call: a run cut into pieces is then bit for bit the run made in one call (Float64 models; a no-op for integer ones)."""
        check(lib().rrrmc_set_resume(self._ctx, 1 if on else 0), self._ctx)

    def set_debug_checks(self, on=True):
        """After every standardMC call the library re-runs ``energy`` on the device and compares it (and, for GraphSKNormal, the cached
        local fields) with what the sampler tracked — the reference's own commented-out asserts as a switch; a mismatch surfaces as an
        RRRMCError (status 2) at the next sync."""
        check(lib().rrrmc_set_debug_checks(self._ctx, 1 if on else 0), self._ctx)

    def tracked_energy(self):
        """The energy the sampler tracks (E += dE per accepted move, src/RRRMC.jl:117) after the last standardMC call."""
        if not self._f64:
            return self.energy()
        E = np.zeros(self.R, np.float64)
        check(lib().rrrmc_tracked_energy_f64(self._ctx, E), self._ctx)
        return E

    def run_energy(self):
        """The energy the last sampler call tracked (``E += ΔE`` per accepted move: what the reference hands to its hooks), read without
        disturbing the run a resumed call continues (``energy()`` rebuilds caches, like the reference's ``energy(X, C)``)."""
        if self._f64:
            E = np.zeros(self.R, np.float64)
            check(lib().rrrmc_tracked_energy_f64(self._ctx, E), self._ctx)
            return E
        E = np.zeros(self.R, np.int64)
        check(lib().rrrmc_tracked_energy(self._ctx, E), self._ctx)
        return self.X.energy_value(E) if self._units else E

    def standard_mc_async(self, beta, iters, step=1):
        check(lib().rrrmc_standard_mc_async(self._ctx, float(beta), int(iters), int(step)), self._ctx)
        self._last = (int(iters), int(step))

    def standard_mc_fast_async(self, beta, iters, step=1):
        """Opt-in fast standardMC for GraphRRGNormal / GraphEANormal (bit-sliced replicas, per-site threshold tables): a faithful
        chain, not the bit-exact default one (see include/rrrmc_hip.h).  Then ``sync()`` and ``fetch_results()``."""
        check(lib().rrrmc_standard_mc_fast_async(self._ctx, float(beta), int(iters), int(step)), self._ctx)
        self._last = (int(iters), int(step))

    def standard_mc_fast(self, beta, iters, step=1):
        self.standard_mc_fast_async(beta, iters, step)
        self.sync()
        return self.fetch_results()

    def sync(self):
        check(lib().rrrmc_sync(self._ctx), self._ctx)

    def fetch_results(self, want_energies=True):
        nsamp = int(lib().rrrmc_results_samples(self._ctx))       # iters // step, or what a resumed call took (rrrmc_set_resume)
        Es = np.zeros((self.R, nsamp), np.float64 if self._f64 else np.int64)
        acc = np.zeros(self.R, np.int64)
        fn = lib().rrrmc_fetch_results_f64 if self._f64 else lib().rrrmc_fetch_results
        check(fn(self._ctx, Es.ctypes.data if (want_energies and nsamp) else None, acc.ctypes.data), self._ctx)
        return (self.X.energy_value(Es) if self._units else Es), acc

    # -- colour-parallel sweeps (build-defined checkerboard sampler for large sparse graphs) ---------
    def set_coloring(self, color):
        color = np.ascontiguousarray(color, np.int32)
        check(lib().rrrmc_set_coloring(self._ctx, color, int(color.max()) + 1), self._ctx)

    def colored_count_accepted(self, on=True):
        """Make the colour sweeps count every replica's accepted moves (off by default: the counters then read -1)."""
        check(lib().rrrmc_colored_count_accepted(self._ctx, 1 if on else 0), self._ctx)

    def colored_sweeps_async(self, beta, sweeps, step=1):
        check(lib().rrrmc_colored_sweeps_async(self._ctx, float(beta), int(sweeps), int(step)), self._ctx)
        self._last = (int(sweeps), int(step))

    def colored_sweeps(self, beta, sweeps, step=1):
        """`sweeps` colour-parallel sweeps; returns Es[R, sweeps // step] (energy before sweep k*step)."""
        self.colored_sweeps_async(beta, sweeps, step)
        self.sync()
        return self.fetch_results()[0]

    # -- reduced-rejection-rate sampler (GraphQuant) ------------------------------------------------
    def rrr_mc(self, beta, iters, step=1, staged_thr=None, staged_thr_fact=5.0, want_energies=True):
        """rrrMC(X::DoubleGraph, β, iters; step, staged_thr, staged_thr_fact) (src/RRRMC.jl:221-290).
        Returns (Es[R, iters // step], accepted[R], staged_iters[R])."""
        if staged_thr is None:          # RRRMC.jl:162-164 (SimpleGraph 0.8, DiscrGraph 0.5) and :226 (DoubleGraph 0.5)
            staged_thr = 0.8 if self.X.model_kind in (MODEL_SK_NORMAL, MODEL_SK_BINARY, MODEL_SPARSE_F64) else 0.5
        check(lib().rrrmc_rrr_mc_async(self._ctx, float(beta), float(getattr(self.X, "fourK", 0.0)), int(iters), int(step),
                                       float(staged_thr), float(staged_thr_fact)), self._ctx)
        self._last = (int(iters), int(step))
        self.sync()
        Es, acc = self.fetch_results(want_energies)
        st = np.zeros(self.R, np.int64)
        check(lib().rrrmc_rrr_stats(self._ctx, st), self._ctx)
        return Es, acc, st

    def bkl_mc(self, beta, iters, step=1):
        """bklMC(X, β, iters; step) (src/RRRMC.jl:311-359).  Returns (Es[R, iters // step], accepted = true moves [R])."""
        check(lib().rrrmc_bkl_mc_async(self._ctx, float(beta), int(iters), int(step)), self._ctx)
        self._last = (int(iters), int(step))
        self.sync()
        return self.fetch_results()

    def wtm_mc(self, beta, samples, step=1.0):
        """wtmMC(X, β, samples; step) (src/RRRMC.jl:376-426).  Returns (Es[R, samples], num_moves[R], global_time[R])."""
        check(lib().rrrmc_wtm_mc_async(self._ctx, float(beta), int(samples), float(step)), self._ctx)
        self._last = (int(samples), 1)
        self.sync()
        Es, moves = self.fetch_results()
        t = np.zeros(self.R)
        check(lib().rrrmc_wtm_times(self._ctx, t), self._ctx)
        return Es, moves, t

    def extremal_opt(self, tau, iters, step=1):
        """extremal_opt(X, τ, iters; step) (src/RRRMC.jl:474-521).  Returns (Es[R, iters // step], Emin[R], Cmin: Config, itmin[R])."""
        N = self.X.N
        ftau = np.cumsum(np.arange(1, N + 1, dtype=np.float64) ** (-float(tau)))          # DeltaE.jl:444
        check(lib().rrrmc_extremal_opt_async(self._ctx, ftau, int(iters), int(step)), self._ctx)
        self._last = (int(iters), int(step))
        self.sync()
        Es, _ = self.fetch_results()
        Emin = np.zeros(self.R, np.float64 if self._f64 else np.int64)
        itmin = np.zeros(self.R, np.int64)
        Cmin = Config(N, self.R)
        res = lib().rrrmc_extremal_opt_results_f64 if self._f64 else lib().rrrmc_extremal_opt_results     # EOCacheCont / EOCache
        check(res(self._ctx, Emin, Cmin.s.reshape(-1), itmin), self._ctx)
        return Es, (self.X.energy_value(Emin) if self._units else Emin), Cmin, itmin

    def rrr_cache(self):
        """(pos[R, N], sizes[R, 4]) of the DeltaECache after the last rrrMC call (GraphQuant); sizes[R, 16] for the DiscrGraphs
        (GraphRRG / GraphEA, rrrMC and bklMC) and the discretised DoubleGraphs, class k of replica r at [r, k]."""
        pos = np.zeros((self.R, self.X.N), np.int8)
        sizes = np.zeros((self.R, 4 if self.X.model_kind == MODEL_QUANT_RRG else 16), np.int32)
        check(lib().rrrmc_rrr_cache(self._ctx, pos.ctypes.data, sizes.ctypes.data), self._ctx)
        return pos, sizes

    # -- snapshots and observables ---------------------------------------------
    def snapshot_reserve(self, nslots):
        check(lib().rrrmc_snapshot_reserve(self._ctx, int(nslots)), self._ctx)

    def snapshot_store(self, slot):
        """Keep a device-side copy of the live configuration (what the reference's hooks do with ``copy(C.s)``)."""
        check(lib().rrrmc_snapshot_store(self._ctx, int(slot)), self._ctx)

    def snapshot_get(self, slot, out=None):
        if out is None:
            out = Config(self.X.N, self.R)
        check(lib().rrrmc_snapshot_get(self._ctx, int(slot), out.s), self._ctx)
        return out

    def overlaps(self, slotA, slotB):
        """q[p, r] = pm1dot(replica r of slotA[p], replica r of slotB[p]) (scripts/scripts.jl:283-295); slot -1 = live."""
        a = np.ascontiguousarray(np.atleast_1d(slotA), np.int32)
        b = np.ascontiguousarray(np.atleast_1d(slotB), np.int32)
        if a.shape != b.shape:
            raise ValueError("slotA and slotB must have the same length")
        q = np.zeros((a.size, self.R), np.int32)
        check(lib().rrrmc_overlaps(self._ctx, a.size, a, b, q.reshape(-1)), self._ctx)
        return q

    def quant_observables(self, beta=None, Gamma=None):
        """(Qenergy[R], transverse_mag[R], overlaps[R, M // 2]) of a GraphQuant (src/graphs/QT.jl:113-122, 213-268)."""
        X = self.X
        Q = np.zeros(self.R)
        tm = np.zeros(self.R)
        ov = np.zeros((self.R, X.M // 2))
        check(lib().rrrmc_quant_observables(self._ctx, float(X.beta if beta is None else beta), float(X.Gamma if Gamma is None else Gamma),
                                            Q.ctypes.data, tm.ctypes.data, ov.ctypes.data), self._ctx)
        return Q, tm, ov

    def spf_team_build(self):
        """(waves, width, slots) of the spf_team_kernel build the default standardMC of a GraphRRGNormal / GraphEANormal launches here;
        zeros when the one-wavefront kernel runs instead."""
        nw, tw, m = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(lib().rrrmc_spf_team_build(self._ctx, C.byref(nw), C.byref(tw), C.byref(m)), self._ctx)
        return nw.value, tw.value, m.value

    def timing_accumulate(self, on=True, reserve_launches=1):
        """Give every sweep launch of the following async calls its own HIP-event pair (see ``timing_total``); the events of the
        first ``reserve_launches`` launches are created now, so that the calls in between create none."""
        check(lib().rrrmc_timing_accumulate(self._ctx, max(1, int(reserve_launches)) if on else 0), self._ctx)

    def timing_total(self):
        """(sweep_ms, sweep_launches) summed over every sampling call since ``timing_accumulate()``; synchronises."""
        s, n = C.c_double(0), C.c_int64(0)
        check(lib().rrrmc_timing_total(self._ctx, C.byref(s), C.byref(n)), self._ctx)
        return s.value, n.value

    def last_timing(self):
        """(total_ms, sweep_ms, sweep_launches) of the last sampling call, from HIP events on the ctx's stream."""
        t, s, n = C.c_double(0), C.c_double(0), C.c_int32(0)
        check(lib().rrrmc_last_timing(self._ctx, C.byref(t), C.byref(s), C.byref(n)), self._ctx)
        return t.value, s.value, n.value


class EnergyProbe:
    """``energy(X, C)`` (src/Interface.jl:105) for use INSIDE a hook: the reference's test hook checks ``E ≈ energy(X, C)`` at every sample
    (test/runtests.jl:12-20).  The sampler's own context must not be asked — ``rrrmc_energy`` rebuilds caches and ends the run a resumed call
    continues — so the probe keeps a context of its own for the same graph."""

    def __init__(self, X, R=1, device=0):
        self.eng = Engine(X, R, device=device)

    def __call__(self, C):
        self.eng.set_config(C)
        return self.eng.energy()

    def close(self):
        self.eng.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def energy(X, C, device=0):
    """``energy(X, C)`` of every replica of ``C`` on a context made for the call (see ``EnergyProbe`` for repeated use)."""
    with EnergyProbe(X, C.R, device) as probe:
        return probe(C)


class _HookRun:
    """Book-keeping of a hooked run over R replicas.  The reference's hook ends ONE chain (``hook(...) || break``, src/RRRMC.jl:107,187,256,
    341,404,501); here a hook may return one flag for all replicas or one per replica: a replica whose flag is False is FROZEN at that sample —
    its configuration, samples and counts are what the reference's chain would return, the hook keeps seeing them — while the others go on
    (replicas are independent); the run ends when none is left."""

    def __init__(self, X, eng, Cfg):
        self.X, self.eng, self.Cfg = X, eng, Cfg
        R = eng.R
        self.frozen = np.zeros(R, bool)
        self.cfg = Config(X.N, R)
        self.nsamp = np.zeros(R, np.int64)
        self.vals = {}                      # name -> frozen per-replica values (accepted counts, energies, ...)
        self.samples = []

    def view(self, name, live):
        """``live`` with the frozen replicas' values of that moment"""
        live = np.asarray(live)
        if name not in self.vals:
            self.vals[name] = np.zeros_like(live)
        return np.where(self.frozen, self.vals[name], live) if self.frozen.any() else live

    def call(self, hook, it, E, args, live):
        """Record the sample ``E``, fetch the configuration and call ``hook(it, X, C, *args)``; ``live`` = {name: per-replica array} of
        what must be kept for a replica that stops here.  Returns False when the run is over."""
        self.samples.append(np.array(E))
        self.eng.get_config(self.Cfg)
        if self.frozen.any():
            self.Cfg.s[self.frozen] = self.cfg.s[self.frozen]
        go = hook(it, self.X, self.Cfg, *args)
        if np.ndim(go) == 0:
            return bool(go)
        go = np.asarray(go, bool)
        if go.shape != (self.eng.R,):
            raise ValueError("a hook returns one flag, or one per replica (%d)" % self.eng.R)
        new = ~go & ~self.frozen
        self.cfg.s[new] = self.Cfg.s[new]
        self.nsamp[new] = len(self.samples)
        for name, v in live.items():
            v = np.asarray(v)
            if name not in self.vals:
                self.vals[name] = np.zeros_like(v)
            self.vals[name][new] = v[new]
        self.frozen |= new
        return not self.frozen.all()

    def finish(self, dtype):
        """(Es, C): the engine is given the frozen replicas' configurations back, so that C and the device agree"""
        Es = np.stack(self.samples, axis=1) if self.samples else np.zeros((self.eng.R, 0), dtype)
        self.eng.get_config(self.Cfg)
        if self.frozen.any():
            self.Cfg.s[self.frozen] = self.cfg.s[self.frozen]
            self.eng.set_config(self.Cfg)
            Es = [Es[r, :self.nsamp[r]] if self.frozen[r] else Es[r] for r in range(self.eng.R)]
        return Es, self.Cfg


def _setup(X, seed, C0, replicas, device, replica0, engine):
    own = engine is None
    R = replicas if replicas is not None else (C0.R if C0 is not None else (engine.R if engine else 1))
    if C0 is not None and C0.N != X.N:
        raise ValueError("Invalid C0, wrong N, expected %d, given: %d" % (X.N, C0.N))       # RRRMC.jl:172,234,320,385,482
    eng = engine if engine is not None else Engine(X, R, device=device, replica0=replica0)
    try:
        if seed > 0 or own:
            eng.seed(seed if seed > 0 else 0)
        if C0 is not None:
            eng.set_config(C0)
        elif own:
            eng.init_spins_random()
    except Exception:
        if own:
            eng.close()
        raise
    return own, eng, (C0 if C0 is not None else Config(X.N, eng.R))


def _nsamples(Es):
    return Es.shape[1] if hasattr(Es, "shape") else [len(e) for e in Es]


def rrrMC(X, beta, iters, *, seed=DEFAULT_SEED, step=1, hook=None, C0=None, staged_thr=None, staged_thr_fact=5.0, quiet=False,
          replicas=None, device=0, replica0=0, engine=None):
    """``rrrMC(X, β, iters; seed, step, hook, C0, staged_thr, staged_thr_fact, quiet)`` (src/RRRMC.jl:149-219 for a SingleGraph, :221-290 for a
    DoubleGraph) for a batch of replicas.  Returns ``(Es, C)`` like ``standardMC``.
    ``hook(it, X, C, accepted, E)`` (:186,255) is called every ``step`` iterations — before the move of iteration ``it``, as the reference does
    (:184-188) — with per-replica arrays; returning False stops the run (one flag per replica: see ``_HookRun``).  The run is cut at the hook
    points and the pieces RESUME one another (``rrrmc_set_resume``): the DeltaECache, E and the acceptance-rate average live on, so a hooked
    run is the un-hooked chain bit for bit."""
    import math
    if not math.isfinite(beta):
        raise ValueError("β must be finite, given: %r" % beta)                     # RRRMC.jl:166,230
    own, eng, Cfg = _setup(X, seed, C0, replicas, device, replica0, engine)
    try:
        iters, step = int(iters), int(step)
        if hook is None:
            Es, acc, staged = eng.rrr_mc(beta, iters, step, staged_thr, staged_thr_fact)
            eng.get_config(Cfg)
            it = iters
        else:
            run = _HookRun(X, eng, Cfg)
            acc = np.zeros(eng.R, np.int64)
            staged = np.zeros(eng.R, np.int64)

            def piece(n):
                _, a, st = eng.rrr_mc(beta, n, step, staged_thr, staged_thr_fact, want_energies=False)
                acc[:] += a
                staged[:] += st

            eng.set_resume(False)
            piece(min(step - 1, iters))             # a fresh run (energy(X, C), gen_ΔEcache: :177-178), up to just before the first sampled iteration
            it = min(step - 1, iters)
            eng.set_resume(True)
            try:
                while it + 1 <= iters:
                    E = eng.run_energy()
                    if not run.call(hook, it + 1, run.view("E", E), (run.view("acc", acc), run.view("E", E)), {"acc": acc, "E": E}):
                        it += 1                     # the reference has counted the iteration its hook ended (:183)
                        break
                    n = min(step, iters - it)       # the move of the sampled iteration and the step - 1 after it
                    piece(n)
                    it += n
            finally:
                eng.set_resume(False)
            Es, Cfg = run.finish(X.energy_dtype)
            acc = run.view("acc", acc)
        if not quiet:
            print("samples = ", _nsamples(Es))
            print("iters = ", it)
            print("accept rate = ", float(acc.mean()) / max(it, 1))
            print("frac. staged iters = ", float(staged.mean()) / max(it, 1))
        return Es, Cfg
    finally:
        if own:
            eng.close()


def bklMC(X, beta, iters, *, seed=DEFAULT_SEED, step=1, hook=None, C0=None, quiet=False, replicas=None, device=0, replica0=0, engine=None):
    """``bklMC(X, β, iters; seed, step, hook, C0, quiet)`` (src/RRRMC.jl:311-359) for a batch of replicas.  ``hook(nextstep, X, C, accepted, E)``
    (:341) is called at every sample point the skipped iterations pass; the run is cut there and resumed (``it``, ``nextstep`` and the pending
    (skip, move) draw carry on): the hooked run is the un-hooked chain bit for bit."""
    own, eng, Cfg = _setup(X, seed, C0, replicas, device, replica0, engine)
    try:
        iters, step = int(iters), int(step)
        nsamp = iters // step
        if hook is None or nsamp == 0:
            Es, moves = eng.bkl_mc(beta, iters, step)
            eng.get_config(Cfg)
        else:
            run = _HookRun(X, eng, Cfg)
            moves = np.zeros(eng.R, np.int64)
            eng.set_resume(False)
            try:
                for k in range(1, nsamp + 1):       # a call of `step` iterations ends at the next sample point (the reference goes `out` after the last one, :343)
                    _, m = eng.bkl_mc(beta, step, step)
                    eng.set_resume(True)
                    moves += m
                    E = eng.run_energy()
                    if not run.call(hook, k * step, run.view("E", E), (run.view("acc", moves), run.view("E", E)), {"acc": moves, "E": E}):
                        break
            finally:
                eng.set_resume(False)
            Es, Cfg = run.finish(X.energy_dtype)
            moves = run.view("acc", moves)
        if not quiet:
            print("samples = ", _nsamples(Es))
            print("accept rate = ", float(moves.mean()) / max(iters, 1))
            print("true it = ", float(moves.mean()))
        return Es, Cfg
    finally:
        if own:
            eng.close()


def wtmMC(X, beta, samples, *, seed=DEFAULT_SEED, step=1.0, hook=None, C0=None, quiet=False, replicas=None, device=0, replica0=0, engine=None):
    """``wtmMC(X, β, samples; seed, step, hook, C0, quiet)`` (src/RRRMC.jl:376-426) for a batch of replicas.  ``hook(nextstep, X, C, num_moves, E)``
    (:404) is called at every sample time (``nextstep`` = the global time of the sample, k additions of step / N as :391,405 accumulate it);
    the run is cut there and resumed (the heap of waiting times, the global time and ``nextstep`` carry on)."""
    own, eng, Cfg = _setup(X, seed, C0, replicas, device, replica0, engine)
    try:
        samples = int(samples)
        if hook is None:
            Es, moves, t = eng.wtm_mc(beta, samples, step)
            eng.get_config(Cfg)
        else:
            run = _HookRun(X, eng, Cfg)
            moves = np.zeros(eng.R, np.int64)
            t = np.zeros(eng.R)
            st = float(step) / X.N                  # :391
            nextstep = st
            eng.set_resume(False)
            try:
                for _ in range(samples):
                    _, m, t = eng.wtm_mc(beta, 1, step)
                    eng.set_resume(True)
                    moves += m
                    E = eng.run_energy()
                    if not run.call(hook, nextstep, run.view("E", E), (run.view("acc", moves), run.view("E", E)), {"acc": moves, "E": E, "t": t}):
                        break
                    nextstep += st                  # :405
            finally:
                eng.set_resume(False)
            Es, Cfg = run.finish(X.energy_dtype)
            moves, t = run.view("acc", moves), run.view("t", t)
        if not quiet:
            print("samples = ", _nsamples(Es))
            print("num_moves = ", float(moves.mean()))
            print("global time = ", float(t.mean()))
            print("ratio = ", float(t.mean()) / max(float(moves.mean()), 1.0))
        return Es, Cfg
    finally:
        if own:
            eng.close()


def extremal_opt(X, tau, iters, *, seed=DEFAULT_SEED, step=1, hook=None, C0=None, quiet=False, replicas=None, device=0, replica0=0, engine=None):
    """``extremal_opt(X, τ, iters; seed, step, hook, C0, quiet)`` (src/RRRMC.jl:474-521) for a batch of replicas of a GraphRRG / GraphEA (EOCache) or of
    any other graph (the generic EOCacheCont).  Returns ``(C, Emin, Cmin, itmin)`` like the reference, with per-replica arrays.
    ``hook(it, X, C, E, Emin)`` (:501 — note the signature) is called every ``step`` iterations, before the move of iteration ``it``; the run is
    cut there and resumed (the ranking, E, Emin / Cmin / itmin carry on)."""
    own, eng, Cfg = _setup(X, seed, C0, replicas, device, replica0, engine)
    try:
        iters, step = int(iters), int(step)
        if hook is None:
            _, Emin, Cmin, itmin = eng.extremal_opt(tau, iters, step)
            eng.get_config(Cfg)
            it = iters
        else:
            run = _HookRun(X, eng, Cfg)
            eng.set_resume(False)
            _, Emin, Cmin, itmin = eng.extremal_opt(tau, min(step - 1, iters), step)
            it = min(step - 1, iters)
            eng.set_resume(True)
            try:
                while it + 1 <= iters:
                    E = eng.run_energy()
                    live = {"E": E, "Emin": Emin, "itmin": itmin, "Cmin": Cmin.s}
                    if not run.call(hook, it + 1, run.view("E", E), (run.view("E", E), run.view("Emin", Emin)), live):
                        it += 1
                        break
                    n = min(step, iters - it)
                    _, Emin, Cmin, itmin = eng.extremal_opt(tau, n, step)
                    it += n
            finally:
                eng.set_resume(False)
            _, Cfg = run.finish(X.energy_dtype)
            Emin, itmin = run.view("Emin", Emin), run.view("itmin", itmin)
            if run.frozen.any():
                Cmin.s[run.frozen] = run.vals["Cmin"][run.frozen]
        if not quiet:
            print("iters = ", it)
            print("min [it = %s] = %s" % (np.asarray(itmin).tolist(), np.asarray(Emin).tolist()))
        return Cfg, Emin, Cmin, itmin
    finally:
        if own:
            eng.close()


def standardMC(X, beta, iters, *, seed=DEFAULT_SEED, step=1, hook=None, C0=None, quiet=False,
               replicas=None, device=0, replica0=0, engine=None):
    """``standardMC(X, β, iters; seed, step, hook, C0, quiet)`` for a batch of replicas (src/RRRMC.jl:81-127).

    Returns ``(Es, C)``: ``Es[r]`` is replica r's vector of energies (one per ``step`` iterations, taken
    before the move of iteration k*step, RRRMC.jl:104-108) and ``C`` the final ``Config``.  As in the
    reference ``C0`` is resumed and mutated in place, and ``seed <= 0`` does not reseed (meaningful with a
    caller-supplied ``engine``, which carries the stream position).  ``hook(it, X, C, accepted, E)`` is
    called every ``step`` iterations with per-replica arrays; returning False stops the run (:61-64).
    The reference's hook ends ONE chain (``hook(...) || break``, :107): a hook may therefore also return one flag per replica — a replica
    whose flag is False is frozen at that sample (its configuration, energy samples and accepted count are what the reference's chain would
    return; the others go on, replicas being independent), and the run ends when none is left.  With frozen replicas ``Es`` is a list of
    per-replica vectors of different lengths, as R reference calls would return them.
    """
    own = engine is None
    R = replicas if replicas is not None else (C0.R if C0 is not None else (engine.R if engine else 1))
    if C0 is not None and C0.N != X.N:
        raise ValueError("Invalid C0, wrong N, expected %d, given: %d" % (X.N, C0.N))   # RRRMC.jl:94
    eng = engine if engine is not None else Engine(X, R, device=device, replica0=replica0)
    try:
        if seed > 0 or own:
            eng.seed(seed if seed > 0 else 0)
        if C0 is not None:
            eng.set_config(C0)
        elif own:
            eng.init_spins_random()
        Cfg = C0 if C0 is not None else Config(X.N, eng.R)
        accepted = np.zeros(eng.R, np.int64)
        it = 0
        if hook is None:
            Es, accepted = eng.standard_mc(beta, iters, step)
            it = int(iters)
        else:
            samples = []
            frozen = np.zeros(eng.R, bool)                 # replicas whose hook has said "stop" (RRRMC.jl:107), with what they had then
            frozen_cfg = Config(X.N, eng.R)
            frozen_nsamp = np.zeros(eng.R, np.int64)
            frozen_acc = np.zeros(eng.R, np.int64)
            # Run up to the move before iteration k*step, call the hook with that state, continue.  The pieces RESUME one another
            # (Engine.set_resume): the cache and the tracked energy live on across hook calls as in the reference (RRRMC.jl:95-118), so
            # the hooked run of a Float64 model is the un-hooked chain bit for bit and the hook sees the tracked E.
            if it < iters and eng._f64:
                eng.standard_mc(beta, 0, step=1, want_energies=False)      # E = energy(X, C), fresh cache: the start of a reference call
            eng.set_resume(True)
            try:        # whatever the hook (or a call) raises, a caller-supplied engine must not stay in resume mode
                while it < iters:
                    nxt = (it // step + 1) * step          # next sampled iteration
                    n = min(nxt - 1, iters) - it
                    if n > 0:
                        _, a = eng.standard_mc(beta, n, step=n + 1, want_energies=False)
                        accepted += a
                        it += n
                    if nxt > iters:
                        break
                    E = eng.tracked_energy()
                    samples.append(E)
                    eng.get_config(Cfg)
                    # a frozen replica's chain has ended for the caller: the hook keeps seeing the count and configuration it had then
                    if frozen.any():
                        Cfg.s[frozen] = frozen_cfg.s[frozen]
                    go = hook(nxt, X, Cfg, np.where(frozen, frozen_acc, accepted), E)
                    if np.ndim(go) == 0:
                        if not go:
                            it = nxt
                            break
                    else:
                        go = np.asarray(go, bool)
                        if go.shape != (eng.R,):
                            raise ValueError("a hook returns one flag, or one per replica (%d)" % eng.R)
                        new = ~go & ~frozen
                        frozen_cfg.s[new] = Cfg.s[new]
                        frozen_nsamp[new] = len(samples)
                        frozen_acc[new] = accepted[new]
                        frozen |= new
                        if frozen.all():
                            it = nxt
                            break
                    _, a = eng.standard_mc(beta, 1, step=2, want_energies=False)   # the move of iteration nxt
                    accepted += a
                    it = nxt
            finally:
                eng.set_resume(False)
            Es = np.stack(samples, axis=1) if samples else np.zeros((eng.R, 0), X.energy_dtype)
        eng.get_config(Cfg)
        if hook is not None and frozen.any():
            # a frozen replica returns what it had when its hook said stop; the engine is given the same configuration, so that C and the
            # device agree (its chain went on in the meantime: independent of the others, and discarded here)
            Cfg.s[frozen] = frozen_cfg.s[frozen]
            accepted[frozen] = frozen_acc[frozen]
            eng.set_config(Cfg)
            Es = [Es[r, :frozen_nsamp[r]] if frozen[r] else Es[r] for r in range(eng.R)]
        if not quiet:
            print("samples = ", Es.shape[1] if hasattr(Es, "shape") else [len(e) for e in Es])
            print("iters = ", it)
            # a frozen replica's count is over the iterations it ran (frozen_nsamp samples of `step` iterations each)
            ran = np.full(eng.R, max(it, 1), np.float64)
            if hook is not None and frozen.any():
                ran[frozen] = np.maximum(frozen_nsamp[frozen] * step, 1)
            print("accept rate = ", float((accepted / ran).mean()))
        return Es, Cfg
    finally:
        if own:
            eng.close()
