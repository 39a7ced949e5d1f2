"""The hook of every sampler (src/RRRMC.jl:152,186 rrrMC(SingleGraph) · :224,255 rrrMC(DoubleGraph) · :314,341 bklMC · :379,404 wtmMC ·
:477,501 extremal_opt): the library cuts the run at the hook points and RESUMES it (rrrmc_set_resume, include/rrrmc_hip.h), so that

  1. a hooked run is the un-hooked chain bit for bit — samples, configurations, counts, the move-selection cache and everything the next
     iterations depend on (checked by continuing both runs with one more resumed call);
  2. what the hook is handed at every sample — (it, C, accepted, E), or (it, C, E, Emin) for extremal_opt — is what the ORACLE's in-loop hook
     sees at the same sample (oracle.hooked: the restatement calls its hook where the reference does);
  3. a run made in one call followed by a resumed call equals the oracle's single run of the total length (the resumed call continues the
     run's iteration counter, skip position, global time, Emin / itmin);
  4. a hook that returns False ends the chain where the reference's `|| break` does.

Every kernel build the host can pick is exercised through its environment switch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 8426732438942          # test/runtests.jl:25


# ---- the graphs of the verdict's list, each with the oracle's samplers -------------------------------------------------------------
class RRG:
    f64 = False

    def __init__(self, pkg, N=120, K=3):
        self.X = pkg.GraphRRG(N, K, seed=SEED + N)
        self.A, self.J = self.X.A, self.X.J.astype(np.int32)

    def oracle(self, O, smp, ch, r, seed, iters, step, beta=None, tau=None, thr=None):
        if smp == "rrr":
            o = O.rrr_sparse(self.A, self.J, beta, iters, step, seed, ch, replica=r, staged_thr=0.5 if thr is None else thr)
            return o[0], o[1], o[2]
        if smp == "bkl":
            o = O.rrr_sparse(self.A, self.J, beta, iters, step, seed, ch, replica=r, bkl=True)
            return o[0], o[1], o[2]
        if smp == "wtm":
            o = O.wtm_mc_sparse(self.A, self.J, beta, iters, float(step), seed, ch, replica=r)
            return o[0], o[1], o[2]
        o = O.extremal_opt_sparse(self.A, self.J, tau, iters, step, seed, ch, replica=r)
        return o[0], o[1], (o[2], o[3], o[4])


class EA(RRG):
    def __init__(self, pkg, L=2, D=3):          # test/runtests.jl:46: the doubled bonds of L = 2
        self.X = pkg.GraphEA(L, D, seed=SEED + 1)
        self.A, self.J = self.X.A, self.X.J.astype(np.int32)

    def oracle(self, O, smp, ch, r, seed, iters, step, beta=None, tau=None, thr=None):
        if smp == "rrr":
            o = O.rrr_sparse(self.A, self.J, beta, iters, step, seed, ch, replica=r, staged_thr=0.5 if thr is None else thr, form="ea")
            return o[0], o[1], o[2]
        if smp == "bkl":
            o = O.rrr_sparse(self.A, self.J, beta, iters, step, seed, ch, replica=r, bkl=True, form="ea")
            return o[0], o[1], o[2]
        if smp == "wtm":
            o = O.wtm_mc_sparse(self.A, self.J, beta, iters, float(step), seed, ch, replica=r, form="ea")
            return o[0], o[1], o[2]
        o = O.extremal_opt_sparse(self.A, self.J, tau, iters, step, seed, ch, replica=r, form="ea")
        return o[0], o[1], (o[2], o[3], o[4])


class RRGNormal:
    f64 = True

    def __init__(self, pkg, N=100, K=3):
        self.X = pkg.GraphRRGNormal(N, K, seed=SEED + 2 * N)
        self.A, self.J = self.X.A, self.X.J

    def oracle(self, O, smp, ch, r, seed, iters, step, beta=None, tau=None, thr=None):
        if smp == "eo":
            o = O.extremal_opt_cont(self.A, self.J, tau, iters, step, seed, ch, replica=r)
            return o[0], o[1], (o[2], o[3], o[4])
        if smp == "wtm":
            o = O.cont_sparse("wtm", self.A, self.J, beta, iters, 1, seed, ch, replica=r, stepf=float(step))
            return o[0], o[1], int(o[2][0])
        o = O.cont_sparse(smp, self.A, self.J, beta, iters, step, seed, ch, replica=r, staged_thr=0.8 if thr is None else thr)
        return o[0], o[1], int(o[2][0])


class SKNormal:
    f64 = True

    def __init__(self, pkg, N=24):
        self.X = pkg.GraphSKNormal(N, seed=SEED + 3)
        self.J = self.X.J

    def oracle(self, O, smp, ch, r, seed, iters, step, beta=None, tau=None, thr=None):
        if smp == "rrr":
            o = O.rrr_mc_skn(self.J, beta, iters, step, seed, ch, replica=r, staged_thr=0.8 if thr is None else thr)
            return o[0], o[1], o[2]
        if smp == "bkl":
            o = O.bkl_mc_skn(self.J, beta, iters, step, seed, ch, replica=r)
            return o[0], o[1], o[2]
        if smp == "wtm":
            o = O.wtm_mc_skn(self.J, beta, iters, float(step), seed, ch, replica=r)
            return o[0], o[1], o[2]
        o = O.extremal_opt_sk(self.J, tau, iters, step, seed, ch, replica=r)
        return o[0], o[1], (o[2], o[3], o[4])


class Quant:
    f64 = True

    def __init__(self, pkg, Nk=10, M=8, Gamma=0.5, beta=2.0):          # test/runtests.jl:78
        self.X1 = pkg.GraphRRG(Nk, 3, seed=SEED + Nk)
        self.X = pkg.GraphQuant(self.X1, M, Gamma, beta)
        self.A, self.J, self.M = self.X1.A, self.X1.J.astype(np.int32), M

    def oracle(self, O, smp, ch, r, seed, iters, step, beta=None, tau=None, thr=None):
        if smp == "rrr":
            o = O.rrr_mc_quant(self.A, self.J, self.M, self.X.fourK, beta, iters, step, seed, ch, replica=r, staged_thr=0.5 if thr is None else thr)
            return o[0], o[1], o[2]
        if smp == "eo":
            o = O.extremal_opt_quant(self.A, self.J, self.M, self.X.fourK, tau, iters, step, seed, ch, replica=r)
            return o[0], o[1], (o[2], o[3], o[4])
        if smp == "wtm":
            o = O.cont_quant("wtm", self.A, self.J, self.M, self.X.fourK, beta, iters, 1, seed, ch, replica=r, stepf=float(step))
            return o[0], o[1], int(o[2][0])
        o = O.cont_quant("bkl", self.A, self.J, self.M, self.X.fourK, beta, iters, step, seed, ch, replica=r)
        return o[0], o[1], int(o[2][0])


MODELS = {"rrg": RRG, "ea": EA, "rrgn": RRGNormal, "skn": SKNormal, "quant": Quant}


def front(pkg, smp):
    return {"rrr": pkg.rrrMC, "bkl": pkg.bklMC, "wtm": pkg.wtmMC, "eo": pkg.extremal_opt}[smp]


def engine_call(eng, smp, n, step, beta, tau, thr):
    """one (possibly resumed) engine call; returns (Es, count)"""
    if smp == "rrr":
        Es, acc, _ = eng.rrr_mc(beta, n, step, staged_thr=thr)
        return Es, acc
    if smp == "bkl":
        return eng.bkl_mc(beta, n, step)
    if smp == "wtm":
        Es, mv, _ = eng.wtm_mc(beta, n, float(step))
        return Es, mv
    Es, Emin, Cmin, itmin = eng.extremal_opt(tau, n, step)
    return Es, np.stack([np.asarray(Emin, np.float64), np.asarray(itmin, np.float64)], 1)


def run_both(pkg, oracle, M, smp, R, iters, step, beta=2.0, tau=1.3, thr=None, tail=None, check=(0,), engine_kw=None):
    """un-hooked run, hooked run and their resumed continuations on one engine; the oracle's hooked run for the replicas in `check`"""
    X = M.X
    fn = front(pkg, smp)
    kw = {"step": float(step) if smp == "wtm" else step, "quiet": True}
    if smp == "rrr" and thr is not None:
        kw["staged_thr"] = thr
    arg = tau if smp == "eo" else beta
    n_arg = iters // step if smp == "wtm" else iters              # wtmMC takes the number of samples
    tail = step * 3 if tail is None else tail
    n_tail = tail // step if smp == "wtm" else tail
    rec = []

    def hook(it, X_, C, a, b):
        assert X_ is X
        rec.append((float(it), C.s.copy(), np.array(a), np.array(b)))
        return True

    out = {}
    with pkg.Engine(X, R, **(engine_kw or {})) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        C0 = eng.get_config()
        for mode in ("plain", "hooked"):
            res = fn(X, arg, n_arg, seed=SEED, C0=C0.copy(), engine=eng, hook=hook if mode == "hooked" else None, **kw)
            state = {"res": res, "E": eng.run_energy()}
            if smp in ("rrr", "bkl") and X.model_kind in (1, 3, 6, 7) and not (smp == "bkl" and X.model_kind in (3, 6)):
                state["cache"] = eng.rrr_cache()
            # continue the run with one more (resumed) call: everything the next iterations depend on must agree
            eng.set_resume(True)
            state["tail"] = engine_call(eng, smp, n_tail, step, beta, tau, thr)
            eng.set_resume(False)
            state["C_tail"] = eng.get_config().s.copy()
            out[mode] = state
    u, h = out["plain"], out["hooked"]
    # 1. bit for bit
    if smp == "eo":
        Cu, Eminu, Cminu, itu = u["res"]
        Ch, Eminh, Cminh, ith = h["res"]
        assert (Cu.s == Ch.s).all() and (np.asarray(Eminu) == np.asarray(Eminh)).all() and (Cminu.s == Cminh.s).all() and (np.asarray(itu) == np.asarray(ith)).all()
        Es_h = np.stack([x[2] for x in rec], 1)                     # the energies the hook saw
        assert Es_h.shape[1] == iters // step
    else:
        assert (np.asarray(u["res"][0]) == np.asarray(h["res"][0])).all() and (u["res"][1].s == h["res"][1].s).all()
        Es_h = np.asarray(h["res"][0])
        assert Es_h.shape == (R, iters // step)
        assert (np.stack([x[3] for x in rec], 1) == Es_h).all()      # E handed to the hook = the sample
    assert (u["E"] == h["E"]).all()
    if "cache" in u:
        assert (u["cache"][0] == h["cache"][0]).all() and (u["cache"][1] == h["cache"][1]).all()
    assert (np.asarray(u["tail"][0]) == np.asarray(h["tail"][0])).all() and (np.asarray(u["tail"][1]) == np.asarray(h["tail"][1])).all()
    assert (u["C_tail"] == h["C_tail"]).all()
    assert np.asarray(u["tail"][0]).shape[1] == tail // step
    # 2. / 3. the oracle
    for r in check:
        with oracle.hooked() as calls:
            oEs, och, ocnt = M.oracle(oracle, smp, C0.s[r], r, SEED, n_arg, step, beta=beta, tau=tau, thr=thr)
        assert len(calls) == len(rec) == iters // step
        for (it, ch, acc, E, Emin), (hit, hC, ha, hb) in zip(calls, rec):
            assert hit == it and (hC[r] == ch).all()
            if smp == "eo":
                assert ha[r] == E and hb[r] == Emin
            else:
                assert ha[r] == acc and hb[r] == E
        if smp == "eo":
            assert (Cu.s[r] == och).all() and np.asarray(Eminu)[r] == ocnt[0] and (Cminu.s[r] == ocnt[1]).all() and np.asarray(itu)[r] == ocnt[2]
        else:
            assert (Es_h[r] == oEs).all() and (h["res"][1].s[r] == och).all()
        # the run continued by a resumed call = one run of the total length
        tEs, tch, tcnt = M.oracle(oracle, smp, C0.s[r], r, SEED, n_arg + n_tail, step, beta=beta, tau=tau, thr=thr)
        full = np.concatenate([np.asarray(u["res"][0])[r] if smp != "eo" else Es_h[r], np.asarray(u["tail"][0])[r]])
        assert (full == tEs).all() and (u["C_tail"][r] == tch).all()
        if smp == "eo":
            assert u["tail"][1][r, 0] == tcnt[0] and u["tail"][1][r, 1] == tcnt[2]
    return rec


BUILDS_SPARSE = {
    "wave": {},                                                                  # rrr_sparse_wave_kernel (few replicas)
    "wave-respace": {"RRRMC_RRR_WAVE_SLACK": "8"},                               # ... whose segments re-space every few batches
    "lds": {"RRRMC_RRR_NO_WAVE": "1"},                                           # rrr_sparse_kernel<LDS = true>
    "thread": {"RRRMC_RRR_NO_WAVE": "1", "RRRMC_RRR_NO_LDS": "1"},               # rrr_sparse_kernel<LDS = false>
}


@pytest.mark.parametrize("build", list(BUILDS_SPARSE))
@pytest.mark.parametrize("smp,thr", [("rrr", None), ("rrr", 0.0), ("rrr", 1.0), ("bkl", None)])
def test_rrg_rrr_bkl_hooked_through_every_build(pkg, oracle, monkeypatch, build, smp, thr):
    for k, v in BUILDS_SPARSE[build].items():
        monkeypatch.setenv(k, v)
    run_both(pkg, oracle, RRG(pkg), smp, 5, 4000, 100, thr=thr, check=(0, 4))


@pytest.mark.parametrize("smp", ["wtm", "eo"])
def test_rrg_wtm_eo_hooked(pkg, oracle, smp):
    run_both(pkg, oracle, RRG(pkg), smp, 5, 3000, 100, check=(0, 3))


@pytest.mark.parametrize("smp", ["rrr", "bkl", "wtm", "eo"])
def test_ea_l2_hooked(pkg, oracle, smp):
    """GraphEA(2, 3): every neighbour appears twice in a row of A (test/runtests.jl:46)"""
    run_both(pkg, oracle, EA(pkg), smp, 3, 2000, 100, check=(0, 2))


BUILDS_CONT = {"wave": {}, "thread": {"RRRMC_CONT_NO_WAVE": "1", "RRRMC_EO_NO_WAVE": "1"}}


@pytest.mark.parametrize("build", list(BUILDS_CONT))
@pytest.mark.parametrize("smp,thr", [("rrr", None), ("rrr", 0.0), ("rrr", 1.0), ("bkl", None), ("wtm", None), ("eo", None)])
def test_rrgnormal_hooked_through_both_builds(pkg, oracle, monkeypatch, build, smp, thr):
    """GraphRRGNormal: DeltaECacheCont + DynamicSampler (cont_wave_kernel / cont_sparse_kernel), THeap, EOCacheCont (eo_cont_wave_kernel /
    mode 3 of cont_sparse_kernel).  N = 100: the sampler refreshes its tree every 100 setindex! calls — many times inside and across pieces."""
    for k, v in BUILDS_CONT[build].items():
        monkeypatch.setenv(k, v)
    if build == "thread" and smp == "wtm":
        pytest.skip("wtmMC has one build")
    run_both(pkg, oracle, RRGNormal(pkg), smp, 4, 3000, 100, thr=thr, check=(0, 3))


@pytest.mark.parametrize("smp,thr", [("rrr", None), ("rrr", 0.0), ("rrr", 1.0), ("bkl", None), ("wtm", None), ("eo", None)])
def test_sknormal_hooked(pkg, oracle, smp, thr):
    """GraphSKNormal(24): rrr_skn_kernel (the lfields / lfields_last swap, move_last, the tree's refresh countdown carry on), eo_sk_wave_kernel"""
    run_both(pkg, oracle, SKNormal(pkg), smp, 4, 2000, 50, thr=thr, check=(0, 3))


BUILDS_QUANT = {"wave": {}, "wave-respace": {"RRRMC_QUANT_WAVE_SLACK": "2048"}, "lds": {"RRRMC_QUANT_NO_WAVE": "1"},
                "thread": {"RRRMC_QUANT_NO_WAVE": "1", "RRRMC_QUANT_NO_LDS": "1"}}


@pytest.mark.parametrize("build", list(BUILDS_QUANT))
def test_quant_config5_geometry_rrr_hooked_through_every_build(pkg, oracle, monkeypatch, build):
    """BASELINE config 5's geometry: GraphQuant(GraphRRG(1024, 3), M = 32), N = 32 768, rrrMC"""
    for k, v in BUILDS_QUANT[build].items():
        monkeypatch.setenv(k, v)
    run_both(pkg, oracle, Quant(pkg, 1024, 32, 0.5, 2.0), "rrr", 3, 12288, 4096, tail=4096, check=(0,))


@pytest.mark.parametrize("smp,thr", [("rrr", None), ("rrr", 0.0), ("rrr", 1.0), ("bkl", None), ("wtm", None), ("eo", None)])
def test_quant_runtests_graph_hooked(pkg, oracle, smp, thr):
    """GraphQuant(10, 8, 0.5, 2.0, GraphRRG, 10, 3) of test/runtests.jl:78 under every sampler (bklMC / wtmMC / extremal_opt: the generic caches)"""
    run_both(pkg, oracle, Quant(pkg), smp, 3, 2000, 100, thr=thr, check=(0, 2))


# ---- the other graph families: hooked == un-hooked, and the continuation agrees (their un-hooked chains are oracle-checked elsewhere) ----
class Plain:
    def __init__(self, X):
        self.X = X


def other_graphs(pkg):
    return {
        "rrg-levels": lambda: pkg.GraphRRG(40, 3, LEV=(-1, 0, 1), seed=SEED),                       # RRRMC_MODEL_SPARSE_LEVELS (test/runtests.jl:37)
        "ea-levels": lambda: pkg.GraphEA(3, 2, LEV=(-1, 0, 1), seed=SEED),
        "ea-normal-l2": lambda: pkg.GraphEANormal(2, 3, seed=SEED),                                # repeated neighbours: the thread build of the continuous samplers
        "rrg-discretized": lambda: pkg.GraphRRGNormalDiscretized(40, 3, (-1, 0, 1), seed=SEED),    # rrr_dbl_kernel; bkl / wtm / eo over the whole DoubleGraph
        "ea-discretized": lambda: pkg.GraphEANormalDiscretized(3, 2, (-1, 0, 1), seed=SEED),
        "sk-binary": lambda: pkg.GraphSK(20, seed=SEED),                                           # integer fields, delta_energy = lfields / sqrt(N); EO ties
        "qskt": lambda: pkg.GraphQSKT(12, 4, 0.5, 2.0, seed=SEED),                                 # GraphQuant over binary GraphSK slices
        "qsknormal": lambda: pkg.GraphQSKNormalT(10, 4, 0.5, 2.0, seed=SEED),                      # GraphQuant over GraphSKNormal slices
        "qeat": lambda: pkg.GraphQEAT(3, 2, 4, 0.5, 2.0, seed=SEED),                               # GraphQuant over GraphEANormal slices (their Float64 caches and undo records carry on)
    }


@pytest.mark.parametrize("name", ["rrg-levels", "ea-levels", "ea-normal-l2", "rrg-discretized", "ea-discretized", "sk-binary", "qskt", "qsknormal", "qeat"])
@pytest.mark.parametrize("smp", ["rrr", "bkl", "wtm", "eo"])
def test_other_graph_families_hooked(pkg, oracle, name, smp):
    X = other_graphs(pkg)[name]()
    try:
        run_both(pkg, oracle, Plain(X), smp, 3, 1500, 100, check=())
    except pkg.RRRMCError as e:
        if e.code == 3:          # RRRMC_ERR_UNSUPPORTED: a sampler the library does not offer for this graph (include/rrrmc_hip.h)
            pytest.skip(str(e))
        raise


@pytest.mark.parametrize("smp", ["rrr", "bkl", "wtm", "eo"])
def test_hooked_run_on_a_multi_device_context(pkg, oracle, smp):
    """rrrmc_ctx_create_multi: two shards (here both on device 0); every shard keeps its own run, the hook sees the gathered replicas"""
    run_both(pkg, oracle, RRG(pkg, 60), smp, 64, 1000, 100, check=(0, 33, 63), engine_kw={"devices": [0, 0]})


# ---- the reference's own test loop ------------------------------------------------------------------------------------------------
def test_runtests_jl_hook_loop(pkg, oracle):
    """test/runtests.jl:130-162 line for line on its GraphRRG(10, 3), GraphEA(2, 3), GraphSKNormal(10) and GraphQuant(10, 8, 0.5, 2.0, ...):
    every sampler with checkenergy_hook (E ≈ energy(X, C), atol 1e-11, :12-20) and with a hook that stops the run."""
    import time
    beta, iters, st, tau = 2.0, 10_000, 100, 1.3
    samples = iters // st
    graphs = [pkg.GraphRRG(10, 3, seed=SEED), pkg.GraphEA(2, 3, seed=SEED), pkg.GraphSKNormal(10, seed=SEED),
              pkg.GraphQuant(pkg.GraphRRG(10, 3, seed=SEED), 8, 0.5, 2.0)]
    for X in graphs:
        with pkg.EnergyProbe(X, 2) as energy:
            ncalls = [0]

            def checkenergy_hook(it, X_, C, acc, E):
                ncalls[0] += 1
                assert np.allclose(E, energy(C), rtol=0, atol=1e-11)
                return True

            def checkenergy_hook_EO(it, X_, C, E, Emin):
                ncalls[0] += 1
                assert np.allclose(E, energy(C), rtol=0, atol=1e-11) and (np.asarray(Emin) <= np.asarray(E) + 1e-11).all()
                return True

            def gen_timeout_hook(t=0.05):
                t += time.time()
                return lambda *a: time.time() <= t

            kw = dict(quiet=True, replicas=2)
            with pkg.Engine(X, 2) as eng:
                kw["engine"] = eng
                eng.seed(SEED)
                eng.init_spins_random()          # (a call without C0 on a caller's engine continues the engine's configuration)
                E, C = pkg.standardMC(X, beta, iters, step=st, **kw)
                E, C = pkg.standardMC(X, beta, iters, step=st, C0=C, hook=checkenergy_hook, **kw)
                E, C = pkg.standardMC(X, beta, iters, step=st, C0=C, hook=gen_timeout_hook(), **kw)

                E, C = pkg.bklMC(X, beta, iters, step=st, **kw)
                E, C = pkg.bklMC(X, beta, iters, step=st, C0=C, hook=checkenergy_hook, **kw)
                E, C = pkg.bklMC(X, beta, iters, step=st, C0=C, hook=gen_timeout_hook(), **kw)

                E, C = pkg.wtmMC(X, beta, samples, step=float(st), **kw)
                E, C = pkg.wtmMC(X, beta, samples, step=float(st), C0=C, hook=checkenergy_hook, **kw)
                E, C = pkg.wtmMC(X, beta, samples, step=float(st), C0=C, hook=gen_timeout_hook(), **kw)

                E, C = pkg.rrrMC(X, beta, iters, step=st, **kw)
                E, C = pkg.rrrMC(X, beta, iters, step=st, C0=C, hook=checkenergy_hook, **kw)
                E, C = pkg.rrrMC(X, beta, iters, step=st, C0=C, hook=gen_timeout_hook(), **kw)
                E, C = pkg.rrrMC(X, beta, iters, step=st, staged_thr=0.0, hook=checkenergy_hook, **kw)
                E, C = pkg.rrrMC(X, beta, iters, step=st, C0=C, staged_thr=0.0, **kw)
                E, C = pkg.rrrMC(X, beta, iters, step=st, staged_thr=1.0, hook=checkenergy_hook, **kw)
                E, C = pkg.rrrMC(X, beta, iters, step=st, C0=C, staged_thr=1.0, **kw)

                C, Emin, Cmin, itmin = pkg.extremal_opt(X, tau, iters, step=st, **kw)
                C, Emin, Cmin, itmin = pkg.extremal_opt(X, tau, iters, step=st, C0=C, **kw)
                C, Emin, Cmin, itmin = pkg.extremal_opt(X, tau, iters, step=st, hook=checkenergy_hook_EO, **kw)
                C, Emin, Cmin, itmin = pkg.extremal_opt(X, tau, iters, step=st, hook=gen_timeout_hook(), **kw)
            assert ncalls[0] == 7 * samples          # the seven checked runs called their hook at every sample


# ---- hooks that stop ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["rrg", "rrgn", "skn", "quant"])
@pytest.mark.parametrize("smp", ["rrr", "bkl", "wtm", "eo"])
def test_stopping_hook_ends_the_chain_where_the_reference_does(pkg, oracle, name, smp):
    """`hook(...) || break`: the samples include the one the stopping hook saw, the configuration is the one it saw (the move of that
    iteration is not made), extremal_opt returns its Emin / Cmin / itmin of that moment."""
    M = MODELS[name](pkg)
    X = M.X
    iters, step, stop_at = 2000, 100, 7
    fn = front(pkg, smp)
    arg = 1.3 if smp == "eo" else 2.0
    n_arg = iters // step if smp == "wtm" else iters
    count = [0]

    def hook(*a):
        count[0] += 1
        return count[0] < stop_at

    res = fn(X, arg, n_arg, seed=SEED, step=float(step) if smp == "wtm" else step, hook=hook, quiet=True, replicas=2)
    C0 = oracle.init_configs(SEED, 0, 2, X.N)
    for r in range(2):
        ocount = [0]

        def ohook(*a):
            ocount[0] += 1
            return ocount[0] < stop_at

        with oracle.hooked(ohook) as calls:
            oEs, och, ocnt = M.oracle(oracle, smp, C0[r], r, SEED, n_arg, step, beta=2.0, tau=1.3)
        assert len(calls) == stop_at
        if smp == "eo":
            C, Emin, Cmin, itmin = res
            assert (C.s[r] == och).all() and np.asarray(Emin)[r] == ocnt[0] and (Cmin.s[r] == ocnt[1]).all() and np.asarray(itmin)[r] == ocnt[2]
        else:
            Es, C = res
            assert np.asarray(Es).shape == (2, stop_at) and (np.asarray(Es)[r] == oEs[:stop_at]).all() and (C.s[r] == och).all()
    assert count[0] == stop_at


def test_one_replica_stops_the_others_go_on(pkg, oracle):
    """a hook may return one flag per replica: the replica whose flag is False is frozen at that sample (what the reference's chain returns),
    the others run to the end"""
    M = RRG(pkg)
    X = M.X
    R, iters, step = 4, 3000, 100

    def hook(it, X_, C, acc, E):
        return np.array([True, it < 1000, True, it < 2000])

    Es, C = pkg.rrrMC(X, 2.0, iters, seed=SEED, step=step, hook=hook, quiet=True, replicas=R)
    C0 = oracle.init_configs(SEED, 0, R, X.N)
    for r, stop in enumerate([None, 1000, None, 2000]):
        with oracle.hooked((lambda it, *a: True) if stop is None else (lambda it, *a, s=stop: it < s)):
            oEs, och, _ = M.oracle(oracle, "rrr", C0[r], r, SEED, iters, step, beta=2.0, thr=0.5)
        n = iters // step if stop is None else stop // step
        assert len(Es[r]) == n and (np.asarray(Es[r]) == oEs[:n]).all() and (C.s[r] == och).all()


# ---- the C ABI's contract ------------------------------------------------------------------------------------------------------------
def test_resume_contract(pkg, oracle):
    """what continues a run and what ends it (include/rrrmc_hip.h, rrrmc_set_resume)"""
    M = RRG(pkg)
    X = M.X
    with pkg.Engine(X, 2) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        C0 = eng.get_config()
        ref = oracle.rrr_sparse(M.A, M.J, 2.0, 1000, 100, SEED, C0.s[0], replica=0)
        # without resume mode every call starts a run: energy(X, C), a fresh cache
        a = eng.rrr_mc(2.0, 500, 100)
        b = eng.rrr_mc(2.0, 500, 100)
        two = oracle.rrr_sparse(M.A, M.J, 2.0, 500, 100, SEED, oracle.rrr_sparse(M.A, M.J, 2.0, 500, 100, SEED, C0.s[0])[1], it0=500)
        assert (b[0][0] == two[0]).all() and not (np.concatenate([a[0][0], b[0][0]]) == ref[0]).all()
        # with it, the second call continues the first
        eng.seed(SEED)
        eng.set_config(C0)
        eng.set_resume(True)
        a = eng.rrr_mc(2.0, 450, 100)           # samples before iterations 100 .. 400
        b = eng.rrr_mc(2.0, 550, 100)           # ... and 500 .. 1000: multiples of `step` of the RUN's count
        assert a[0].shape == (2, 4) and b[0].shape == (2, 6)
        assert (np.concatenate([a[0][0], b[0][0]]) == ref[0]).all() and (eng.get_config().s[0] == ref[1]).all() and a[1][0] + b[1][0] == ref[2]
        # other parameters, another sampler, a new configuration or an energy call end the run: the next call starts one, as a reference call does
        c = eng.rrr_mc(2.5, 100, 10)
        oc = oracle.rrr_sparse(M.A, M.J, 2.5, 100, 10, SEED, ref[1], it0=1000, replica=0)
        assert (c[0][0] == oc[0]).all()
        b1 = eng.bkl_mc(2.0, 300, 100)
        ob1 = oracle.rrr_sparse(M.A, M.J, 2.0, 300, 100, SEED, oc[1], it0=1100, replica=0, bkl=True)
        assert (b1[0][0] == ob1[0]).all()
        eng.energy()
        d = eng.bkl_mc(2.0, 300, 100)           # a fresh bklMC run, not the continuation of the one before the energy call
        od = oracle.rrr_sparse(M.A, M.J, 2.0, 300, 100, SEED, ob1[1], it0=1400, replica=0, bkl=True)
        assert (d[0][0] == od[0]).all() and (eng.get_config().s[0] == od[1]).all()
        eng.set_resume(False)


# ---- cuts that are not multiples of `step` -------------------------------------------------------------------------------------------------
BKL_CUTS = [(10, [6, 95, 15]), (3, [72, 63, 1, 260, 656]), (50, [18, 11, 41, 108, 16]), (50, [2, 14, 72, 43, 10]), (7, [4, 395, 49, 35, 38]),
            (10, [3, 3]), (10, [30, 5, 2, 3, 60])]


@pytest.mark.parametrize("step,pieces", BKL_CUTS, ids=["%d-%s" % (s, "_".join(map(str, p))) for s, p in BKL_CUTS])
@pytest.mark.parametrize("name,build", [("rrg", {}), ("rrg", {"RRRMC_RRR_NO_WAVE": "1"}), ("rrg", {"RRRMC_RRR_NO_WAVE": "1", "RRRMC_RRR_NO_LDS": "1"}),
                                        ("ea", {}), ("rrgn", {}), ("rrgn", {"RRRMC_CONT_NO_WAVE": "1"}), ("skn", {}), ("quant", {})],
                         ids=["rrg-wave", "rrg-lds", "rrg-thread", "ea-wave", "rrgn-wave", "rrgn-thread", "skn", "quant"])
def test_bkl_run_cut_anywhere_is_the_run_made_in_one_call(pkg, oracle, monkeypatch, name, build, step, pieces):
    """bklMC's loop (RRRMC.jl:327-350) jumps over skipped iterations and ends with its LAST SAMPLE: a resumed call whose allowance ends between
    two sample points must neither take the sample that lies beyond it nor — once the run's last sample is taken — make another move.
    Found by tests/soak/hook_soak.py: pieces shorter than `step` at the start, a piece of one iteration right after a sample point."""
    for k in ("RRRMC_RRR_NO_WAVE", "RRRMC_RRR_NO_LDS", "RRRMC_CONT_NO_WAVE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in build.items():
        monkeypatch.setenv(k, v)
    M = MODELS[name](pkg)
    R, beta, total = 70, 1.0, sum(pieces)
    outs = []
    for plan in ([total], pieces):
        with pkg.Engine(M.X, R) as eng:
            eng.seed(SEED)
            eng.init_spins_random()
            C0 = eng.get_config()
            Es, moves = [], 0
            for i, n in enumerate(plan):
                eng.set_resume(i > 0)
                a, m = eng.bkl_mc(beta, n, step)
                Es.append(np.asarray(a)); moves = moves + np.asarray(m)
            eng.set_resume(False)
            outs.append((np.concatenate(Es, 1), moves, eng.get_config().s.copy(), np.asarray(eng.run_energy())))
    for a, b in zip(*outs):
        assert a.shape == b.shape and (a == b).all()
    assert outs[0][0].shape == (R, total // step)
    if total >= step:
        oEs, och, ocnt = M.oracle(oracle, "bkl", C0.s[3], 3, SEED, total, step, beta=beta)
        assert (outs[1][0][3] == oEs).all() and (outs[1][2][3] == och).all()
