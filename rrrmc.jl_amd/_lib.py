"""ctypes binding of include/rrrmc_hip.h.  No fallback: a missing library or device is an error."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None

u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")

# every symbol include/rrrmc_hip.h declares
SYMBOLS = [
    "rrrmc_version", "rrrmc_last_error", "rrrmc_device_count", "rrrmc_device_copy_bandwidth", "rrrmc_host_alloc", "rrrmc_host_free", "rrrmc_ctx_create", "rrrmc_ctx_create_multi", "rrrmc_ctx_destroy",
    "rrrmc_set_graph", "rrrmc_seed", "rrrmc_init_spins_random", "rrrmc_set_spins", "rrrmc_get_spins",
    "rrrmc_energy", "rrrmc_get_fields", "rrrmc_standard_mc", "rrrmc_standard_mc_async", "rrrmc_sync",
    "rrrmc_fetch_results", "rrrmc_last_timing", "rrrmc_timing_accumulate", "rrrmc_timing_total", "rrrmc_set_resume", "rrrmc_set_debug_checks", "rrrmc_tracked_energy_f64", "rrrmc_tracked_energy", "rrrmc_results_samples", "rrrmc_spf_team_build", "rrrmc_standard_mc_fast_async", "rrrmc_iterations_done", "rrrmc_gen_rrg", "rrrmc_gen_ea",
    "rrrmc_gen_couplings_pm1", "rrrmc_gen_couplings_lev", "rrrmc_set_graph_levels", "rrrmc_set_couplings_dense", "rrrmc_energy_f64", "rrrmc_get_fields_f64",
    "rrrmc_standard_mc_f64", "rrrmc_fetch_results_f64", "rrrmc_gen_sk_gauss",
    "rrrmc_set_couplings_bits", "rrrmc_gen_sk_binary", "rrrmc_set_coloring", "rrrmc_colored_sweeps_async", "rrrmc_colored_count_accepted",
    "rrrmc_ctx_create_quant", "rrrmc_ctx_create_quant_sk", "rrrmc_ctx_create_quant_skn", "rrrmc_ctx_create_quant_f64", "rrrmc_quant_set_field", "rrrmc_quant_slice_form", "rrrmc_rrr_mc_async", "rrrmc_rrr_stats", "rrrmc_rrr_cache", "rrrmc_bkl_mc_async",
    "rrrmc_snapshot_reserve", "rrrmc_snapshot_store", "rrrmc_snapshot_get", "rrrmc_overlaps", "rrrmc_quant_observables",
    "rrrmc_set_graph_f64", "rrrmc_gen_couplings_gauss", "rrrmc_set_graph_discretized", "rrrmc_set_level_scale", "rrrmc_discretize", "rrrmc_discretize_scaled", "rrrmc_wtm_mc_async", "rrrmc_wtm_times", "rrrmc_extremal_opt_async", "rrrmc_extremal_opt_results", "rrrmc_extremal_opt_results_f64",
]


class RRRMCError(RuntimeError):
    """Raised for every non-zero status of the C ABI (the reference raises ArgumentError / ErrorException)."""

    def __init__(self, code, msg):
        super().__init__("rrrmc_hip status %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = _build.lib_path()
    if path == _build.OUT and not os.path.exists(path):
        # the in-tree library is missing: build it with hipcc if there is one (atomic rename, safe with several ranks) — never fall back
        try:
            _build.build()
        except Exception as e:      # no hipcc, or the compilation failed
            if not os.path.exists(path):
                raise RuntimeError("%s is missing and could not be built (%s): run `python -c 'import __graft_entry__ as g; "
                                   "g.build()'` (there is no CPU fallback)" % (path, e))
    if not os.path.exists(path):
        raise RuntimeError("%s is missing (there is no CPU fallback)" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.rrrmc_version.restype = C.c_int32
    L.rrrmc_last_error.restype = C.c_char_p
    L.rrrmc_last_error.argtypes = [vp]
    L.rrrmc_device_count.restype = C.c_int32
    L.rrrmc_host_alloc.restype = C.c_int32
    L.rrrmc_host_alloc.argtypes = [C.c_int64, C.POINTER(C.c_void_p)]
    L.rrrmc_host_free.restype = C.c_int32
    L.rrrmc_host_free.argtypes = [C.c_void_p]
    L.rrrmc_device_copy_bandwidth.restype = C.c_int32
    L.rrrmc_device_copy_bandwidth.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
    L.rrrmc_ctx_create.restype = C.c_int32
    L.rrrmc_ctx_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_uint32]
    L.rrrmc_ctx_create_multi.restype = C.c_int32
    L.rrrmc_ctx_create_multi.argtypes = [C.POINTER(vp), C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, i32p, C.c_int32, C.c_uint32]
    L.rrrmc_ctx_destroy.restype = None
    L.rrrmc_ctx_destroy.argtypes = [vp]
    L.rrrmc_set_graph.restype = C.c_int32
    L.rrrmc_set_graph.argtypes = [vp, i32p, i8p]
    L.rrrmc_seed.restype = C.c_int32
    L.rrrmc_seed.argtypes = [vp, C.c_uint64]
    L.rrrmc_init_spins_random.restype = C.c_int32
    L.rrrmc_init_spins_random.argtypes = [vp]
    L.rrrmc_set_spins.restype = C.c_int32
    L.rrrmc_set_spins.argtypes = [vp, u64p]
    L.rrrmc_get_spins.restype = C.c_int32
    L.rrrmc_get_spins.argtypes = [vp, u64p]
    L.rrrmc_energy.restype = C.c_int32
    L.rrrmc_energy.argtypes = [vp, i64p]
    L.rrrmc_get_fields.restype = C.c_int32
    L.rrrmc_get_fields.argtypes = [vp, i64p]
    L.rrrmc_standard_mc.restype = C.c_int32
    L.rrrmc_standard_mc.argtypes = [vp, C.c_double, C.c_int64, C.c_int64, vp, vp]
    L.rrrmc_standard_mc_async.restype = C.c_int32
    L.rrrmc_standard_mc_async.argtypes = [vp, C.c_double, C.c_int64, C.c_int64]
    L.rrrmc_sync.restype = C.c_int32
    L.rrrmc_sync.argtypes = [vp]
    L.rrrmc_fetch_results.restype = C.c_int32
    L.rrrmc_fetch_results.argtypes = [vp, vp, vp]
    L.rrrmc_last_timing.restype = C.c_int32
    L.rrrmc_last_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.rrrmc_ctx_create_quant_skn.restype = C.c_int32
    L.rrrmc_ctx_create_quant_skn.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_uint32]
    L.rrrmc_ctx_create_quant_f64.restype = C.c_int32
    L.rrrmc_ctx_create_quant_f64.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_uint32]
    L.rrrmc_quant_slice_form.restype = C.c_int32
    L.rrrmc_quant_slice_form.argtypes = [vp, C.c_int32]
    L.rrrmc_standard_mc_fast_async.restype = C.c_int32
    L.rrrmc_standard_mc_fast_async.argtypes = [vp, C.c_double, C.c_int64, C.c_int64]
    L.rrrmc_set_resume.restype = C.c_int32
    L.rrrmc_set_resume.argtypes = [vp, C.c_int32]
    L.rrrmc_set_debug_checks.restype = C.c_int32
    L.rrrmc_set_debug_checks.argtypes = [vp, C.c_int32]
    L.rrrmc_tracked_energy_f64.restype = C.c_int32
    L.rrrmc_tracked_energy_f64.argtypes = [vp, f64p]
    L.rrrmc_tracked_energy.restype = C.c_int32
    L.rrrmc_tracked_energy.argtypes = [vp, i64p]
    L.rrrmc_timing_accumulate.restype = C.c_int32
    L.rrrmc_timing_accumulate.argtypes = [vp, C.c_int32]
    L.rrrmc_timing_total.restype = C.c_int32
    L.rrrmc_timing_total.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.rrrmc_spf_team_build.restype = C.c_int32
    L.rrrmc_spf_team_build.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rrrmc_results_samples.restype = C.c_int64
    L.rrrmc_results_samples.argtypes = [vp]
    L.rrrmc_iterations_done.restype = C.c_int64
    L.rrrmc_iterations_done.argtypes = [vp]
    L.rrrmc_gen_rrg.restype = C.c_int32
    L.rrrmc_gen_rrg.argtypes = [C.c_int64, C.c_int64, C.c_uint64, i32p]
    L.rrrmc_gen_ea.restype = C.c_int32
    L.rrrmc_gen_ea.argtypes = [C.c_int64, C.c_int64, i32p]
    L.rrrmc_gen_couplings_pm1.restype = C.c_int32
    L.rrrmc_gen_couplings_pm1.argtypes = [C.c_int64, C.c_int64, i32p, C.c_uint64, i8p]
    L.rrrmc_gen_couplings_lev.restype = C.c_int32
    L.rrrmc_gen_couplings_lev.argtypes = [C.c_int64, C.c_int64, i32p, C.c_uint64, i32p, C.c_int32, i8p]
    L.rrrmc_set_graph_levels.restype = C.c_int32
    L.rrrmc_set_graph_levels.argtypes = [vp, i32p, i8p, i32p, C.c_int32, C.c_int32]
    L.rrrmc_set_couplings_dense.restype = C.c_int32
    L.rrrmc_set_couplings_dense.argtypes = [vp, f64p]
    L.rrrmc_energy_f64.restype = C.c_int32
    L.rrrmc_energy_f64.argtypes = [vp, f64p]
    L.rrrmc_get_fields_f64.restype = C.c_int32
    L.rrrmc_get_fields_f64.argtypes = [vp, f64p]
    L.rrrmc_standard_mc_f64.restype = C.c_int32
    L.rrrmc_standard_mc_f64.argtypes = [vp, C.c_double, C.c_int64, C.c_int64, vp, vp]
    L.rrrmc_fetch_results_f64.restype = C.c_int32
    L.rrrmc_fetch_results_f64.argtypes = [vp, vp, vp]
    L.rrrmc_gen_sk_gauss.restype = C.c_int32
    L.rrrmc_gen_sk_gauss.argtypes = [C.c_int64, C.c_uint64, f64p]
    L.rrrmc_ctx_create_quant.restype = C.c_int32
    L.rrrmc_ctx_create_quant_sk.restype = C.c_int32
    L.rrrmc_ctx_create_quant_sk.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_uint32]
    L.rrrmc_ctx_create_quant.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_uint32]
    L.rrrmc_quant_set_field.restype = C.c_int32
    L.rrrmc_quant_set_field.argtypes = [vp, C.c_double, C.c_double]
    L.rrrmc_rrr_mc_async.restype = C.c_int32
    L.rrrmc_rrr_mc_async.argtypes = [vp, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double]
    L.rrrmc_rrr_stats.restype = C.c_int32
    L.rrrmc_rrr_stats.argtypes = [vp, i64p]
    L.rrrmc_rrr_cache.restype = C.c_int32
    L.rrrmc_rrr_cache.argtypes = [vp, vp, vp]
    L.rrrmc_set_coloring.restype = C.c_int32
    L.rrrmc_set_coloring.argtypes = [vp, i32p, C.c_int32]
    L.rrrmc_colored_count_accepted.restype = C.c_int32
    L.rrrmc_colored_count_accepted.argtypes = [vp, C.c_int32]
    L.rrrmc_colored_sweeps_async.restype = C.c_int32
    L.rrrmc_colored_sweeps_async.argtypes = [vp, C.c_double, C.c_int64, C.c_int64]
    L.rrrmc_set_couplings_bits.restype = C.c_int32
    L.rrrmc_set_couplings_bits.argtypes = [vp, u64p]
    L.rrrmc_gen_sk_binary.restype = C.c_int32
    L.rrrmc_gen_sk_binary.argtypes = [C.c_int64, C.c_uint64, u64p]
    L.rrrmc_bkl_mc_async.restype = C.c_int32
    L.rrrmc_bkl_mc_async.argtypes = [vp, C.c_double, C.c_int64, C.c_int64]
    L.rrrmc_snapshot_reserve.restype = C.c_int32
    L.rrrmc_snapshot_reserve.argtypes = [vp, C.c_int32]
    L.rrrmc_snapshot_store.restype = C.c_int32
    L.rrrmc_snapshot_store.argtypes = [vp, C.c_int32]
    L.rrrmc_snapshot_get.restype = C.c_int32
    L.rrrmc_snapshot_get.argtypes = [vp, C.c_int32, u64p]
    L.rrrmc_overlaps.restype = C.c_int32
    L.rrrmc_overlaps.argtypes = [vp, C.c_int64, i32p, i32p, i32p]
    L.rrrmc_quant_observables.restype = C.c_int32
    L.rrrmc_quant_observables.argtypes = [vp, C.c_double, C.c_double, vp, vp, vp]
    L.rrrmc_set_graph_f64.restype = C.c_int32
    L.rrrmc_set_graph_f64.argtypes = [vp, i32p, f64p]
    L.rrrmc_gen_couplings_gauss.restype = C.c_int32
    L.rrrmc_gen_couplings_gauss.argtypes = [C.c_int64, C.c_int64, i32p, C.c_uint64, f64p]
    L.rrrmc_set_graph_discretized.restype = C.c_int32
    L.rrrmc_set_graph_discretized.argtypes = [vp, i32p, i8p, f64p, i32p, C.c_int32, C.c_int32]
    L.rrrmc_discretize.restype = C.c_int32
    L.rrrmc_discretize.argtypes = [f64p, C.c_int64, i32p, C.c_int32, i8p, f64p]
    L.rrrmc_discretize_scaled.restype = C.c_int32
    L.rrrmc_discretize_scaled.argtypes = [f64p, C.c_int64, i32p, C.c_int32, C.c_int64, C.c_double, i8p, f64p]
    L.rrrmc_set_level_scale.restype = C.c_int32
    L.rrrmc_set_level_scale.argtypes = [vp, C.c_int64, C.c_double]
    L.rrrmc_wtm_mc_async.restype = C.c_int32
    L.rrrmc_wtm_mc_async.argtypes = [vp, C.c_double, C.c_int64, C.c_double]
    L.rrrmc_wtm_times.restype = C.c_int32
    L.rrrmc_wtm_times.argtypes = [vp, f64p]
    L.rrrmc_extremal_opt_async.restype = C.c_int32
    L.rrrmc_extremal_opt_async.argtypes = [vp, f64p, C.c_int64, C.c_int64]
    L.rrrmc_extremal_opt_results.restype = C.c_int32
    L.rrrmc_extremal_opt_results.argtypes = [vp, i64p, u64p, i64p]
    L.rrrmc_extremal_opt_results_f64.restype = C.c_int32
    L.rrrmc_extremal_opt_results_f64.argtypes = [vp, f64p, u64p, i64p]
    _lib = L
    return L


def check(rc, ctx=None):
    if rc != 0:
        msg = lib().rrrmc_last_error(ctx)
        raise RRRMCError(rc, msg.decode() if msg else "")


def pinned_empty(shape, dtype):
    """An uninitialised numpy array in page-locked host memory (rrrmc_host_alloc): result buffers handed to ``Engine.standard_mc(out=...)``
    / ``Engine.get_config(out=...)`` are then filled at the bus rate.  Freed when the array (and every view of it) is gone."""
    import weakref
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = C.c_void_p()
    rc = lib().rrrmc_host_alloc(max(n, 1), C.byref(p))
    if rc != 0 or not p.value:
        raise RRRMCError(rc or 4, "rrrmc_host_alloc(%d) failed" % n)
    buf = (C.c_char * max(n, 1)).from_address(p.value)
    weakref.finalize(buf, lib().rrrmc_host_free, p.value)
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
