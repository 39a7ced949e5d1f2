"""The Julia binding cannot run here (no Julia in the image), so its contact surface with the library is checked statically: every
`ccall` in julia/*.jl must name a function that include/rrrmc_hip.h declares, with the header's return type, argument count and
argument types.  A signature drift on either side fails the CPU suite instead of crashing a maintainer's Julia session."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Julia ccall type -> the C types it may stand for (const-ness ignored)
J2C = {
    "Int32": {"int32_t"}, "Int64": {"int64_t"}, "UInt32": {"uint32_t"}, "UInt64": {"uint64_t"}, "Float64": {"double"},
    "Cvoid": {"void"}, "Cstring": {"char*"},
    "Ptr{Cvoid}": {"rrrmc_ctx*", "void*"}, "Ref{Ptr{Cvoid}}": {"rrrmc_ctx**", "void**"},
    "Ptr{Int8}": {"int8_t*"}, "Ptr{Int32}": {"int32_t*"}, "Ptr{Int64}": {"int64_t*"}, "Ptr{UInt64}": {"uint64_t*"},
    "Ptr{Float64}": {"double*"}, "Ref{Float64}": {"double*"}, "Ref{Int32}": {"int32_t*"}, "Ref{Int64}": {"int64_t*"},
}


def c_type(t):
    t = re.sub(r"\bconst\b", "", t)
    t = re.sub(r"\s+", "", t)
    return t


def header_signatures():
    src = open(os.path.join(ROOT, "include", "rrrmc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    sigs = {}
    for m in re.finditer(r"RRRMC_API\s+([\w\s\*]+?)\s*\b(rrrmc_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else []
        types = []
        for a in args:
            mm = re.match(r"(.*?)(\w+)$", a)          # strip the parameter name
            types.append(c_type(mm.group(1)))
        sigs[name] = (c_type(ret), types)
    return sigs


def julia_ccalls(path):
    src = open(path).read()
    src = re.sub(r"#[^\n]*", "", src)
    out = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*LIB\),\s*([\w{}]+),\s*\(([^)]*)\)", src):
        name, ret, args = m.group(1), m.group(2), m.group(3)
        types = [a.strip() for a in args.split(",") if a.strip()]
        out.append((name, ret, types, src.count("\n", 0, m.start()) + 1))
    return out


def test_header_parses_completely():
    sigs = header_signatures()
    decl = re.findall(r"RRRMC_API[^;]*?\b(rrrmc_\w+)\s*\(", open(os.path.join(ROOT, "include", "rrrmc_hip.h")).read())
    assert sorted(sigs) == sorted(set(decl)) and len(sigs) >= 67


@pytest.mark.parametrize("jl", ["RRRMCHip.jl"])
def test_every_ccall_matches_the_header(jl):
    sigs = header_signatures()
    calls = julia_ccalls(os.path.join(ROOT, "julia", jl))
    assert len(calls) >= 45
    for name, ret, types, line in calls:
        where = "julia/%s:%d ccall(:%s)" % (jl, line, name)
        assert name in sigs, where + ": not declared in include/rrrmc_hip.h"
        cret, ctypes_ = sigs[name]
        assert ret in J2C, where + ": unknown Julia return type " + ret
        assert cret in J2C[ret], "%s: returns %s in Julia, %s in the header" % (where, ret, cret)
        assert len(types) == len(ctypes_), "%s: %d arguments in Julia, %d in the header" % (where, len(types), len(ctypes_))
        for k, (jt, ct) in enumerate(zip(types, ctypes_)):
            assert jt in J2C, "%s: unknown Julia type %s" % (where, jt)
            assert ct in J2C[jt], "%s: argument %d is %s in Julia, %s in the header" % (where, k + 1, jt, ct)


def test_binding_covers_the_boundary():
    """what SURVEY.md §8b asks the glue to reach: every sampler, every graph family's constructor calls, the multi-device context,
    the colour sweeps of BASELINE config 4, and the resumed (hooked) run of the Float64 models"""
    bound = {c[0] for c in julia_ccalls(os.path.join(ROOT, "julia", "RRRMCHip.jl"))}
    need = {"rrrmc_ctx_create", "rrrmc_ctx_create_multi", "rrrmc_ctx_create_quant", "rrrmc_ctx_create_quant_sk", "rrrmc_ctx_create_quant_skn",
            "rrrmc_ctx_destroy", "rrrmc_set_graph", "rrrmc_set_graph_f64", "rrrmc_set_graph_levels", "rrrmc_set_graph_discretized",
            "rrrmc_set_level_scale", "rrrmc_set_couplings_dense", "rrrmc_set_couplings_bits", "rrrmc_quant_set_field", "rrrmc_quant_slice_form",
            "rrrmc_seed", "rrrmc_init_spins_random", "rrrmc_set_spins", "rrrmc_get_spins", "rrrmc_energy", "rrrmc_energy_f64",
            "rrrmc_standard_mc_async", "rrrmc_sync", "rrrmc_fetch_results", "rrrmc_fetch_results_f64", "rrrmc_set_resume",
            "rrrmc_tracked_energy_f64", "rrrmc_set_coloring", "rrrmc_colored_sweeps_async", "rrrmc_rrr_mc_async", "rrrmc_rrr_stats",
            "rrrmc_bkl_mc_async", "rrrmc_wtm_mc_async", "rrrmc_wtm_times", "rrrmc_extremal_opt_async", "rrrmc_extremal_opt_results",
            "rrrmc_extremal_opt_results_f64", "rrrmc_quant_observables", "rrrmc_snapshot_reserve", "rrrmc_snapshot_store", "rrrmc_overlaps",
            "rrrmc_last_error"}
    assert need <= bound, sorted(need - bound)


def test_python_table_lists_every_header_symbol():
    """rrrmc.jl_amd/_lib.py's SYMBOLS is what __graft_entry__.build() checks the library's exports against"""
    src = open(os.path.join(ROOT, "rrrmc.jl_amd", "_lib.py")).read()
    syms = set(re.findall(r'"(rrrmc_\w+)"', src[src.index("SYMBOLS = ["):src.index("]", src.index("SYMBOLS = ["))]))
    assert syms == set(header_signatures())
