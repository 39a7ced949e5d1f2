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
            "rrrmc_tracked_energy_f64", "rrrmc_tracked_energy", "rrrmc_results_samples", "rrrmc_set_coloring", "rrrmc_colored_sweeps_async", "rrrmc_rrr_mc_async", "rrrmc_rrr_stats",
            "rrrmc_bkl_mc_async", "rrrmc_wtm_mc_async", "rrrmc_wtm_times", "rrrmc_extremal_opt_async", "rrrmc_extremal_opt_results",
            "rrrmc_extremal_opt_results_f64", "rrrmc_quant_observables", "rrrmc_snapshot_reserve", "rrrmc_snapshot_store", "rrrmc_overlaps",
            "rrrmc_last_error"}
    assert need <= bound, sorted(need - bound)


def test_python_table_lists_every_header_symbol():
    """rrrmc.jl_amd/_lib.py's SYMBOLS is what __graft_entry__.build() checks the library's exports against"""
    src = open(os.path.join(ROOT, "rrrmc.jl_amd", "_lib.py")).read()
    syms = set(re.findall(r'"(rrrmc_\w+)"', src[src.index("SYMBOLS = ["):src.index("]", src.index("SYMBOLS = ["))]))
    assert syms == set(header_signatures())


def _julia_src():
    return open(os.path.join(ROOT, "julia", "RRRMCHip.jl")).read()


def _method_keywords(src, name, first_arg):
    """keyword names of `function RRRMC.<name>(<first_arg>, ...; kw...)` (the text between ';' and the closing parenthesis)"""
    m = re.search(r"function RRRMC\.%s\(%s[^;]*;(.*?)\)\n" % (name, re.escape(first_arg)), src, re.S)
    assert m, "julia/RRRMCHip.jl: no method RRRMC.%s(%s, ...)" % (name, first_arg)
    kws, depth, cur = [], 0, ""
    for ch in m.group(1) + ",":           # split at top-level commas (defaults contain commas inside brackets)
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            kws.append(cur.strip()); cur = ""
        else:
            cur += ch
    return [re.match(r"(\w+)", k).group(1) for k in kws if k]


def test_reference_signature_methods_exist():
    """a reference script must run with its graph wrapped and nothing else changed — the samplers' own signatures
    (src/RRRMC.jl:81-88, 149-157, 221-229, 311-317, 376-382, 474-480) on `OnGPU`, keyword for keyword (`pp` is dead code in the reference:
    only read by commented-out lines :96-98,110-112)."""
    src = _julia_src()
    assert re.search(r"struct OnGPU\{G<:RRRMC\.Interface\.AbstractGraph\}", src) and "export OnGPU" in src
    want = {
        "standardMC": ["seed", "step", "hook", "C0", "quiet"],
        "rrrMC": ["seed", "step", "hook", "C0", "staged_thr", "staged_thr_fact", "quiet"],
        "bklMC": ["seed", "step", "hook", "C0", "quiet"],
        "wtmMC": ["seed", "step", "hook", "C0", "quiet"],
        "extremal_opt": ["seed", "step", "hook", "C0", "quiet"],
    }
    for name, kws in want.items():
        assert _method_keywords(src, name, "G::OnGPU") == kws, name
        ctx_kws = _method_keywords(src, name, "ctx::Ctx")                      # the context-first layer keeps the same names, `hook` included
        assert kws == ctx_kws, name
    # every sampler takes the reference's hook (src/RRRMC.jl:152,224,314,379,477): none refuses it, all go through the resumed pieces
    assert "no_hook" not in src and src.count("resume!(ctx, true)") >= 4 and src.count("hook = hook1(G, hook)") == 5
    # one chain returns (Es::Vector, C::Config) as RRRMC.jl:126 does: the wrappers unwrap through unwrap1 on both results
    assert src.count("return unwrap1(G, Es), unwrap1(G, Cs)") == 4 and "unwrap1(G, Cs), unwrap1(G, Emin), unwrap1(G, Cmin), unwrap1(G, itmin)" in src
    # a GraphQuant over dense slices reaches rrrmc_ctx_create_multi (VERDICT r4 missing 5)
    assert "create(QUANT_SK, Nk, 0, M, R" in src and "create(QUANT_SKN, Nk, 0, M, R" in src
    assert "has no multi-device context" not in src


def test_julia_file_is_structurally_sound():
    """No Julia here, so the cheap structural mistakes are caught by hand: block openers and `end`s balance, and no string literal directly
    follows a docstring (`"doc" \"\"\"doc2\"\"\" function f` makes Base.Docs refuse the file)."""
    src = _julia_src()
    code = re.sub(r'"""(.|\n)*?"""', '""', src)              # drop docstrings / long strings, then line comments and short strings
    code = re.sub(r'"(\\.|[^"\\\n])*"', '""', code)
    code = re.sub(r"#[^\n]*", "", code)
    # `end` used as an index inside brackets (x[1:end]) is not a block terminator
    flat, depth = [], 0
    for ch in code:
        depth += ch == "["
        depth -= ch == "]"
        flat.append(" " if depth > 0 else ch)
    code = "".join(flat)
    openers = len(re.findall(r"(?<![\w.])(function|if|for|while|try|let|do|begin|struct|module|quote)(?![\w!])", code))
    openers -= len(re.findall(r"(?<![\w.])mutable struct", code)) * 0        # `mutable struct` counts once through `struct`
    ends = len(re.findall(r"(?<![\w.:])end(?![\w!])", code))
    assert openers == ends, (openers, ends)
    # a docstring must be followed by code, not by another string literal
    assert not re.search(r'"\s*\n\s*"""', re.sub(r"#[^\n]*", "", src).replace('"""\n"""', "")), "two adjacent string literals: the first one is not a docstring"
    for m in re.finditer(r'^"[^"\n]*"\n(.*)$', src, re.M):
        assert not m.group(1).lstrip().startswith('"'), "julia/RRRMCHip.jl: docstring followed by a string literal near: " + m.group(0)[:80]
