# RRRMCHip.jl — Julia binding of the MI355X library (include/rrrmc_hip.h) behind RRRMC.jl's graph / sampler API.
#
# Drop next to src/RRRMC.jl (or `include` it from a session that has RRRMC loaded) and point RRRMC_HIP_LIB at
# rrrmc.jl_amd/lib/librrrmc_hip.so.  The file has never been executed (there is no Julia in the build image); what keeps it honest:
#   * tests/test_julia_binding.py parses every `ccall` below and checks symbol, return type, argument count and argument types against
#     include/rrrmc_hip.h (a signature drift on either side fails the CPU suite);
#   * the same ABI is driven from C (tests/abi_smoke.c, compiled against the header) and from Python (rrrmc.jl_amd/_lib.py) on the GPU;
#   * tests/replay_tape.jl checks the build's oracle against the reference itself on a machine that has Julia.
#
# One `Ctx` = the R replicas of one graph, on one device (`device = k`) or sharded over several from this one process
# (`devices = [0, 1, ...]` -> rrrmc_ctx_create_multi: the library runs one stream and one host thread per device and hands back
# gathered results).  The random streams are addressed by (seed, global replica id, iteration): results never depend on the sharding.
# Two layers of sampler methods:
#   * the reference's OWN signatures on a wrapped graph — `standardMC(OnGPU(X), β, iters; seed, step, hook, C0, quiet) -> (Es::Vector{ET},
#     C::Config)` (src/RRRMC.jl:81-88,126), likewise rrrMC / bklMC / wtmMC / extremal_opt (:149,221,311,376,474): a reference script runs
#     with ONE line changed (`X = OnGPU(X)`; INTEGRATION.md shows scripts/scripts.jl:76-101).  `OnGPU(X; replicas = R, devices = [...])`
#     runs R chains at once and returns `(Es::Matrix samples×R, Cs::Vector{Config})`.  Each call makes and finalises its own Ctx.
#   * the same with an explicit context first (`standardMC(ctx, X, β, iters; ...)`) for callers that keep the device state between calls;
#     results per replica: `Es` is samples × R (column r = the vector the reference returns for one chain), `Cs::Vector{Config}`.
# Hooks (every sampler, as src/RRRMC.jl:152,224,314,379,477): `hook(it, X, C, acc, E)` — `hook(it, X, C, E, Emin)` for extremal_opt — is
# called at the reference's sample points with the device's own tracked energy and accepted count; returning `false` ends that
# replica's run (the batch keeps running on the device; the replica's reported energy, configuration and count are those of its stop).
# Underneath, the sampler calls are RESUMED (`resume!(ctx)` = rrrmc_set_resume): pieces of (step - 1, step, step, ...) iterations for
# rrrMC / extremal_opt, of `step` iterations for bklMC (each ends at a sample point), of one sample for wtmMC — the library continues the
# run (DeltaECache / DynamicSampler / THeap / EO ranking, acc_rate, the pending bklMC draw, Emin / Cmin / itmin stay on the device), so a
# hooked run is the un-hooked run bit for bit (`HookRun`, `sample!`, `finish!` below; tests/test_gpu_hooks.py checks the Python twin of
# this logic against the oracle on the GPU).
module RRRMCHip
using RRRMC
const LIB = get(ENV, "RRRMC_HIP_LIB", "librrrmc_hip.so")
export OnGPU
const DEFAULT_SEED = 167432777111                                  # the reference's default, src/RRRMC.jl:82

check(rc, ctx = C_NULL) = rc == 0 ? nothing :
    (msg = unsafe_string(ccall((:rrrmc_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx));
     rc == 1 ? throw(ArgumentError(msg)) : error("rrrmc_hip status $rc: $msg"))

mutable struct Ctx
    p::Ptr{Cvoid}
    R::Int
    N::Int
    f64::Bool            # energies are Float64 (everything but the integer-coupling sparse graphs)
    stopped::Vector{Int} # per replica: the iteration at which its hook ended it in the last hooked standardMC (0 = none)
    function Ctx(p::Ptr{Cvoid}, R::Integer, N::Integer, f64::Bool)
        ctx = new(p, R, N, f64, zeros(Int, R))
        finalizer(c -> (c.p == C_NULL || ccall((:rrrmc_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), c.p); c.p = C_NULL), ctx)
        return ctx
    end
end

device_count() = Int(ccall((:rrrmc_device_count, LIB), Int32, ()))
version() = Int(ccall((:rrrmc_version, LIB), Int32, ()))

# model kinds of include/rrrmc_hip.h
const SPARSE_PM1, SK_NORMAL, QUANT_RRG, SK_BINARY, SPARSE_F64, SPARSE_DISCRETIZED, SPARSE_LEVELS = 1, 2, 3, 4, 5, 6, 7
const QUANT_SK, QUANT_SKN, QUANT_F64 = 8, 9, 10      # selectors of rrrmc_ctx_create_multi: GraphQuant over GraphSK / GraphSKNormal / sparse Float64 slices

# rrrmc_ctx_create / rrrmc_ctx_create_quant on one device, rrrmc_ctx_create_multi on several (N = Nk for a GraphQuant)
function create(model::Integer, N::Integer, K::Integer, M::Integer, R::Integer; device = 0, replica0 = 0, devices = nothing)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    if devices !== nothing
        ids = Int32.(collect(devices))
        GC.@preserve ids check(ccall((:rrrmc_ctx_create_multi, LIB), Int32,
                                     (Ref{Ptr{Cvoid}}, Int32, Int64, Int64, Int64, Int64, Ptr{Int32}, Int32, UInt32),
                                     ref, model, N, K, M, R, ids, length(ids), replica0))
    elseif model == QUANT_RRG
        check(ccall((:rrrmc_ctx_create_quant, LIB), Int32, (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Int64, Int32, UInt32),
                    ref, N, K, M, R, device, replica0))
    else
        check(ccall((:rrrmc_ctx_create, LIB), Int32, (Ref{Ptr{Cvoid}}, Int32, Int64, Int64, Int64, Int32, UInt32),
                    ref, model, N, K, R, device, replica0))
    end
    return ref[]
end

# X.A / X.J are Vector{NTuple{K,T}} stored inline (src/graphs/RRG.jl:118-119, EA.jl:141-142): flat N*K row-major, neighbours 1-based
flatA(X) = Int32.(reinterpret(Int, X.A) .- 1)
const PM1Graph{K} = Union{RRRMC.RRG.GraphRRG{Int,(-1,1),K}, RRRMC.EA.GraphEA{Int,(-1,1),K}}
const LevGraph{ET,LEV,K} = Union{RRRMC.RRG.GraphRRG{ET,LEV,K}, RRRMC.EA.GraphEA{ET,LEV,K}}
const F64Graph{K} = Union{RRRMC.RRG.GraphRRGNormal{K}, RRRMC.EA.GraphEANormal{K}}
const DiscGraph{ET,LEV,K} = Union{RRRMC.RRG.GraphRRGNormalDiscretized{ET,LEV,K}, RRRMC.EA.GraphEANormalDiscretized{ET,LEV,K}}
is_ea(X) = X isa RRRMC.EA.GraphEA || X isa RRRMC.EA.GraphEANormal || X isa RRRMC.EA.GraphEANormalDiscretized

# ---- GraphRRG / GraphEA with +-1 couplings (model 1: the bit-sliced kernels, BASELINE configs 1, 2, 4) --------------------------------
function Ctx(X::PM1Graph{K}, R::Integer; kw...) where {K}
    N = RRRMC.getN(X)
    ctx = Ctx(create(SPARSE_PM1, N, K, 0, R; kw...), R, N, false)
    A = flatA(X); J = Int8.(reinterpret(Int, X.J))
    GC.@preserve A J check(ccall((:rrrmc_set_graph, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int8}), ctx.p, A, J), ctx.p)
    return ctx
end

# ---- GraphRRG{ET,LEV,K} / GraphEA{ET,LEV,2D} with other levels (model 7): integer level units + their scale -------------------------
# Int levels: units = levels, scale (1, 1.0); DFloat64 (src/DFloats.jl: the Int64 payload t = round(x 10^5)): units = t ÷ g, scale (g, 1e5)
level_units(::Type{Int}, LEV) = (collect(Int, LEV), 1, 1.0)
function level_units(::Type{RRRMC.DFloats.DFloat64}, LEV)
    t = [reinterpret(Int64, l) for l in LEV]; g = max(1, gcd(t))
    return t .÷ g, g, 1e5
end
function Ctx(X::LevGraph{ET,LEV,K}, R::Integer; kw...) where {ET,LEV,K}
    N = RRRMC.getN(X)
    units, g, dv = level_units(ET, LEV)
    ctx = Ctx(create(SPARSE_LEVELS, N, K, 0, R; kw...), R, N, false)
    A = flatA(X); lev = Int32.(units)
    J = Int8.(ET === Int ? reinterpret(Int, X.J) : reinterpret(Int64, X.J) .÷ g)
    GC.@preserve A J lev check(ccall((:rrrmc_set_graph_levels, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int8}, Ptr{Int32}, Int32, Int32),
                                     ctx.p, A, J, lev, length(lev), is_ea(X) ? 1 : 0), ctx.p)
    check(ccall((:rrrmc_set_level_scale, LIB), Int32, (Ptr{Cvoid}, Int64, Float64), ctx.p, g, dv), ctx.p)
    return ctx
end

# ---- GraphRRGNormal / GraphEANormal (model 5: sparse, Float64 couplings; src/graphs/RRG.jl:503-520, EA.jl:534-552) -------------------
function Ctx(X::F64Graph{K}, R::Integer; kw...) where {K}
    N = RRRMC.getN(X)
    ctx = Ctx(create(SPARSE_F64, N, K, 0, R; kw...), R, N, true)
    A = flatA(X); J = collect(reinterpret(Float64, X.J))
    GC.@preserve A J check(ccall((:rrrmc_set_graph_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float64}), ctx.p, A, J), ctx.p)
    return ctx
end

# ---- GraphRRGNormalDiscretized / GraphEANormalDiscretized (model 6: DoubleGraph, src/graphs/RRG.jl:285-307, EA.jl:311-352) -----------
function Ctx(X::DiscGraph{ET,LEV,K}, R::Integer; kw...) where {ET,LEV,K}
    N = RRRMC.getN(X)
    units, g, dv = level_units(ET, LEV)
    ctx = Ctx(create(SPARSE_DISCRETIZED, N, K, 0, R; kw...), R, N, true)
    A = flatA(X); lev = Int32.(units); rJ = collect(reinterpret(Float64, X.rJ))
    dJ = Int8.(ET === Int ? reinterpret(Int, X.X0.J) : reinterpret(Int64, X.X0.J) .÷ g)
    GC.@preserve A dJ rJ lev check(ccall((:rrrmc_set_graph_discretized, LIB), Int32,
                                         (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int8}, Ptr{Float64}, Ptr{Int32}, Int32, Int32),
                                         ctx.p, A, dJ, rJ, lev, length(lev), is_ea(X) ? 1 : 0), ctx.p)
    check(ccall((:rrrmc_set_level_scale, LIB), Int32, (Ptr{Cvoid}, Int64, Float64), ctx.p, g, dv), ctx.p)
    return ctx
end

# ---- GraphSKNormal (model 2, BASELINE config 3) and the binary GraphSK (model 4) -----------------------------------------------------
# X.J::Vector{Vector{Float64}} (src/graphs/SK.jl:183) packs to the N×N row-major matrix the library takes
function Ctx(X::RRRMC.SK.GraphSKNormal, R::Integer; kw...)
    N = RRRMC.getN(X)
    ctx = Ctx(create(SK_NORMAL, N, 0, 0, R; kw...), R, N, true)
    Jm = Matrix{Float64}(undef, N, N); for i = 1:N; Jm[:, i] = X.J[i]; end             # column i of a Julia matrix = row i in C order
    GC.@preserve Jm check(ccall((:rrrmc_set_couplings_dense, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), ctx.p, Jm), ctx.p)
    return ctx
end
# X.J::Vector{BitVector} (SK.jl:31): row i = X.J[i].chunks
sk_bits(J::Vector{BitVector}) = (Jc = Matrix{UInt64}(undef, length(J[1].chunks), length(J)); for i = 1:length(J); Jc[:, i] = J[i].chunks; end; Jc)
function Ctx(X::RRRMC.SK.GraphSK, R::Integer; kw...)
    N = RRRMC.getN(X)
    ctx = Ctx(create(SK_BINARY, N, 0, 0, R; kw...), R, N, true)
    Jc = sk_bits(X.J)
    GC.@preserve Jc check(ccall((:rrrmc_set_couplings_bits, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt64}), ctx.p, Jc), ctx.p)
    return ctx
end

# ---- GraphQuant (BASELINE config 5): M Trotter slices of one slice graph (src/graphs/QT.jl:126-170, src/QAliases.jl) -----------------
# spins are slice-major, so C.s.chunks passes as is.  β is the one the graph's fourK was derived from (QT.jl:165); Γ is only needed
# by the observables.
function Ctx(X::RRRMC.QT.GraphQuant{fourK,G}, R::Integer, β::Real; device = 0, replica0 = 0, devices = nothing) where {fourK,G}
    X1 = X.X1[1]; Nk = X.Nk; M = X.M
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    if G <: RRRMC.SK.GraphSK
        if devices === nothing
            check(ccall((:rrrmc_ctx_create_quant_sk, LIB), Int32, (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Int32, UInt32), ref, Nk, M, R, device, replica0))
        else
            ref[] = create(QUANT_SK, Nk, 0, M, R; replica0 = replica0, devices = devices)
        end
        ctx = Ctx(ref[], R, Nk * M, true)
        Jc = sk_bits(X1.J)
        GC.@preserve Jc check(ccall((:rrrmc_set_couplings_bits, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt64}), ctx.p, Jc), ctx.p)
    elseif G <: RRRMC.SK.GraphSKNormal
        if devices === nothing
            check(ccall((:rrrmc_ctx_create_quant_skn, LIB), Int32, (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Int32, UInt32), ref, Nk, M, R, device, replica0))
        else
            ref[] = create(QUANT_SKN, Nk, 0, M, R; replica0 = replica0, devices = devices)
        end
        ctx = Ctx(ref[], R, Nk * M, true)
        Jm = Matrix{Float64}(undef, Nk, Nk); for i = 1:Nk; Jm[:, i] = X1.J[i]; end
        GC.@preserve Jm check(ccall((:rrrmc_set_couplings_dense, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), ctx.p, Jm), ctx.p)
    elseif G <: F64Graph
        # GraphQEAT = GraphQuant{fourK,GraphEANormal{twoD}} (src/QAliases.jl:50-83), and the same over a GraphRRGNormal: sparse Float64 slices
        K = length(X1.A[1])
        if devices === nothing
            check(ccall((:rrrmc_ctx_create_quant_f64, LIB), Int32, (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Int64, Int32, UInt32), ref, Nk, K, M, R, device, replica0))
        else
            ref[] = create(QUANT_F64, Nk, K, M, R; replica0 = replica0, devices = devices)
        end
        ctx = Ctx(ref[], R, Nk * M, true)
        A = flatA(X1); J = collect(reinterpret(Float64, X1.J))
        GC.@preserve A J check(ccall((:rrrmc_set_graph_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float64}), ctx.p, A, J), ctx.p)
    else
        K = length(X1.A[1])
        ctx = Ctx(create(QUANT_RRG, Nk, K, M, R; device = device, replica0 = replica0, devices = devices), R, Nk * M, true)
        check(ccall((:rrrmc_quant_slice_form, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.p, is_ea(X1) ? 1 : 0), ctx.p)
        A = flatA(X1); J = Int8.(reinterpret(Int, X1.J))
        GC.@preserve A J check(ccall((:rrrmc_set_graph, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int8}), ctx.p, A, J), ctx.p)
    end
    check(ccall((:rrrmc_quant_set_field, LIB), Int32, (Ptr{Cvoid}, Float64, Float64), ctx.p, β, fourK), ctx.p)
    return ctx
end

# ---- configurations ------------------------------------------------------------------------------------------------------------------
seed!(ctx::Ctx, seed) = seed > 0 && check(ccall((:rrrmc_seed, LIB), Int32, (Ptr{Cvoid}, UInt64), ctx.p, seed), ctx.p)   # seed ≤ 0: keep going (RRRMC.jl:89)

function set_configs!(ctx::Ctx, C0::Union{Vector{RRRMC.Config},Nothing})
    N = ctx.N; nch = (N + 63) >> 6
    chunks = Matrix{UInt64}(undef, nch, ctx.R)                   # column r = C.s.chunks of replica r
    if C0 ≡ nothing
        check(ccall((:rrrmc_init_spins_random, LIB), Int32, (Ptr{Cvoid},), ctx.p), ctx.p)
    else
        all(c -> c.N == N, C0) || throw(ArgumentError("Invalid C0, wrong N, expected $N"))     # src/RRRMC.jl:94
        length(C0) == ctx.R || throw(ArgumentError("Invalid C0, expected $(ctx.R) configurations"))
        for r = 1:ctx.R; chunks[:, r] = C0[r].s.chunks; end
        GC.@preserve chunks check(ccall((:rrrmc_set_spins, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt64}), ctx.p, chunks), ctx.p)
    end
    return chunks
end

function get_configs!(ctx::Ctx, chunks::Matrix{UInt64}, C0)
    GC.@preserve chunks check(ccall((:rrrmc_get_spins, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt64}), ctx.p, chunks), ctx.p)
    Cs = C0 ≡ nothing ? [RRRMC.Config(ctx.N, init = false) for _ = 1:ctx.R] : C0      # C0 is mutated in place (src/RRRMC.jl:93)
    for r = 1:ctx.R; Cs[r].s.chunks .= chunks[:, r]; end
    return Cs
end

"write `chunks` (column r = replica r) to the device and into the Configs"
function put_configs!(ctx::Ctx, chunks::Matrix{UInt64}, C0)
    GC.@preserve chunks check(ccall((:rrrmc_set_spins, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt64}), ctx.p, chunks), ctx.p)
    Cs = C0 ≡ nothing ? [RRRMC.Config(ctx.N, init = false) for _ = 1:ctx.R] : C0
    for r = 1:ctx.R; Cs[r].s.chunks .= chunks[:, r]; end
    return Cs
end

"energy(X, C) of every replica (src/Interface.jl:105); also rebuilds the device-side caches, as the reference's `energy` does"
function energies(ctx::Ctx)
    if ctx.f64
        E = Vector{Float64}(undef, ctx.R)
        check(ccall((:rrrmc_energy_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), ctx.p, E), ctx.p)
        return E
    end
    E = Vector{Int}(undef, ctx.R)
    check(ccall((:rrrmc_energy, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), ctx.p, E), ctx.p)
    return E
end

sync(ctx::Ctx) = check(ccall((:rrrmc_sync, LIB), Int32, (Ptr{Cvoid},), ctx.p), ctx.p)
"debug mode: after every standardMC call the library re-runs `energy` (and compares the cached fields) on the device, as the reference's
commented-out asserts in update_cache! would (src/graphs/RRG.jl:229-231, SK.jl:268-273); a mismatch throws at the next sync"
debug_checks!(ctx::Ctx, on::Bool = true) = check(ccall((:rrrmc_set_debug_checks, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.p, on ? 1 : 0), ctx.p)

# results of the last sampling call: Es (samples × R, the graph's ET), accepted (R)
function fetch(ctx::Ctx, nsamples::Integer)
    acc = Vector{Int}(undef, ctx.R)
    if ctx.f64
        Es = Matrix{Float64}(undef, nsamples, ctx.R)
        check(ccall((:rrrmc_fetch_results_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}), ctx.p, Es, acc), ctx.p)
        return Es, acc
    end
    Es = Matrix{Int}(undef, nsamples, ctx.R)
    check(ccall((:rrrmc_fetch_results, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), ctx.p, Es, acc), ctx.p)
    return Es, acc
end

# ---- standardMC (src/RRRMC.jl:81-127) -------------------------------------------------------------------------------------------------
"""
    standardMC(ctx, X, β, iters; seed, step, hook, C0, quiet) -> (Es::Matrix{ET} samples×R, Cs::Vector{Config})

For `ctx.R` replicas of `X` on the GPU(s); `seed ≤ 0` keeps the streams going, as the reference keeps the global RNG (:89).
With a `hook(it, X, Cs, accepted, E)::Bool` (the reference's hook, :61-64, handed the vectors of all replicas) the run is cut at the
hook points and RESUMED (`rrrmc_set_resume`): cache and tracked energy live on across the pieces exactly as inside one reference
call (:95-118), so a hooked run is the un-hooked chain bit for bit — for the Float64 models too — and `E` is the tracked energy.
The reference's hook ends ONE chain (`hook(...) || break`, :107): the hook may also return a `Vector{Bool}`, one flag per replica; a
replica whose flag is `false` is frozen at that sample: from then on its column of `Es` repeats the energy it had, the `Config` and the
accepted count handed to later hooks (and returned) are the ones of that moment, and `stopped_at(ctx)` tells the iteration; the others go
on; the run ends when none is left.
"""
function RRRMC.standardMC(ctx::Ctx, X::RRRMC.Interface.AbstractGraph, β::Real, iters::Integer; seed = DEFAULT_SEED, step::Integer = 1,
                          hook = nothing, C0::Union{Vector{RRRMC.Config},Nothing} = nothing, quiet = false)
    seed!(ctx, seed)
    chunks = set_configs!(ctx, C0)
    if hook ≡ nothing
        check(ccall((:rrrmc_standard_mc_async, LIB), Int32, (Ptr{Cvoid}, Float64, Int64, Int64), ctx.p, β, iters, step), ctx.p)
        sync(ctx)
        Es, acc = fetch(ctx, iters ÷ step)
        Cs = get_configs!(ctx, chunks, C0)
        quiet || println("samples = ", size(Es, 1), "\niters = ", iters, "\naccept rate = ", sum(acc) / (iters * ctx.R))
        return Es, Cs
    end
    ET = ctx.f64 ? Float64 : Int
    Es = Vector{Vector{ET}}(); accepted = zeros(Int, ctx.R); it = 0
    # a piece of n iterations that samples nothing (step n + 1): only the accepted counts are fetched
    piece!(n) = n > 0 && (check(ccall((:rrrmc_standard_mc_async, LIB), Int32, (Ptr{Cvoid}, Float64, Int64, Int64), ctx.p, β, n, n + 1), ctx.p);
                          sync(ctx); accepted .+= fetch(ctx, 0)[2])
    if ctx.f64 && iters > 0          # a call of zero iterations = the start of a reference call: E = energy(X, C), fresh cache (:95)
        check(ccall((:rrrmc_standard_mc_async, LIB), Int32, (Ptr{Cvoid}, Float64, Int64, Int64), ctx.p, β, 0, 1), ctx.p); sync(ctx)
    end
    check(ccall((:rrrmc_set_resume, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.p, 1), ctx.p)
    fill!(ctx.stopped, 0)
    frozen_chunks = copy(chunks); frozen_E = Vector{ET}(undef, ctx.R); frozen_acc = zeros(Int, ctx.R)
    try
        while it < iters
            nxt = (it ÷ step + 1) * step
            piece!(min(nxt - 1, iters) - it); it = min(nxt - 1, iters)      # up to just before the sampled iteration
            nxt > iters && break
            E = Vector{ET}(undef, ctx.R)                                    # the energy BEFORE the move of iteration nxt (:104-108)
            if ctx.f64                                                      # what the reference hands to its hook: the tracked E
                check(ccall((:rrrmc_tracked_energy_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), ctx.p, E), ctx.p)
            else
                E .= energies(ctx)                                          # integer models: recomputed == tracked, exactly
            end
            Cs = get_configs!(ctx, chunks, C0)
            for r in 1:ctx.R                                                # a frozen replica's chain has ended for the caller
                ctx.stopped[r] > 0 && (E[r] = frozen_E[r]; Cs[r].s.chunks .= frozen_chunks[:, r])
            end
            push!(Es, E)
            go = hook(nxt, X, Cs, [ctx.stopped[r] > 0 ? frozen_acc[r] : accepted[r] for r in 1:ctx.R], E)
            if go isa Bool
                go || (it = nxt; break)
            else
                nch = size(chunks, 1)
                for r in 1:ctx.R
                    if !go[r] && ctx.stopped[r] == 0                        # frozen now: keep what the reference's chain would return
                        ctx.stopped[r] = nxt
                        frozen_chunks[:, r] .= chunks[:, r]; frozen_E[r] = E[r]; frozen_acc[r] = accepted[r]
                    end
                end
                all(>(0), ctx.stopped) && (it = nxt; break)
            end
            piece!(1); it = nxt
        end
    finally
        check(ccall((:rrrmc_set_resume, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.p, 0), ctx.p)
    end
    Cs = get_configs!(ctx, chunks, C0)
    if any(>(0), ctx.stopped)                                               # frozen replicas: the configuration of their last hook, on both sides
        for r in 1:ctx.R
            ctx.stopped[r] > 0 && (chunks[:, r] .= frozen_chunks[:, r])
        end
        Cs = put_configs!(ctx, chunks, C0)
    end
    quiet || println("samples = ", length(Es), "\niters = ", it, "\naccept rate = ",
                     sum([ctx.stopped[r] > 0 ? frozen_acc[r] / ctx.stopped[r] : accepted[r] / max(it, 1) for r in 1:ctx.R]) / ctx.R)
    return isempty(Es) ? Matrix{ET}(undef, 0, ctx.R) : permutedims(reduce(hcat, Es)), Cs
end
"iteration at which replica r's hook said stop in the last hooked `standardMC` (0 = it ran to the end)"
stopped_at(ctx::Ctx) = copy(ctx.stopped)

# ---- colour-parallel sweeps (build-defined sampler for large lattices: BASELINE config 4) ---------------------------------------------
"color[x] ∈ 0:ncolors-1, adjacent sites differ (checked by the library); e.g. the checkerboard parity of an even-L lattice"
set_coloring!(ctx::Ctx, color::Vector{Int32}, ncolors::Integer) =
    GC.@preserve color check(ccall((:rrrmc_set_coloring, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}, Int32), ctx.p, color, ncolors), ctx.p)
count_accepted!(ctx::Ctx, on::Bool) = check(ccall((:rrrmc_colored_count_accepted, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.p, on ? 1 : 0), ctx.p)
"`sweeps` sweeps over the colouring (one sweep attempts every site once); an energy sample BEFORE sweep k*step -> (Es, Cs)"
function colored_sweeps(ctx::Ctx, β::Real, sweeps::Integer; seed = DEFAULT_SEED, step::Integer = 1, C0 = nothing)
    seed!(ctx, seed)
    chunks = set_configs!(ctx, C0)
    check(ccall((:rrrmc_colored_sweeps_async, LIB), Int32, (Ptr{Cvoid}, Float64, Int64, Int64), ctx.p, β, sweeps, step), ctx.p)
    sync(ctx)
    Es, _ = fetch(ctx, sweeps ÷ step)
    return Es, get_configs!(ctx, chunks, C0)
end
function checkerboard(L::Integer, D::Integer)             # parity of the lattice coordinates, in gen_EA's column-major site order (EA.jl:24-43)
    iseven(L) || throw(ArgumentError("the checkerboard needs an even L"))
    return Int32[sum(Tuple(I)) % 2 for I in CartesianIndices(ntuple(_ -> L, D))][:]
end

# ---- hooked runs of rrrMC / bklMC / wtmMC / extremal_opt --------------------------------------------------------------------------------
# Every sampler of the reference takes `hook` (src/RRRMC.jl:152,224,314,379,477) and calls it at each sample, inside its loop, with the chain's
# state of that moment (:186,255,341,404,501).  The library runs a whole call on the device, so the run is cut at the hook points and the pieces
# RESUME one another (rrrmc_set_resume, include/rrrmc_hip.h): the move-selection cache, the tracked E, rrrMC's acceptance-rate average, bklMC's
# `it` / `nextstep` and pending draw, wtmMC's heap and global time, extremal_opt's Emin / Cmin / itmin live on the device across the pieces —
# a hooked run is the un-hooked chain bit for bit (tests/test_gpu_hooks.py checks exactly that through this ABI).
resume!(ctx::Ctx, on::Bool) = check(ccall((:rrrmc_set_resume, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.p, on ? 1 : 0), ctx.p)
"samples per replica the last sampling call took (iters ÷ step for a call that starts a run; see rrrmc_set_resume for a resumed one)"
nsamples(ctx::Ctx) = Int(ccall((:rrrmc_results_samples, LIB), Int64, (Ptr{Cvoid},), ctx.p))
"the energy the last sampler call tracked — what the reference hands to its hook — read without disturbing the run (`energies` ends it)"
function run_energy(ctx::Ctx)
    if ctx.f64
        E = Vector{Float64}(undef, ctx.R)
        check(ccall((:rrrmc_tracked_energy_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), ctx.p, E), ctx.p)
        return E
    end
    E = Vector{Int}(undef, ctx.R)
    check(ccall((:rrrmc_tracked_energy, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), ctx.p, E), ctx.p)
    return E
end
"accepted moves of the last call, per replica (no energies are copied)"
function counts(ctx::Ctx)
    acc = Vector{Int}(undef, ctx.R)
    if ctx.f64
        check(ccall((:rrrmc_fetch_results_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}), ctx.p, C_NULL, acc), ctx.p)
    else
        check(ccall((:rrrmc_fetch_results, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), ctx.p, C_NULL, acc), ctx.p)
    end
    return acc
end
"staged iterations (rrrMC) / moves made (bklMC) of the last call"
function stats(ctx::Ctx)
    st = Vector{Int}(undef, ctx.R)
    check(ccall((:rrrmc_rrr_stats, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), ctx.p, st), ctx.p)
    return st
end

# Book-keeping of a hooked run over R replicas.  The reference's hook ends ONE chain (`hook(...) || break`); here the hook may return one
# `Bool` for all replicas or a `Vector{Bool}`, one flag per replica: a replica whose flag is `false` is frozen at that sample — the `Config`,
# the counts and the energy handed to later hooks (and returned) are the ones of that moment — the others go on; `stopped_at(ctx)` tells where.
mutable struct HookRun
    ctx::Ctx
    X::RRRMC.Interface.AbstractGraph
    C0::Union{Vector{RRRMC.Config},Nothing}
    chunks::Matrix{UInt64}
    frozen_chunks::Matrix{UInt64}
    kept::Dict{Symbol,Vector}
    Es::Vector{Vector}
end
function HookRun(ctx::Ctx, X, C0, chunks::Matrix{UInt64})
    fill!(ctx.stopped, 0)
    return HookRun(ctx, X, C0, chunks, copy(chunks), Dict{Symbol,Vector}(), Vector{Vector}())
end
"`live` with, for the replicas whose hook has ended them, the values they had then"
function seen(run::HookRun, name::Symbol, live::Vector)
    haskey(run.kept, name) || return copy(live)
    return [run.ctx.stopped[r] > 0 ? run.kept[name][r] : live[r] for r in 1:run.ctx.R]
end
"record the sample `E`, call `hook(it, X, Cs, args...)`; `live`: what to keep of a replica that stops here.  `false`: the run is over"
function sample!(run::HookRun, hook, it, E::Vector, args::Tuple, live::Dict{Symbol,<:Vector})
    ctx = run.ctx
    push!(run.Es, E)
    Cs = get_configs!(ctx, run.chunks, run.C0)
    for r in 1:ctx.R
        ctx.stopped[r] > 0 && (Cs[r].s.chunks .= run.frozen_chunks[:, r])
    end
    go = hook(it, run.X, Cs, args...)
    go isa Bool && return go
    for r in 1:ctx.R
        if !go[r] && ctx.stopped[r] == 0
            for (name, v) in live
                haskey(run.kept, name) || (run.kept[name] = copy(v))
                run.kept[name][r] = v[r]
            end
            run.frozen_chunks[:, r] .= run.chunks[:, r]
            ctx.stopped[r] = it isa Integer ? it : length(run.Es)
        end
    end
    return !all(>(0), ctx.stopped)
end
"(Es samples×R, Cs): the frozen replicas get the configuration of their last hook back, on both sides"
function finish!(run::HookRun, ET)
    ctx = run.ctx
    Cs = get_configs!(ctx, run.chunks, run.C0)
    if any(>(0), ctx.stopped)
        for r in 1:ctx.R
            ctx.stopped[r] > 0 && (run.chunks[:, r] .= run.frozen_chunks[:, r])
        end
        Cs = put_configs!(ctx, run.chunks, run.C0)
    end
    Es = isempty(run.Es) ? Matrix{ET}(undef, 0, ctx.R) : permutedims(reduce(hcat, run.Es))
    return Es, Cs
end

# ---- rrrMC (src/RRRMC.jl:149-219 SingleGraph, :221-290 DoubleGraph) -------------------------------------------------------------------
"""
    rrrMC(ctx, X, β, iters; seed, step, hook, C0, staged_thr, staged_thr_fact, quiet) -> (Es, Cs, accepted, staged)

Serves `rrrMC(X::SingleGraph)` — GraphRRG / GraphEA (DeltaECache{Int,L}), GraphRRGNormal / GraphEANormal / GraphSKNormal / GraphSK
(DeltaECacheCont + DynamicSampler) — and `rrrMC(X::DoubleGraph)` — GraphQuant (its fourK is the type parameter, QT.jl:126,165) and the
discretised graphs.  `staged_thr` defaults as the reference's: 0.5 for a DoubleGraph (:224), 0.8 otherwise (:152).
`hook(it, X, Cs, accepted, E)` (:186,255) is called every `step` iterations, before the move of iteration `it` (:184-188), with the vectors
of all replicas; `false` ends the run (a `Vector{Bool}`: per replica, see `HookRun`).
"""
function RRRMC.rrrMC(ctx::Ctx, X::RRRMC.Interface.AbstractGraph, β::Real, iters::Integer; seed = DEFAULT_SEED, step::Integer = 1,
                     hook = nothing, C0::Union{Vector{RRRMC.Config},Nothing} = nothing,
                     staged_thr::Real = X isa RRRMC.Interface.DoubleGraph ? 0.5 : 0.8, staged_thr_fact::Real = 5.0, quiet = false)
    isfinite(β) || throw(ArgumentError("β must be finite, given: $β"))
    fourK = X isa RRRMC.QT.GraphQuant ? typeof(X).parameters[1] : 0.0
    seed!(ctx, seed)
    chunks = set_configs!(ctx, C0)
    call!(n) = (check(ccall((:rrrmc_rrr_mc_async, LIB), Int32, (Ptr{Cvoid}, Float64, Float64, Int64, Int64, Float64, Float64),
                            ctx.p, β, fourK, n, step, staged_thr, staged_thr_fact), ctx.p); sync(ctx))
    if hook ≡ nothing
        call!(iters)
        Es, acc = fetch(ctx, nsamples(ctx))
        staged = stats(ctx)
        quiet || println("samples = ", size(Es, 1), "\niters = ", iters, "\naccept rate = ", sum(acc) / (iters * ctx.R),
                         "\nfrac. staged iters = ", sum(staged) / (iters * ctx.R))                     # RRRMC.jl:284-288
        return Es, get_configs!(ctx, chunks, C0), acc, staged
    end
    run = HookRun(ctx, X, C0, chunks)
    acc = zeros(Int, ctx.R); staged = zeros(Int, ctx.R)
    piece!(n) = (call!(n); acc .+= counts(ctx); staged .+= stats(ctx))
    resume!(ctx, false)
    it = min(step - 1, iters)
    piece!(it)                                   # a fresh run (energy(X, C), gen_ΔEcache: :177-178), up to just before the first sampled iteration
    resume!(ctx, true)
    try
        while it + 1 ≤ iters
            E = run_energy(ctx)
            if !sample!(run, hook, it + 1, seen(run, :E, E), (seen(run, :acc, acc), seen(run, :E, E)), Dict(:acc => acc, :E => E))
                it += 1                          # the reference has counted the iteration its hook ended (:183)
                break
            end
            n = min(step, iters - it)            # the move of the sampled iteration and the step - 1 after it
            piece!(n); it += n
        end
    finally
        resume!(ctx, false)
    end
    Es, Cs = finish!(run, ctx.f64 ? Float64 : Int)
    acc = seen(run, :acc, acc)
    quiet || println("samples = ", size(Es, 1), "\niters = ", it, "\naccept rate = ", sum(acc) / (max(it, 1) * ctx.R),
                     "\nfrac. staged iters = ", sum(staged) / (max(it, 1) * ctx.R))
    return Es, Cs, acc, staged
end

# ---- bklMC (src/RRRMC.jl:311-359): `iters` counts the skipped rejections too; `moves` = the moves actually made ("true it") ------------
"`hook(nextstep, X, Cs, accepted, E)` (:341) at every sample point the skipped iterations pass: a resumed call of `step` iterations ends there"
function RRRMC.bklMC(ctx::Ctx, X::RRRMC.Interface.AbstractGraph, β::Real, iters::Integer; seed = DEFAULT_SEED, step::Integer = 1,
                     hook = nothing, C0::Union{Vector{RRRMC.Config},Nothing} = nothing, quiet = false)
    seed!(ctx, seed)
    chunks = set_configs!(ctx, C0)
    call!(n) = (check(ccall((:rrrmc_bkl_mc_async, LIB), Int32, (Ptr{Cvoid}, Float64, Int64, Int64), ctx.p, β, n, step), ctx.p); sync(ctx))
    if hook ≡ nothing || iters < step
        call!(iters)
        Es, _ = fetch(ctx, nsamples(ctx))
        moves = stats(ctx)
        quiet || println("samples = ", size(Es, 1), "\niters = ", iters, "\ntrue it = ", sum(moves) / ctx.R)
        return Es, get_configs!(ctx, chunks, C0), moves
    end
    run = HookRun(ctx, X, C0, chunks)
    moves = zeros(Int, ctx.R)
    resume!(ctx, false)
    try
        for k = 1:(iters ÷ step)                 # the reference goes `out` after the last sample (:343): nothing runs behind it
            call!(step); resume!(ctx, true)
            moves .+= stats(ctx)
            E = run_energy(ctx)
            sample!(run, hook, k * step, seen(run, :E, E), (seen(run, :acc, moves), seen(run, :E, E)), Dict(:acc => moves, :E => E)) || break
        end
    finally
        resume!(ctx, false)
    end
    Es, Cs = finish!(run, ctx.f64 ? Float64 : Int)
    moves = seen(run, :acc, moves)
    quiet || println("samples = ", size(Es, 1), "\niters = ", iters, "\ntrue it = ", sum(moves) / ctx.R)
    return Es, Cs, moves
end

# ---- wtmMC (src/RRRMC.jl:376-426, src/WaitingTimes.jl): `step::Float64` in sweeps, `samples` energies at global times k*step/N ---------
"`hook(nextstep, X, Cs, num_moves, E)` (:404) at every sample time (`nextstep`: k additions of step / N, as :391,405 accumulate it)"
function RRRMC.wtmMC(ctx::Ctx, X::RRRMC.Interface.AbstractGraph, β::Real, samples::Integer; seed = DEFAULT_SEED, step::Float64 = 1.0,
                     hook = nothing, C0::Union{Vector{RRRMC.Config},Nothing} = nothing, quiet = false)
    seed!(ctx, seed)
    chunks = set_configs!(ctx, C0)
    call!(n) = (check(ccall((:rrrmc_wtm_mc_async, LIB), Int32, (Ptr{Cvoid}, Float64, Int64, Float64), ctx.p, β, n, step), ctx.p); sync(ctx))
    times() = (t = Vector{Float64}(undef, ctx.R); check(ccall((:rrrmc_wtm_times, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), ctx.p, t), ctx.p); t)
    if hook ≡ nothing
        call!(samples)
        Es, moves = fetch(ctx, nsamples(ctx))
        t = times()
        quiet || println("samples = ", size(Es, 1), "\nnum. moves = ", sum(moves) / ctx.R, "\nglobal time = ", sum(t) / ctx.R)     # :419-423
        return Es, get_configs!(ctx, chunks, C0), moves, t
    end
    run = HookRun(ctx, X, C0, chunks)
    moves = zeros(Int, ctx.R); t = zeros(Float64, ctx.R)
    st = step / ctx.N                            # :391
    nextstep = st
    resume!(ctx, false)
    try
        for _ = 1:samples
            call!(1); resume!(ctx, true)
            moves .+= counts(ctx); t = times()
            E = run_energy(ctx)
            sample!(run, hook, nextstep, seen(run, :E, E), (seen(run, :acc, moves), seen(run, :E, E)), Dict(:acc => moves, :E => E, :t => t)) || break
            nextstep += st                       # :405
        end
    finally
        resume!(ctx, false)
    end
    Es, Cs = finish!(run, ctx.f64 ? Float64 : Int)
    moves = seen(run, :acc, moves); t = seen(run, :t, t)
    quiet || println("samples = ", size(Es, 1), "\nnum. moves = ", sum(moves) / ctx.R, "\nglobal time = ", sum(t) / ctx.R)
    return Es, Cs, moves, t
end

# ---- extremal_opt (src/RRRMC.jl:474-521): EOCache{Int,L} on the DiscrGraphs, the generic EOCacheCont elsewhere -------------------------
"""-> (Cs, Emin, Cmin::Vector{Config}, itmin) per replica, the reference's return tuple (:520); `Es` (what the hook sees) as 5th value.
`hook(it, X, Cs, E, Emin)` (:501 — not the samplers' signature) every `step` iterations, before the move of iteration `it`."""
function RRRMC.extremal_opt(ctx::Ctx, X::RRRMC.Interface.AbstractGraph, τ::Real, iters::Integer; seed = DEFAULT_SEED, step::Integer = 1,
                            hook = nothing, C0::Union{Vector{RRRMC.Config},Nothing} = nothing, quiet = false)
    N = ctx.N; nch = (N + 63) >> 6
    ftau = cumsum([j^(-τ) for j = 1:N])                                # as DeltaE.jl:444-445 computes it
    seed!(ctx, seed)
    chunks = set_configs!(ctx, C0)
    call!(n) = (GC.@preserve ftau check(ccall((:rrrmc_extremal_opt_async, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64), ctx.p, ftau, n, step), ctx.p); sync(ctx))
    function results()                                                  # Emin, Cmin (chunks), itmin of the run so far
        cmin = Matrix{UInt64}(undef, nch, ctx.R); itmin = Vector{Int}(undef, ctx.R)
        if ctx.f64
            Emin = Vector{Float64}(undef, ctx.R)
            check(ccall((:rrrmc_extremal_opt_results_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{UInt64}, Ptr{Int64}), ctx.p, Emin, cmin, itmin), ctx.p)
        else
            Emin = Vector{Int}(undef, ctx.R)
            check(ccall((:rrrmc_extremal_opt_results, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{UInt64}, Ptr{Int64}), ctx.p, Emin, cmin, itmin), ctx.p)
        end
        return Emin, cmin, itmin
    end
    ET = ctx.f64 ? Float64 : Int
    it = iters
    if hook ≡ nothing
        call!(iters)
        Es, _ = fetch(ctx, nsamples(ctx))
        Emin, cmin, itmin = results()
        Cs = get_configs!(ctx, chunks, C0)
    else
        run = HookRun(ctx, X, C0, chunks)
        resume!(ctx, false)
        it = min(step - 1, iters)
        call!(it)
        Emin, cmin, itmin = results()
        resume!(ctx, true)
        try
            while it + 1 ≤ iters
                E = run_energy(ctx)
                live = Dict(:E => E, :Emin => Emin, :itmin => itmin, :cmin => [cmin[:, r] for r in 1:ctx.R])
                if !sample!(run, hook, it + 1, seen(run, :E, E), (seen(run, :E, E), seen(run, :Emin, Emin)), live)
                    it += 1
                    break
                end
                n = min(step, iters - it)
                call!(n); it += n
                Emin, cmin, itmin = results()
            end
        finally
            resume!(ctx, false)
        end
        Es, Cs = finish!(run, ET)
        Emin = seen(run, :Emin, Emin); itmin = seen(run, :itmin, itmin)
        kept = seen(run, :cmin, [cmin[:, r] for r in 1:ctx.R])
        for r = 1:ctx.R; cmin[:, r] .= kept[r]; end
    end
    Cmin = [RRRMC.Config(N, init = false) for _ = 1:ctx.R]
    for r = 1:ctx.R; Cmin[r].s.chunks .= cmin[:, r]; end
    quiet || println("iters = ", it, "\nmin [it = ", itmin, "] = ", Emin)                             # :516-519
    return Cs, Emin, Cmin, itmin, Es
end

# ---- GraphQuant observables of the live configuration (src/graphs/QT.jl:113-122, 213-268) ---------------------------------------------
function quant_observables(ctx::Ctx, X::RRRMC.QT.GraphQuant, β::Real, Γ::Real)
    Q = Vector{Float64}(undef, ctx.R); tm = similar(Q); ovs = Matrix{Float64}(undef, X.M ÷ 2, ctx.R)
    check(ccall((:rrrmc_quant_observables, LIB), Int32, (Ptr{Cvoid}, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                ctx.p, β, Γ, Q, tm, ovs), ctx.p)
    return Q, tm, ovs                                                   # Qenergy, transverse_mag, overlaps (column r = replica r)
end

# ---- device-side snapshots for the scripts' hooks (scripts/scripts.jl:51-69: copy(C.s) per sample, pm1dot / parseovs afterwards) -------
snapshot_reserve(ctx::Ctx, n) = check(ccall((:rrrmc_snapshot_reserve, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.p, n), ctx.p)
snapshot_store(ctx::Ctx, slot) = check(ccall((:rrrmc_snapshot_store, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.p, slot), ctx.p)
function snapshot_get(ctx::Ctx, slot)                                    # a column of to_mat's BitMatrix per replica (:13-21)
    chunks = Matrix{UInt64}(undef, (ctx.N + 63) >> 6, ctx.R)
    check(ccall((:rrrmc_snapshot_get, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{UInt64}), ctx.p, slot, chunks), ctx.p)
    return chunks
end
"q[r, p] = pm1dot(replica r of slot ia[p], replica r of slot ib[p]) (0-based slots, -1 = the live configuration)"
function overlaps(ctx::Ctx, ia::Vector{Int32}, ib::Vector{Int32})
    q = Matrix{Int32}(undef, ctx.R, length(ia))
    GC.@preserve ia ib check(ccall((:rrrmc_overlaps, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}), ctx.p, length(ia), ia, ib, q), ctx.p)
    return q
end

# ---- cache views for parity checks against the reference's own objects ----------------------------------------------------------------
"X.cache.lfields of every replica (N × R), recomputed from the current spins for the integer models (src/Common.jl:27-36)"
function fields(ctx::Ctx, X)
    if X isa RRRMC.SK.GraphSKNormal || X isa F64Graph
        lf = Matrix{Float64}(undef, ctx.N, ctx.R)
        check(ccall((:rrrmc_get_fields_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), ctx.p, lf), ctx.p)
        return lf
    end
    lf = Matrix{Int}(undef, ctx.N, ctx.R)
    check(ccall((:rrrmc_get_fields, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), ctx.p, lf), ctx.p)
    return lf
end
"DeltaECache.pos (N × R, 0-based class a + 2 up) and the class sizes after the last rrrMC call on a DoubleGraph (src/DeltaE.jl:63-73)"
function rrr_cache(ctx::Ctx, X)
    stride = X isa RRRMC.QT.GraphQuant ? 4 : 16
    pos = Matrix{Int8}(undef, ctx.N, ctx.R); sizes = Matrix{Int32}(undef, stride, ctx.R)
    check(ccall((:rrrmc_rrr_cache, LIB), Int32, (Ptr{Cvoid}, Ptr{Int8}, Ptr{Int32}), ctx.p, pos, sizes), ctx.p)
    return pos, sizes
end

"""
    host_alloc(T, dims...) -> Array{T}

Result buffers in page-locked host memory (`rrrmc_host_alloc`): `fetch_results!` / `get_spins!` fill them at the bus rate.  Release with
`host_free(A)`; the array must not be used afterwards.
"""
function host_alloc(::Type{T}, dims::Integer...) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:rrrmc_host_alloc, LIB), Int32, (Int64, Ref{Ptr{Cvoid}}), Int64(prod(dims) * sizeof(T)), p)
    rc == 0 || error("rrrmc_host_alloc failed with status $rc")
    return unsafe_wrap(Array, Ptr{T}(p[]), dims; own = false)
end
host_free(A::Array) = (ccall((:rrrmc_host_free, LIB), Int32, (Ptr{Cvoid},), pointer(A)); nothing)

"kernel time of the last sampling call: (total ms, dominant-kernel ms, its launches); the slowest device of a multi-device Ctx"
function last_timing(ctx::Ctx)
    tot = Ref{Float64}(0.0); sw = Ref{Float64}(0.0); nl = Ref{Int32}(0)
    check(ccall((:rrrmc_last_timing, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}, Ref{Float64}, Ref{Int32}), ctx.p, tot, sw, nl), ctx.p)
    return tot[], sw[], Int(nl[])
end

# ---- the reference's own signatures: wrap the graph, change nothing else ----------------------------------------------------------------
"""
    OnGPU(X; replicas = 1, device = 0, devices = nothing, replica0 = 0)

`standardMC(OnGPU(X), β, iters; seed, step, hook, C0, quiet)` is the reference's `standardMC(X, β, iters; ...)` (src/RRRMC.jl:81-88) run by
the library: same arguments, same keywords, same return value `(Es::Vector{ET}, C::Config)` (:126), `C0` resumed and mutated in place (:93),
the hook called as `hook(it, X, C, accepted, E)` with the UNWRAPPED graph (:107).  `rrrMC`, `bklMC`, `wtmMC` and `extremal_opt` likewise
(:149-157, 221-229, 311, 376, 474).  With `replicas = R > 1` the call runs R independent chains (replica ids `replica0 .+ (0:R-1)` address
the random streams) and returns `(Es::Matrix{ET} samples×R, Cs::Vector{Config})`; `C0` is then a `Vector{Config}` and the hook gets the
vectors of all replicas (see the context-first `standardMC`).  `devices = [0, 1, ...]` shards the replicas over several GPUs from this one
process.  Every call makes its own context and finalises it before returning.
"""
struct OnGPU{G<:RRRMC.Interface.AbstractGraph}
    X::G
    replicas::Int
    device::Int
    devices::Union{Nothing,Vector{Int}}
    replica0::Int
end
OnGPU(X::RRRMC.Interface.AbstractGraph; replicas::Integer = 1, device::Integer = 0, devices = nothing, replica0::Integer = 0) =
    OnGPU{typeof(X)}(X, replicas, device, devices === nothing ? nothing : collect(Int, devices), replica0)
RRRMC.getN(G::OnGPU) = RRRMC.getN(G.X)

# one context per call (β is needed by a GraphQuant's context only)
function with_ctx(f, G::OnGPU, β::Real)
    ctx = if G.X isa RRRMC.QT.GraphQuant
        Ctx(G.X, G.replicas, β; device = G.device, replica0 = G.replica0, devices = G.devices)
    else
        Ctx(G.X, G.replicas; device = G.device, replica0 = G.replica0, devices = G.devices)
    end
    try
        return f(ctx)
    finally
        finalize(ctx)
    end
end
single(G::OnGPU) = G.replicas == 1
configs_in(G::OnGPU, C0) = C0 ≡ nothing ? nothing : (C0 isa RRRMC.Config ? RRRMC.Config[C0] : C0)
# one chain: a Vector and a Config, as the reference returns them; several: the matrix and the vector of Configs
unwrap1(G::OnGPU, Es::AbstractMatrix) = single(G) ? Es[:, 1] : Es
unwrap1(G::OnGPU, v::AbstractVector) = single(G) ? v[1] : v
# the hook of ONE chain takes scalars and the Config (src/RRRMC.jl:61-64); the context-first layer hands vectors over the replicas
hook1(G::OnGPU, hook) = hook ≡ nothing || !single(G) ? hook : (it, X, Cs, a, b) -> hook(it, X, Cs[1], a[1], b[1])

function RRRMC.standardMC(G::OnGPU, β::Real, iters::Integer; seed = DEFAULT_SEED, step::Integer = 1, hook = nothing,
                          C0::Union{RRRMC.Config,Vector{RRRMC.Config},Nothing} = nothing, quiet::Bool = false)
    with_ctx(G, β) do ctx
        Es, Cs = RRRMC.standardMC(ctx, G.X, β, iters; seed = seed, step = step, hook = hook1(G, hook), C0 = configs_in(G, C0), quiet = quiet)
        return unwrap1(G, Es), unwrap1(G, Cs)
    end
end

function RRRMC.rrrMC(G::OnGPU, β::Real, iters::Integer; seed = DEFAULT_SEED, step::Integer = 1, hook = nothing,
                     C0::Union{RRRMC.Config,Vector{RRRMC.Config},Nothing} = nothing,
                     staged_thr::Real = G.X isa RRRMC.Interface.DoubleGraph ? 0.5 : NaN, staged_thr_fact::Real = 5.0, quiet::Bool = false)
    thr = isnan(staged_thr) ? (G.X isa RRRMC.Interface.DiscrGraph ? 0.5 : 0.8) : staged_thr          # src/RRRMC.jl:162-164
    with_ctx(G, β) do ctx
        Es, Cs, _, _ = RRRMC.rrrMC(ctx, G.X, β, iters; seed = seed, step = step, hook = hook1(G, hook), C0 = configs_in(G, C0), staged_thr = thr,
                                   staged_thr_fact = staged_thr_fact, quiet = quiet)
        return unwrap1(G, Es), unwrap1(G, Cs)
    end
end

function RRRMC.bklMC(G::OnGPU, β::Real, iters::Integer; seed = DEFAULT_SEED, step::Integer = 1, hook = nothing,
                     C0::Union{RRRMC.Config,Vector{RRRMC.Config},Nothing} = nothing, quiet::Bool = false)
    with_ctx(G, β) do ctx
        Es, Cs, _ = RRRMC.bklMC(ctx, G.X, β, iters; seed = seed, step = step, hook = hook1(G, hook), C0 = configs_in(G, C0), quiet = quiet)
        return unwrap1(G, Es), unwrap1(G, Cs)
    end
end

function RRRMC.wtmMC(G::OnGPU, β::Real, samples::Integer; seed = DEFAULT_SEED, step::Float64 = 1.0, hook = nothing,
                     C0::Union{RRRMC.Config,Vector{RRRMC.Config},Nothing} = nothing, quiet::Bool = false)
    with_ctx(G, β) do ctx
        Es, Cs, _, _ = RRRMC.wtmMC(ctx, G.X, β, samples; seed = seed, step = step, hook = hook1(G, hook), C0 = configs_in(G, C0), quiet = quiet)
        return unwrap1(G, Es), unwrap1(G, Cs)
    end
end

"-> (C, Emin, Cmin, itmin), the reference's return tuple (src/RRRMC.jl:520); vectors of them for several replicas"
function RRRMC.extremal_opt(G::OnGPU, τ::Real, iters::Integer; seed = DEFAULT_SEED, step::Integer = 1, hook = nothing,
                            C0::Union{RRRMC.Config,Vector{RRRMC.Config},Nothing} = nothing, quiet::Bool = false)
    with_ctx(G, 1.0) do ctx
        Cs, Emin, Cmin, itmin, _ = RRRMC.extremal_opt(ctx, G.X, τ, iters; seed = seed, step = step, hook = hook1(G, hook), C0 = configs_in(G, C0), quiet = quiet)
        return unwrap1(G, Cs), unwrap1(G, Emin), unwrap1(G, Cmin), unwrap1(G, itmin)
    end
end

end # module
