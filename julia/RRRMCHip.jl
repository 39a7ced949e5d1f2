# RRRMCHip.jl — Julia binding of the MI355X library (include/rrrmc_hip.h) behind RRRMC.jl's graph / sampler API.
#
# Drop next to src/RRRMC.jl (or `include` it from a session that has RRRMC loaded) and point RRRMC_HIP_LIB at
# rrrmc.jl_amd/lib/librrrmc_hip.so.  Every ccall below is also exercised, with the same argument types, by the Python ctypes
# binding rrrmc.jl_amd/_lib.py, which is what the parity tests drive; this file itself has never been executed (there is no Julia
# in the build image) — julia/replay_tape.jl is the companion that checks the build's oracle against the reference on a machine
# that has Julia.
#
# One `Ctx` = one device (`device = k`); replicas of a multi-GPU job are sharded by global replica id (`replica0`), e.g. one Ctx per
# device from one Julia task each (`Threads.@spawn`) or one process per GPU — the streams are addressed by (seed, global replica,
# iteration), so the results do not depend on the sharding.
module RRRMCHip
using RRRMC
const LIB = get(ENV, "RRRMC_HIP_LIB", "librrrmc_hip.so")
const DEFAULT_SEED = 167432777111

check(rc, ctx = C_NULL) = rc == 0 ? nothing :
    (msg = unsafe_string(ccall((:rrrmc_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx));
     rc == 1 ? throw(ArgumentError(msg)) : error("rrrmc_hip status $rc: $msg"))

mutable struct Ctx
    p::Ptr{Cvoid}
    R::Int
    function Ctx(p::Ptr{Cvoid}, R::Integer)
        ctx = new(p, R)
        finalizer(c -> (c.p == C_NULL || ccall((:rrrmc_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), c.p); c.p = C_NULL), ctx)
        return ctx
    end
end

device_count() = Int(ccall((:rrrmc_device_count, LIB), Int32, ()))

# ---- GraphRRG / GraphEA with +-1 couplings (model 1) -------------------------------------------------------------------------
# X.A / X.J are Vector{NTuple{K,Int}} stored inline (src/graphs/RRG.jl:118-119, EA.jl:141-142): flat N*K row-major, 1-based
function Ctx(X::Union{RRRMC.RRG.GraphRRG{Int,(-1,1),K}, RRRMC.EA.GraphEA{Int,(-1,1),K}}, R::Integer;
             device = 0, replica0 = 0) where {K}
    N = RRRMC.getN(X)
    A = Int32.(reinterpret(Int, X.A) .- 1)
    J = Int8.(reinterpret(Int, X.J))
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rrrmc_ctx_create, LIB), Int32, (Ref{Ptr{Cvoid}}, Int32, Int64, Int64, Int64, Int32, UInt32),
                ref, 1, N, K, R, device, replica0))
    ctx = Ctx(ref[], R)
    GC.@preserve A J check(ccall((:rrrmc_set_graph, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int8}), ctx.p, A, J), ctx.p)
    return ctx
end

function set_configs!(ctx::Ctx, N::Integer, C0::Union{Vector{RRRMC.Config},Nothing})
    nch = (N + 63) >> 6
    chunks = Matrix{UInt64}(undef, nch, ctx.R)                   # column r = C.s.chunks of replica r
    if C0 ≡ nothing
        check(ccall((:rrrmc_init_spins_random, LIB), Int32, (Ptr{Cvoid},), ctx.p), ctx.p)
    else
        all(c -> c.N == N, C0) || throw(ArgumentError("Invalid C0, wrong N, expected $N"))     # src/RRRMC.jl:94
        length(C0) == ctx.R || throw(ArgumentError("Invalid C0, expected $(ctx.R) configurations"))
        for r = 1:ctx.R; chunks[:, r] = C0[r].s.chunks; end
        check(ccall((:rrrmc_set_spins, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt64}), ctx.p, chunks), ctx.p)
    end
    return chunks
end

function get_configs!(ctx::Ctx, N::Integer, chunks::Matrix{UInt64}, C0)
    check(ccall((:rrrmc_get_spins, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt64}), ctx.p, chunks), ctx.p)
    Cs = C0 ≡ nothing ? [RRRMC.Config(N, init = false) for _ = 1:ctx.R] : C0      # C0 is mutated in place (src/RRRMC.jl:93)
    for r = 1:ctx.R; Cs[r].s.chunks .= chunks[:, r]; end
    return Cs
end

"""
    standardMC(ctx, X, β, iters; seed, step, C0, quiet) -> (Es::Matrix{Int} samples×R, Cs::Vector{Config})

`standardMC` (src/RRRMC.jl:81-127) for `ctx.R` replicas of `X` on the GPU; column `r` of `Es` is replica `r`'s vector.
`seed ≤ 0` keeps the streams going, as the reference keeps the global RNG (:89).
"""
function RRRMC.standardMC(ctx::Ctx, X::RRRMC.Interface.DiscrGraph{Int}, β::Real, iters::Integer; seed = DEFAULT_SEED, step::Integer = 1,
                          C0::Union{Vector{RRRMC.Config},Nothing} = nothing, quiet = false)
    N = RRRMC.getN(X)
    seed > 0 && check(ccall((:rrrmc_seed, LIB), Int32, (Ptr{Cvoid}, UInt64), ctx.p, seed), ctx.p)
    chunks = set_configs!(ctx, N, C0)
    Es = Matrix{Int}(undef, iters ÷ step, ctx.R)
    acc = Vector{Int}(undef, ctx.R)
    check(ccall((:rrrmc_standard_mc, LIB), Int32, (Ptr{Cvoid}, Float64, Int64, Int64, Ptr{Int64}, Ptr{Int64}),
                ctx.p, β, iters, step, Es, acc), ctx.p)
    Cs = get_configs!(ctx, N, chunks, C0)
    quiet || println("samples = ", size(Es, 1), "\niters = ", iters, "\naccept rate = ", sum(acc) / (iters * ctx.R))
    return Es, Cs
end

"""
    standardMC_hooked(ctx, X, β, iters; step, hook, ...) — the reference's `hook(it, X, C, accepted, E)` (src/RRRMC.jl:61-64), called
every `step` iterations with the vectors of all replicas.  Integer models: the library is re-entered per segment (`seed = 0` keeps
the streams going; the recomputed energy equals the tracked one exactly).
"""
function standardMC_hooked(ctx::Ctx, X, β::Real, iters::Integer; seed = DEFAULT_SEED, step::Integer = 1, hook = (x...) -> true,
                           C0::Union{Vector{RRRMC.Config},Nothing} = nothing)
    N = RRRMC.getN(X)
    seed > 0 && check(ccall((:rrrmc_seed, LIB), Int32, (Ptr{Cvoid}, UInt64), ctx.p, seed), ctx.p)
    chunks = set_configs!(ctx, N, C0)
    Es = Vector{Vector{Int}}()
    accepted = zeros(Int, ctx.R); acc = similar(accepted); E = similar(accepted)
    it = 0
    run!(n) = n > 0 && (check(ccall((:rrrmc_standard_mc, LIB), Int32, (Ptr{Cvoid}, Float64, Int64, Int64, Ptr{Int64}, Ptr{Int64}),
                                    ctx.p, β, n, n + 1, C_NULL, acc), ctx.p); accepted .+= acc)
    Cs = nothing
    while it < iters
        nxt = (it ÷ step + 1) * step
        run!(min(nxt - 1, iters) - it); it = min(nxt - 1, iters)
        nxt > iters && break
        check(ccall((:rrrmc_energy, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), ctx.p, E), ctx.p)
        push!(Es, copy(E))
        Cs = get_configs!(ctx, N, chunks, C0)
        hook(nxt, X, Cs, copy(accepted), E) || (it = nxt; break)
        run!(1); it = nxt
    end
    return Es, get_configs!(ctx, N, chunks, C0)
end

# ---- GraphQuant over GraphRRG / GraphEA slices (config 5) ------------------------------------------------------------------------
# X.X1[k] are M copies of one slice graph (src/graphs/QT.jl:139-170); spins are slice-major, so C.s.chunks passes as is
function QuantCtx(X::RRRMC.QT.GraphQuant, R::Integer; device = 0, replica0 = 0)
    X1 = X.X1[1]; Nk = RRRMC.getN(X1); M = length(X.X1); K = length(X1.A[1])
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rrrmc_ctx_create_quant, LIB), Int32, (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Int64, Int32, UInt32),
                ref, Nk, K, M, R, device, replica0))
    ctx = Ctx(ref[], R)
    A = Int32.(reinterpret(Int, X1.A) .- 1); J = Int8.(reinterpret(Int, X1.J))
    GC.@preserve A J check(ccall((:rrrmc_set_graph, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int8}), ctx.p, A, J), ctx.p)
    return ctx
end

"""
    rrrMC(ctx, X::GraphQuant{fourK}, β, iters; seed, step, C0, staged_thr, staged_thr_fact)
        -> (Es::Matrix{Float64} samples×R, Cs, accepted, staged iterations)

`rrrMC(X::DoubleGraph, ...)` (src/RRRMC.jl:221-290) for `ctx.R` replicas.
"""
function RRRMC.rrrMC(ctx::Ctx, X::RRRMC.QT.GraphQuant{fourK}, β::Real, iters::Integer; seed = DEFAULT_SEED, step::Integer = 1,
                     C0::Union{Vector{RRRMC.Config},Nothing} = nothing, staged_thr::Real = 0.5, staged_thr_fact::Real = 5.0) where {fourK}
    isfinite(β) || throw(ArgumentError("β must be finite, given: $β"))
    N = RRRMC.getN(X)
    seed > 0 && check(ccall((:rrrmc_seed, LIB), Int32, (Ptr{Cvoid}, UInt64), ctx.p, seed), ctx.p)
    chunks = set_configs!(ctx, N, C0)
    check(ccall((:rrrmc_rrr_mc_async, LIB), Int32, (Ptr{Cvoid}, Float64, Float64, Int64, Int64, Float64, Float64),
                ctx.p, β, fourK, iters, step, staged_thr, staged_thr_fact), ctx.p)           # fourK: the type parameter, QT.jl:126,165
    check(ccall((:rrrmc_sync, LIB), Int32, (Ptr{Cvoid},), ctx.p), ctx.p)
    Es = Matrix{Float64}(undef, iters ÷ step, ctx.R); acc = Vector{Int}(undef, ctx.R); staged = similar(acc)
    check(ccall((:rrrmc_fetch_results_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}), ctx.p, Es, acc), ctx.p)
    check(ccall((:rrrmc_rrr_stats, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), ctx.p, staged), ctx.p)
    return Es, get_configs!(ctx, N, chunks, C0), acc, staged
end

# ---- GraphSKNormal (config 3) ------------------------------------------------------------------------------------------------------
# X.J::Vector{Vector{Float64}} (src/graphs/SK.jl:183) packs to the N×N row-major matrix the library takes
function SKCtx(X::RRRMC.SK.GraphSKNormal, R::Integer; device = 0, replica0 = 0)
    N = RRRMC.getN(X); ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rrrmc_ctx_create, LIB), Int32, (Ref{Ptr{Cvoid}}, Int32, Int64, Int64, Int64, Int32, UInt32), ref, 2, N, 0, R, device, replica0))
    ctx = Ctx(ref[], R)
    Jm = Matrix{Float64}(undef, N, N); for i = 1:N; Jm[:, i] = X.J[i]; end                 # column i of a Julia matrix = row i in C order
    check(ccall((:rrrmc_set_couplings_dense, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), ctx.p, Jm), ctx.p)
    return ctx
end

function RRRMC.standardMC(ctx::Ctx, X::RRRMC.SK.GraphSKNormal, β::Real, iters::Integer; seed = DEFAULT_SEED, step::Integer = 1,
                          C0::Union{Vector{RRRMC.Config},Nothing} = nothing, quiet = false)
    N = RRRMC.getN(X)
    seed > 0 && check(ccall((:rrrmc_seed, LIB), Int32, (Ptr{Cvoid}, UInt64), ctx.p, seed), ctx.p)
    chunks = set_configs!(ctx, N, C0)
    Es = Matrix{Float64}(undef, iters ÷ step, ctx.R); acc = Vector{Int}(undef, ctx.R)
    check(ccall((:rrrmc_standard_mc_f64, LIB), Int32, (Ptr{Cvoid}, Float64, Int64, Int64, Ptr{Float64}, Ptr{Int64}),
                ctx.p, β, iters, step, Es, acc), ctx.p)
    Cs = get_configs!(ctx, N, chunks, C0)
    quiet || println("samples = ", size(Es, 1), "\niters = ", iters, "\naccept rate = ", sum(acc) / (iters * ctx.R))
    return Es, Cs
end

# ---- device-side snapshots for the scripts' hooks (scripts/scripts.jl:51-69: copy(C.s) per sample, pm1dot / parseovs afterwards) ----
snapshot_reserve(ctx::Ctx, n) = check(ccall((:rrrmc_snapshot_reserve, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.p, n), ctx.p)
snapshot_store(ctx::Ctx, slot) = check(ccall((:rrrmc_snapshot_store, LIB), Int32, (Ptr{Cvoid}, Int32), ctx.p, slot), ctx.p)
"q[r, p] = pm1dot(replica r of slot ia[p], replica r of slot ib[p]) (0-based slots, -1 = the live configuration)"
function overlaps(ctx::Ctx, ia::Vector{Int32}, ib::Vector{Int32})
    q = Matrix{Int32}(undef, ctx.R, length(ia))
    check(ccall((:rrrmc_overlaps, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}), ctx.p, length(ia), ia, ib, q), ctx.p)
    return q
end

end # module
