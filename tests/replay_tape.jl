# Replay an RNG-free tape (tests/golden/tape_*.txt) through the reference's OWN graph code and compare with the results the tape
# holds (written by the build's C oracle; the HIP library is checked against the same tapes in tests/test_tapes.py).
#
#   julia --project=<an environment that has RRRMC.jl> tests/replay_tape.jl                     (all fourteen tapes)
#   julia ...                                          tests/replay_tape.jl tests/golden/tape_skn_n24.txt ...
#
# Tapes: standardMC on GraphRRG(128, 3); rrrMC on a GraphQuant (staged and direct branch); round 3: standardMC on GraphEA(2, 3) (doubled
# bonds, undo path of update_cache!), on GraphSKNormal(24) and the binary GraphSK(10) (whole-array swap of SK.jl:247-250 / :106-109), and
# rrrMC on GraphSKNormal(10) through DeltaECacheCont / DynamicSampler (refresh! included); round 4: standardMC on GraphRRGNormal(16, 3)
# (Float64 sparse model; 263 undo swaps of RRG.jl:566-577) and rrrMC on GraphRRGNormal(64, 3) (DeltaECacheCont over sparse neighbourhoods).
#
# With the random draws fixed, standardMC / rrrMC are deterministic functions of the reference's energy, delta_energy, spinflip!,
# DeltaECache and ArraySet code: this script restates only the few lines of the sampler loops that consume random numbers
# (src/RRRMC.jl:39-44,100-119,249-282; src/DeltaE.jl:146-167) and calls the reference for everything else.
# It cannot run in the build container (no Julia there): it is the one-command pin for any machine that has Julia.
# tests/tape_replay.py is the same replay in plain Python.
using RRRMC
using RRRMC: Config, energy, delta_energy, spinflip!, getN, inner_graph, delta_energy_residual
import RRRMC.DeltaE
import RRRMC.DeltaE: gen_ΔEcache, apply_move!, compute_staged!, compute_reverse_probabilities!, apply_staged!, get_z
import RRRMC.DynamicSamplers: getel
import RRRMC.Interface: neighbors

function read_tape(path)
    d = Dict{String,Any}()
    name, want, buf = "", 0, String[]
    for line in eachline(path)
        line = strip(line)
        (isempty(line) || startswith(line, "#")) && continue
        if startswith(line, "@")
            parts = split(line[2:end])
            if length(parts) == 3 && parts[2] == "array"
                name, want, buf = String(parts[1]), parse(Int, parts[3]), String[]
                d[name] = buf
            else
                d[String(parts[1])] = String(parts[2])
                name = ""
            end
            continue
        end
        name == "" || append!(buf, String.(split(line)))
    end
    return d
end

ints(v) = parse.(Int, v)
tuples(v, K) = [ntuple(k -> v[(x - 1) * K + k], K) for x = 1:(length(v) ÷ K)]
function config_from(chunks::Vector{String}, N)
    C = Config(N, init = false)
    C.s.chunks .= parse.(UInt64, chunks, base = 16)
    return C
end
chunks_hex(C) = [string(c, base = 16, pad = 16) for c in C.s.chunks]

# rand_move of a DeltaECache (src/DeltaE.jl:146-167) with its two draws taken from the tape: u1 = the rand() of the class pick, u2 = the
# 64 random bits behind rand(1:t) of the member pick (src/ArraySets.jl:83).  The one restated piece of the reference that both the
# GraphQuant and the GraphRRG replays share.
function rand_move(cache, u1, u2)
    ΔElist, ascache, T, z = cache.ΔElist, cache.ascache, cache.T, cache.z
    L = length(ΔElist)
    r = u1 * z
    k = 0
    cT = 0.0
    for outer k = 1:2L
        cT += T[k]
        r < cT && break
    end
    r < cT || while T[k] == 0
        k -= 1
    end
    ΔE = k ≤ L ? -ΔElist[k] : ΔElist[k - L]
    as = ascache[k]
    return as.v[Int((UInt128(u2) * as.t) >> 64) + 1], ΔE
end

function replay_standardMC(t)
    N, K = parse(Int, t["N"]), parse(Int, t["K"])
    β, iters, step = parse(Float64, t["beta"]), parse(Int, t["iters"]), parse(Int, t["step"])
    A, J = tuples(ints(t["A"]), K), tuples(ints(t["J"]), K)
    X = RRRMC.RRG.GraphRRG{Int,(-1, 1),K}(A, J)                  # the inner constructor, src/graphs/RRG.jl:122
    C = config_from(t["C0"], N)
    sites, us = ints(t["sites"]), parse.(Float64, t["uniforms"])
    Es = Int[]
    E = energy(X, C)                                             # src/RRRMC.jl:95
    accepted = 0
    flips = Int[]
    for it = 1:iters
        it % step == 0 && push!(Es, E)                           # :104-108
        i = sites[it]                                            # rand(1:N), :113
        ΔE = delta_energy(X, C, i)
        x = -β * ΔE
        ok = x ≥ 0 || us[it] < exp(x)                            # accept, :39
        push!(flips, ok)
        ok || continue
        spinflip!(X, C, i)
        E += ΔE
        accepted += 1
    end
    @assert E == energy(X, C)                                    # test/runtests.jl:12-20
    ok = Es == ints(t["expected_Es"]) && chunks_hex(C) == t["expected_chunks"] &&
         accepted == parse(Int, t["expected_accepted"]) && flips == ints(t["expected_flips"])
    println(ok ? "standardMC tape: reference == tape ($(iters) iterations, $(accepted) accepted)" : "standardMC tape: MISMATCH")
    return ok
end

function replay_rrrMC_quant(t)
    Nk, K, M = parse(Int, t["Nk"]), parse(Int, t["K"]), parse(Int, t["M"])
    β, Γ = parse(Float64, t["beta"]), parse(Float64, t["Gamma"])
    iters, step = parse(Int, t["iters"]), parse(Int, t["step"])
    staged_thr, staged_thr_fact = parse(Float64, t["staged_thr"]), parse(Float64, t["staged_thr_fact"])
    A = tuples(ints(t["A"]), K)
    X = if get(t, "slices", "int") == "f64"
        # GraphQEAT = GraphQuant{fourK,GraphEANormal{twoD}} as its constructors build it (src/QAliases.jl:64,72,82): M slices GraphEANormal{K}(L, A, J)
        Jf = parse.(Float64, t["J"])
        Jt = [ntuple(k -> Jf[(x - 1) * K + k], K) for x = 1:Nk]
        RRRMC.QT.GraphQuant(Nk, M, Γ, β, RRRMC.EA.GraphEANormal{K}, parse(Int, t["L"]), A, Jt)
    else
        RRRMC.QT.GraphQuant(Nk, M, Γ, β, RRRMC.RRG.GraphRRG{Int,(-1, 1),K}, A, tuples(ints(t["J"]), K))     # src/graphs/QT.jl:161-168: M slice graphs over (A, J)
    end
    N = getN(X)
    C = config_from(t["C0"], N)
    ucls, uacc = parse.(Float64, t["u_class"]), parse.(Float64, t["u_accept"])
    umem = parse.(UInt64, t["u_member"])
    accept(c, x, u) = (c ≥ 1 && x ≥ 0) || (a = c * exp(x); a ≥ 1 || u < a)           # src/RRRMC.jl:40-44
    Es = Float64[]
    E = energy(X, C)
    X0 = inner_graph(X)
    cache = gen_ΔEcache(X0, C, β)
    λ = staged_thr_fact / N
    staged_its, accepted, acc_rate = 0, 0, 0.5
    for it = 1:iters                                                                   # src/RRRMC.jl:249-282
        it % step == 0 && push!(Es, E)
        acc = false
        if acc_rate < staged_thr
            staged_its += 1
            z = get_z(cache)
            move, ΔE0 = rand_move(cache, ucls[it], umem[it])
            compute_staged!(X0, C, move, cache)
            z′ = compute_reverse_probabilities!(cache)
            c = z / z′
            ΔE1 = delta_energy_residual(X, C, move)
            if accept(c, -β * ΔE1, uacc[it])
                spinflip!(X, C, move)
                apply_staged!(cache)
                E += ΔE0 + ΔE1
                accepted += 1
                acc = true
            end
        else
            move, ΔE0 = rand_move(cache, ucls[it], umem[it])
            ΔE1 = delta_energy_residual(X, C, move)
            c = apply_move!(X, C, move, cache)
            if accept(c, -β * ΔE1, uacc[it])
                E += ΔE0 + ΔE1
                accepted += 1
                acc = true
            else
                apply_move!(X, C, move, cache)
            end
        end
        acc_rate = acc_rate * (1 - λ) + acc * λ
    end
    DeltaE.check_consistency(cache)
    ok = isapprox(Es, parse.(Float64, t["expected_Es"]), rtol = 1e-12, atol = 1e-12) && chunks_hex(C) == t["expected_chunks"] &&
         accepted == parse(Int, t["expected_accepted"]) && staged_its == parse(Int, t["expected_staged_its"]) &&
         [length(a) for a in cache.ascache] == ints(t["expected_sizes"]) && cache.pos == ints(t["expected_pos"])
    println(ok ? "rrrMC(GraphQuant) tape: reference == tape ($(iters) iterations, $(accepted) accepted, $(staged_its) staged)" :
                 "rrrMC(GraphQuant) tape: MISMATCH")
    return ok
end

# standardMC (src/RRRMC.jl:81-127) over any graph object of the reference with the draws of the tape; returns (Es, accepted, flips)
function run_standardMC(X, C, β, iters, step, sites, us)
    ET = typeof(energy(X, C))
    Es = ET[]
    E = energy(X, C)                                             # :95
    accepted, flips = 0, Int[]
    for it = 1:iters
        it % step == 0 && push!(Es, E)                           # :104-108
        i = sites[it]                                            # rand(1:N), :113
        ΔE = delta_energy(X, C, i)
        x = -β * ΔE
        ok = x ≥ 0 || us[it] < exp(x)                            # accept, :39
        push!(flips, ok)
        ok || continue
        spinflip!(X, C, i)
        E += ΔE
        accepted += 1
    end
    @assert abs(E - energy(X, C)) < 1e-9                         # test/runtests.jl:12-20
    return Es, accepted, flips
end

function replay_standardMC_ea(t)
    N, K = parse(Int, t["N"]), parse(Int, t["K"])
    A, J = tuples(ints(t["A"]), K), tuples(ints(t["J"]), K)
    X = RRRMC.EA.GraphEA{Int,(-1, 1),K}(A, J)                    # the inner constructor, src/graphs/EA.jl:145 (L = 2: uA = every second entry)
    C = config_from(t["C0"], N)
    Es, accepted, flips = run_standardMC(X, C, parse(Float64, t["beta"]), parse(Int, t["iters"]), parse(Int, t["step"]),
                                         ints(t["sites"]), parse.(Float64, t["uniforms"]))
    ok = Es == ints(t["expected_Es"]) && chunks_hex(C) == t["expected_chunks"] && accepted == parse(Int, t["expected_accepted"]) &&
         flips == ints(t["expected_flips"])
    println(ok ? "standardMC(GraphEA L=2) tape: reference == tape ($(accepted) accepted)" : "standardMC(GraphEA L=2) tape: MISMATCH")
    return ok
end

function sk_graph(t)
    N = parse(Int, t["N"])
    if endswith(t["kind"], "skb")                                # GraphSK(J::Vector{BitVector}), src/graphs/SK.jl:34-48
        nch = (N + 63) >> 6
        rows = parse.(UInt64, t["J_chunks"], base = 16)
        J = BitVector[(b = BitVector(undef, N); b.chunks .= rows[((i - 1) * nch + 1):(i * nch)]; b) for i = 1:N]
        return RRRMC.SK.GraphSK(J)
    end
    Jf = parse.(Float64, t["J"])
    return RRRMC.SK.GraphSKNormal([Jf[((i - 1) * N + 1):(i * N)] for i = 1:N])       # GraphSKNormal(J; check = true), SK.jl:184-195
end

function replay_standardMC_sk(t)
    N = parse(Int, t["N"])
    X = sk_graph(t)
    C = config_from(t["C0"], N)
    Es, accepted, _ = run_standardMC(X, C, parse(Float64, t["beta"]), parse(Int, t["iters"]), parse(Int, t["step"]),
                                     ints(t["sites"]), parse.(Float64, t["uniforms"]))
    lf = Float64.(X.cache.lfields)
    # the same IEEE operations in the same order: the reference reproduces the tape to the last bit wherever its @simd loops are
    # element-wise (they are); isapprox with a 1e-12 guard keeps the verdict robust to a different summation order in `energy`
    ok = isapprox(Es, parse.(Float64, t["expected_Es"]), rtol = 1e-12, atol = 1e-12) && chunks_hex(C) == t["expected_chunks"] &&
         accepted == parse(Int, t["expected_accepted"]) && isapprox(lf, parse.(Float64, t["expected_lfields"]), rtol = 1e-12, atol = 1e-12)
    println(ok ? "standardMC($(typeof(X))) tape: reference == tape ($(accepted) accepted, bit-equal energies: $(Es == parse.(Float64, t["expected_Es"])))" :
                 "standardMC($(typeof(X))) tape: MISMATCH")
    return ok
end

# standardMC on GraphRRGNormal (src/graphs/RRG.jl:503-609).  The struct has no constructor from (A, J): one is drawn at random by the
# reference (GraphRRGNormal{K}(N)) and its A and J vectors are overwritten in place with the tape's; energy() rebuilds the cache from them.
function rrgn_graph(t)
    N, K = parse(Int, t["N"]), parse(Int, t["K"])
    X = RRRMC.RRG.GraphRRGNormal(N, K)
    Af, Jf = ints(t["A"]), parse.(Float64, t["J"])
    for x = 1:N
        X.A[x] = ntuple(k -> Af[(x - 1) * K + k], K)
        X.J[x] = ntuple(k -> Jf[(x - 1) * K + k], K)
    end
    return X
end

function replay_standardMC_rrgn(t)
    N = parse(Int, t["N"])
    X = rrgn_graph(t)
    C = config_from(t["C0"], N)
    Es, accepted, _ = run_standardMC(X, C, parse(Float64, t["beta"]), parse(Int, t["iters"]), parse(Int, t["step"]),
                                     ints(t["sites"]), parse.(Float64, t["uniforms"]))
    lf = Float64.(X.cache.lfields)
    ok = isapprox(Es, parse.(Float64, t["expected_Es"]), rtol = 1e-12, atol = 1e-12) && chunks_hex(C) == t["expected_chunks"] &&
         accepted == parse(Int, t["expected_accepted"]) && isapprox(lf, parse.(Float64, t["expected_lfields"]), rtol = 1e-12, atol = 1e-12)
    println(ok ? "standardMC(GraphRRGNormal) tape: reference == tape ($(accepted) accepted, bit-equal energies: $(Es == parse.(Float64, t["expected_Es"])))" :
                 "standardMC(GraphRRGNormal) tape: MISMATCH")
    return ok
end

# rrrMC(X::SingleGraph) (src/RRRMC.jl:149-219) on GraphSKNormal with DeltaECacheCont: rand(dynsmp) = getel(dynsmp, u) (DynamicSamplers.jl:154)
function replay_rrrMC_skn(t)
    N = parse(Int, t["N"])
    β, iters, step = parse(Float64, t["beta"]), parse(Int, t["iters"]), parse(Int, t["step"])
    staged_thr, staged_thr_fact = parse(Float64, t["staged_thr"]), parse(Float64, t["staged_thr_fact"])
    X = t["kind"] == "rrrMC_rrgn" ? rrgn_graph(t) : sk_graph(t)          # the same sampler over GraphRRGNormal (round 4)
    C = config_from(t["C0"], N)
    umove, uacc = parse.(Float64, t["u_move"]), parse.(Float64, t["u_accept"])
    Es = Float64[]
    E = energy(X, C)
    cache = gen_ΔEcache(X, C, β)                                 # DeltaECacheCont, src/DeltaE.jl:299-315
    rand_move(u) = (move = getel(cache.dynsmp, u); (move, cache.ΔEs[move]))          # src/DeltaE.jl:327-333
    λ = staged_thr_fact / N
    staged_its, accepted, acc_rate = 0, 0, 0.5
    for it = 1:iters
        it % step == 0 && push!(Es, E)
        acc = false
        if acc_rate < staged_thr
            staged_its += 1
            z = get_z(cache)                                     # step_rrr, src/RRRMC.jl:131-138
            move, ΔE = rand_move(umove[it])
            compute_staged!(X, C, move, cache)
            z′ = compute_reverse_probabilities!(cache)
            c = z / z′
            if uacc[it] < c                                      # :192
                spinflip!(X, C, move)
                apply_staged!(cache)
                E += ΔE
                accepted += 1
                acc = true
            end
        else
            move, ΔE = rand_move(umove[it])
            c = apply_move!(X, C, move, cache)
            if uacc[it] < c                                      # :202
                E += ΔE
                accepted += 1
                acc = true
            else
                apply_move!(X, C, move, cache)
            end
        end
        acc_rate = acc_rate * (1 - λ) + acc * λ
    end
    ok = isapprox(Es, parse.(Float64, t["expected_Es"]), rtol = 1e-10, atol = 1e-10) && chunks_hex(C) == t["expected_chunks"] &&
         accepted == parse(Int, t["expected_accepted"]) && staged_its == parse(Int, t["expected_staged_its"]) &&
         (!haskey(t, "expected_dEs") || (isapprox(cache.ΔEs, parse.(Float64, t["expected_dEs"]), rtol = 1e-10, atol = 1e-10) &&
                                        isapprox(cache.dynsmp.z, parse(Float64, t["expected_z"]), rtol = 1e-10)))
    println(ok ? "rrrMC($(typeof(X))) tape: reference == tape ($(accepted) accepted, $(staged_its) staged)" : "rrrMC($(typeof(X))) tape: MISMATCH")
    return ok
end

# rrrMC(X::SingleGraph) (src/RRRMC.jl:149-219) and bklMC (:311-359) on GraphRRG{Int,(-1,1),K} with the reference's own DeltaECache{Int,L}:
# gen_ΔEcache, compute_staged!, compute_reverse_probabilities!, apply_staged!, apply_move! are the reference's; rand_move, rand_skip and the
# `rand() < c` take their rand() from the tape.
function replay_rrr_bkl_rrg(t)
    bkl = t["kind"] == "bklMC_rrg"
    N, K = parse(Int, t["N"]), parse(Int, t["K"])
    β, iters, step = parse(Float64, t["beta"]), parse(Int, t["iters"]), parse(Int, t["step"])
    A, J = tuples(ints(t["A"]), K), tuples(ints(t["J"]), K)
    X = RRRMC.RRG.GraphRRG{Int,(-1, 1),K}(A, J)
    C = config_from(t["C0"], N)
    ucls, umem = parse.(Float64, t["u_class"]), parse.(UInt64, t["u_member"])
    Es = Int[]
    E = energy(X, C)
    accepted, staged_its, it = 0, 0, 0
    if !bkl
        uacc = parse.(Float64, t["u_accept"])
        staged_thr, λ = parse(Float64, t["staged_thr"]), parse(Float64, t["staged_thr_fact"]) / N
        cache = gen_ΔEcache(X, C, β)
        acc_rate = 0.5
        for it = 1:iters                                                               # src/RRRMC.jl:178-210
            it % step == 0 && push!(Es, E)
            acc = false
            if acc_rate < staged_thr
                staged_its += 1
                z = get_z(cache)                                                       # step_rrr, src/RRRMC.jl:131-138
                move, ΔE = rand_move(cache, ucls[it], umem[it])
                compute_staged!(X, C, move, cache)
                c = z / compute_reverse_probabilities!(cache)
                if uacc[it] < c
                    spinflip!(X, C, move)
                    apply_staged!(cache)
                    E += ΔE; accepted += 1; acc = true
                end
            else
                move, ΔE = rand_move(cache, ucls[it], umem[it])
                c = apply_move!(X, C, move, cache)
                if uacc[it] < c
                    E += ΔE; accepted += 1; acc = true
                else
                    apply_move!(X, C, move, cache)
                end
            end
            acc_rate = acc_rate * (1 - λ) + acc * λ
        end
        it = iters
    else
        uskip = parse.(Float64, t["u_skip"])
        cache = gen_ΔEcache(X, C, β, false)
        nextstep, m = step, 0
        while it < iters                                                               # src/RRRMC.jl:331-349
            m += 1
            skip = floor(Int, Base.log1p(-uskip[m]) / Base.log1p(-cache.z / N))        # rand_skip, src/DeltaE.jl:141-144
            move, ΔE = rand_move(cache, ucls[m], umem[m])
            out = false
            while it + skip + 1 ≥ nextstep
                push!(Es, E)
                nextstep += step
                nextstep > iters && (out = true; break)
            end
            out && break
            apply_move!(X, C, move, cache)                                             # apply_step_bkl!, src/RRRMC.jl:297-298
            it += skip + 1
            E += ΔE
            accepted += 1
        end
        staged_its = accepted
    end
    DeltaE.check_consistency(cache)
    ok = Es == ints(t["expected_Es"]) && chunks_hex(C) == t["expected_chunks"] && accepted == parse(Int, t["expected_accepted"]) &&
         staged_its == parse(Int, t["expected_staged_its"]) && it == parse(Int, t["expected_iters_done"]) && E == energy(X, C) &&
         [length(a) for a in cache.ascache] == ints(t["expected_sizes"]) && cache.pos == ints(t["expected_pos"])
    println(ok ? "$(t["kind"]) tape: reference == tape ($(it) iterations, $(accepted) moves accepted, $(staged_its) staged)" : "$(t["kind"]) tape: MISMATCH")
    return ok
end

# wtmMC (src/RRRMC.jl:376-426) on GraphRRG{Int,(-1,1),K}: the reference's graph functions and the heap it uses (DataStructures'
# MutableBinaryMinHeap, src/WaitingTimes.jl:24); THeap(X, C, β) and update_heap! (:26-52) are restated with gen_wt's rand() from the tape.
function replay_wtmMC_rrg(t)
    N, K = parse(Int, t["N"]), parse(Int, t["K"])
    β, samples, step = parse(Float64, t["beta"]), parse(Int, t["samples"]), parse(Float64, t["step"])
    A, J = tuples(ints(t["A"]), K), tuples(ints(t["J"]), K)
    X = RRRMC.RRG.GraphRRG{Int,(-1, 1),K}(A, J)
    C = config_from(t["C0"], N)
    us = parse.(Float64, t["uniforms"])
    nd = 0
    gen_wt(τ) = (nd += 1; -τ * log1p(-us[nd]))                                          # src/WaitingTimes.jl:18-22
    τΔE(ΔE) = max(1.0, Float64(exp(β * ΔE)))                                           # :16
    theap = RRRMC.WaitingTimes.THeap()
    for i = 1:N
        j = push!(theap, gen_wt(τΔE(delta_energy(X, C, i))))
        @assert j == i
    end
    Es = Int[]
    E = energy(X, C)
    step /= N
    tmax = step * samples
    tg, nextstep, num_moves, out = 0.0, step, 0, false
    while tg < tmax && !out                                                            # src/RRRMC.jl:400-417
        t′, move = RRRMC.WaitingTimes.pick_next(theap)
        while t′ ≥ nextstep
            push!(Es, E)
            nextstep += step
            nextstep > tmax + 1e-10 && (out = true; break)
        end
        out && break
        tg = t′
        ΔE = delta_energy(X, C, move)                                                  # update_heap!, src/WaitingTimes.jl:40-52
        spinflip!(X, C, move)
        RRRMC.WaitingTimes.update!(theap, move, tg + gen_wt(τΔE(-ΔE)))
        for j in neighbors(X, move)
            RRRMC.WaitingTimes.update!(theap, j, tg + gen_wt(τΔE(delta_energy(X, C, j))))
        end
        E += ΔE
        num_moves += 1
    end
    ok = Es == ints(t["expected_Es"]) && chunks_hex(C) == t["expected_chunks"] && num_moves == parse(Int, t["expected_num_moves"]) &&
         isapprox(tg, parse(Float64, t["expected_t"]), rtol = 1e-12) && E == energy(X, C)
    println(ok ? "wtmMC(GraphRRG) tape: reference == tape ($(num_moves) moves, global time $(tg))" : "wtmMC(GraphRRG) tape: MISMATCH")
    return ok
end

# extremal_opt (src/RRRMC.jl:474-521) on GraphRRG{Int,(-1,1),K}: the reference's EOCache (gen_EOcache, apply_move!, src/DeltaE.jl:422-552);
# rand_move (:484-516) is restated with its two draws from the tape.
function replay_extremal_opt_rrg(t)
    N, K = parse(Int, t["N"]), parse(Int, t["K"])
    τ, iters, step = parse(Float64, t["tau"]), parse(Int, t["iters"]), parse(Int, t["step"])
    A, J = tuples(ints(t["A"]), K), tuples(ints(t["J"]), K)
    X = RRRMC.RRG.GraphRRG{Int,(-1, 1),K}(A, J)
    C = config_from(t["C0"], N)
    urank, umem = parse.(Float64, t["u_rank"]), parse.(UInt64, t["u_member"])
    cache = DeltaE.gen_EOcache(X, C, τ)
    function rand_move(cache, u1, u2)
        ΔElist, has_zero, ascache, fτ, z = cache.ΔElist, cache.has_zero, cache.ascache, cache.fτ, cache.z
        L = length(ΔElist)
        r = (1 - u1) * z
        i = searchsortedfirst(fτ, r)
        k = 0
        tt = 0
        while i > tt
            k += 1
            tt += length(ascache[k])
        end
        ΔE = k ≤ L ? -ΔElist[L + 1 - k] : ΔElist[k - L + has_zero]
        as = ascache[k]
        return as.v[Int((UInt128(u2) * as.t) >> 64) + 1], ΔE
    end
    Es = Int[]
    E = energy(X, C)
    Emin, Cmin, itmin = E, copy(C), 0
    for it = 1:iters                                                                   # src/RRRMC.jl:494-513
        it % step == 0 && push!(Es, E)
        move, ΔE = rand_move(cache, urank[it], umem[it])
        apply_move!(X, C, move, cache)
        E += ΔE
        if E < Emin
            Emin = E
            copy!(Cmin, C)
            itmin = it
        end
    end
    DeltaE.check_consistency(cache)
    ok = Es == ints(t["expected_Es"]) && chunks_hex(C) == t["expected_chunks"] && Emin == parse(Int, t["expected_Emin"]) &&
         itmin == parse(Int, t["expected_itmin"]) && chunks_hex(Cmin) == t["expected_Cmin"] && E == energy(X, C)
    println(ok ? "extremal_opt(GraphRRG) tape: reference == tape (Emin $(Emin) at iteration $(itmin))" : "extremal_opt(GraphRRG) tape: MISMATCH")
    return ok
end

function main(paths)
    allok = true
    for p in paths
        t = read_tape(p)
        k = t["kind"]
        allok &= k == "standardMC" ? (get(t, "form", "rrg") == "ea" ? replay_standardMC_ea(t) : replay_standardMC(t)) :
                 k == "standardMC_rrgn" ? replay_standardMC_rrgn(t) :
                 k == "rrrMC_quant" ? replay_rrrMC_quant(t) :
                 (k == "rrrMC_skn" || k == "rrrMC_rrgn") ? replay_rrrMC_skn(t) :
                 (k == "rrrMC_rrg" || k == "bklMC_rrg") ? replay_rrr_bkl_rrg(t) :
                 k == "wtmMC_rrg" ? replay_wtmMC_rrg(t) :
                 k == "extremal_opt_rrg" ? replay_extremal_opt_rrg(t) : replay_standardMC_sk(t)
    end
    exit(allok ? 0 : 1)
end

main(isempty(ARGS) ? [joinpath(@__DIR__, "golden", f) for f in
                      ("tape_rrg_n128.txt", "tape_quant_nk16_m4.txt", "tape_quant_direct.txt", "tape_ea_l2_d3.txt", "tape_skn_n24.txt",
                       "tape_sk_n10.txt", "tape_rrr_skn_n10.txt", "tape_rrr_rrg_n64.txt", "tape_bkl_rrg_n64.txt", "tape_wtm_rrg_n64.txt", "tape_eo_rrg_n64.txt", "tape_rrgn_n16.txt", "tape_rrr_rrgn_n64.txt",
                       "tape_quant_qeat_l4_m8.txt")] : ARGS)
