// gfx950 kernel for standardMC (src/RRRMC.jl:81-127) on the dense Gaussian SK model GraphSKNormal (src/graphs/SK.jl:170-297), blocked
// form (round 3; sk_sweep_kernel of sk_kernels.hpp is the one-attempt-per-two-barriers form it replaces, kept as RRRMC_SK_LEGACY=1).
//
// The chain is sequential, but most of it is not on its critical path.  The sites and the acceptance uniforms do not depend on the
// state (SITE / ACCEPT_F64 streams), a rejected attempt changes nothing, and the decision of attempt t only reads ONE local field,
// lfields[site_t].  So the kernel takes W = 64 attempts at a time (a *block*):
//
//   gather   the owners of the block's 64 sites publish those sites' lfields / lfields_last / spin into a *window* (LDS);
//   decide   one wavefront per replica, lane m = attempt m of the block.  Every lane keeps its verdict u_m < exp(-beta f_m) current,
//            so the next ACCEPTED attempt is the first set bit of a ballot: rejected attempts cost nothing.  An accepted attempt k
//            updates the lanes m > k exactly as update_cache! (SK.jl:239-276) updates those sites' entries — f_m += 4 sigma J[k][m]
//            (one row of the block's 64 x 64 coupling sub-matrix, LDS), lfields[move] = -lfm for a repeated site, the array swap of
//            SK.jl:247-250 when the previous accepted move was the same site — and re-evaluates their verdicts (ln u against x, the
//            exponential only inside a 1e-9 band: see the decide phase);
//   apply    all threads apply the block's accepted moves in order to the full field arrays, which live in registers (thread t owns
//            sites t, t + NTH, ... of the workgroup's 8 or 4 replicas, as in sk_sweep_kernel); the rows of 4J are loaded a group of
//            PF attempts ahead.
//
// Three workgroup barriers per 64 attempts instead of two per attempt, and the chain's length is counted in accepted moves.  Every
// field — tracked in the window or in the registers — receives the reference's sequence of IEEE operations, so trajectories, energies
// and the cache are bit-identical to the oracle (tests/emulate_sk_block.py is the CPU model of this schedule, checked against it).
// The block's sites and its coupling sub-matrix 4 J[site_k][site_m] are the same for every workgroup: sk_block_prep_kernel writes them
// once per segment of <= kSkSegIters iterations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sk_kernels.hpp"

namespace rrrmc {

constexpr int kSkW = 64;                      // attempts per block (= lanes of the deciding wavefront)
constexpr int64_t kSkSegIters = 1 << 16;      // iterations per launch: 1024 blocks, 32 MiB of coupling sub-matrices

struct SkBlockParams {
    const double* J4;          // [N][ldJ]  4 * J (exact); rows zero-padded to ldJ = N rounded up to 1024, so that row loads need no bounds test
    const double* blkJw;       // [nblk][64][64]  4 J[site_k][site_m] of every block of the segment
    const uint32_t* blkSites;  // [nblk + 1][64]  sites (0 past the end of the segment)
    double* lf;                // [G][N][kSkRB]
    double* lfl;
    double* hl;                // sk_hblock_kernel: [G * kSkRB][N] scratch, lfields_last in the kernel's sign-free form during a launch
    int32_t* move_last;        // [G][kSkRB]
    uint8_t* spins;            // [G][N]
    double* E_cur;             // [G * kSkRB]
    int64_t* acc_cur;
    double* Es;                // [nsamples][G * kSkRB]; may be null
    double beta;
    double sN;                 // BIN: sqrt(N) (delta_energy = lfields / sqrt(N), SK.jl:137-140); unused otherwise
    uint64_t g0;               // stream position at the start of the segment
    int64_t iters, step, it_base;   // this segment's iterations; it_base = iterations of the call before the segment
    uint32_t k0, k1, replica0;
    int N, ldJ;
};

// sites and coupling sub-matrix of every block of a segment (state independent, shared by all workgroups)
__global__ __launch_bounds__(256) void sk_block_prep_kernel(const double* __restrict__ J4, uint32_t* __restrict__ blkSites, double* __restrict__ blkJw,
                                                            uint64_t g0, int64_t iters, int64_t nblk, uint32_t k0, uint32_t k1, int N, int ldJ)
{
    __shared__ uint32_t s[kSkW];
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid < kSkW) {
        const int64_t it = b * kSkW + tid + 1;
        const uint32_t v = (b < nblk && it <= iters) ? site_of(k0, k1, g0 + (uint64_t)it, (uint32_t)N) : 0u;
        s[tid] = v;
        blkSites[b * kSkW + tid] = v;
    }
    __syncthreads();
    if (b >= nblk) return;
    for (int e = tid; e < kSkW * kSkW; e += 256) {
        const int k = e >> 6, m = e & 63;
        blkJw[b * (kSkW * kSkW) + e] = J4[(size_t)s[k] * ldJ + s[m]];
    }
}

// det_exp (sk_kernels.hpp) without branches: the same operations in the same order for every x it does not short-cut
__device__ __forceinline__ double det_exp_v(double x, const double (&c)[17])
{
    const double k = floor(__dadd_rn(__dmul_rn(x, c[14]), 0.5));
    const double r = __dadd_rn(__dadd_rn(x, -__dmul_rn(k, c[15])), -__dmul_rn(k, c[16]));
    double p = c[13];
#pragma unroll
    for (int n = 12; n >= 0; --n) p = __dadd_rn(__dmul_rn(p, r), c[n]);
    double e = ldexp(p, (int)k);
    e = x < -745.2 ? 0.0 : e;
    e = x > 709.7 ? __builtin_inf() : e;
    return x != x ? x : e;
}

__device__ __forceinline__ double sk_readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// One accepted attempt applied to the fields of ONE replica held by this wavefront (update_cache!, SK.jl:239-276), as one asm statement
// with its control flow INSIDE: at the C++ level the statement is unconditional straight-line code over tied operands, so every field
// stays in its register (written as C++ under wave-uniform branches, the compiler's merges copied every field of the replica around
// each branch — two thirds of the loop).  w = the attempt's word from the deciding wavefronts (bit r accepted, 8 + r undo swap, 16 + r the
// moved spin before its flip); own = 0 in the wavefronts that do not hold the moved site, else 1 + its q; lb = the moved site's lane bit.
//   not accepted:  two scalar instructions.
//   accepted:      lfields_last[j] = lfj (all q first); lfields[j] = lfj + 4 sigma J (SK.jl:256-262) as one fused multiply-add per site word,
//                  D * sigma + lfj with sigma = +-1.0 (exact product: the add's result) — sigma = -1 in the lanes whose spin equals the moved
//                  spin before its flip; then lfields[move] = -lfm in the owner's lane (SK.jl:263-264; x * -1.0 is exact) and the spin flip
//                  in the lane mask (spinflip!, Interface.jl:89-92).
//   undo swap:     lfields <-> lfields_last (SK.jl:247-250) and the spin flips back.
// All lanes are active at every call site (uniform control flow of a full workgroup): EXEC returns to all ones.  T0 / T1 / TP name the
// statements' scratch pair (declared as clobbered).
#define SK_CPY_Q(LF, LFL) "v_mov_b64 " LFL ", " LF "\n\t"
// lfields[j] = lfj + 4 sigma J as ONE fused multiply-add with sigma = +-1.0 built from the spin mask (x * +-1.0 is exact, so the result is
// the add's, bit for bit): v[254:255] = {0, sm ? 0xbff00000 : 0x3ff00000}; the sign of the moved spin picks the code path (D or -D as
// the multiplicand), so EXEC is never written here — the compute unit's one scalar unit is what two co-resident workgroups compete
// for, and the two half-masked adds of the first version cost two scalar instructions per site word and replica
#define SK_FMA_Q(LF, LFL, SM, D, T1, TP) "v_cndmask_b32 " T1 ", %[c1], %[cm1], " SM "\n\tv_fma_f64 " LF ", " D ", " TP ", " LFL "\n\t"
#define SK_OWN_Q(QP1, LF, LFL, SM) \
    "s_cmp_eq_u32 %[own], " #QP1 "\n\ts_cbranch_scc0 1f\n\tv_mul_f64 " LF ", " LFL ", -1.0\n\ts_xor_b64 " SM ", " SM ", %[lb]\n1:\n\t"
#define SK_SWP_Q(LF, LFL, TP) "v_mov_b64 " TP ", " LF "\n\tv_mov_b64 " LF ", " LFL "\n\tv_mov_b64 " LFL ", " TP "\n\t"
#define SK_SWO_Q(QP1, SM) "s_cmp_eq_u32 %[own], " #QP1 "\n\ts_cbranch_scc0 1f\n\ts_xor_b64 " SM ", " SM ", %[lb]\n1:\n\t"
// not accepted -> 9; undo swap -> 5; then the copies (shared), and the moved spin's sign picks the multiplicand: spin bit 1 -> 4 (sigma = -1 where
// the lane's spin equals it: +D), else fall through (-D)
#define SK_HEAD(R, T0) \
    "s_bitcmp1_b32 %[w], " #R "\n\ts_cbranch_scc0 9f\n\ts_bitcmp1_b32 %[w], " #R "+8\n\ts_cbranch_scc1 5f\n\tv_mov_b32 " T0 ", 0\n\t"
#define SK_SIGN(R) "s_bitcmp1_b32 %[w], " #R "+16\n\ts_cbranch_scc1 4f\n\t"
#define SK_MID "6:\n\ts_cmp_eq_u32 %[own], 0\n\ts_cbranch_scc1 9f\n\ts_mov_b64 exec, %[lb]\n\t"
#define SK_TAIL "s_mov_b64 exec, -1\n\ts_branch 9f\n5:\n\t"
#define SK_OPS_COMMON [c1] "v"(c_one), [cm1] "v"(c_mone), [w] "s"(w), [own] "s"(own), [lb] "s"(lb)
#define SK_APPLY1(R, lf, lfl, sm, d, T0, T1, TP) \
    asm volatile(SK_HEAD(R, T0) SK_CPY_Q("%[a0]", "%[b0]") SK_SIGN(R) SK_FMA_Q("%[a0]", "%[b0]", "%[m0]", "-%[d0]", T1, TP) "s_branch 6f\n4:\n\t" \
                 SK_FMA_Q("%[a0]", "%[b0]", "%[m0]", "%[d0]", T1, TP) SK_MID SK_OWN_Q(1, "%[a0]", "%[b0]", "%[m0]") SK_TAIL \
                 SK_SWP_Q("%[a0]", "%[b0]", TP) SK_SWO_Q(1, "%[m0]") "9:" \
                 : [a0] "+v"(lf[0][R]), [b0] "+v"(lfl[0][R]), [m0] "+s"(sm[0][R]) \
                 : [d0] "v"(d[0]), SK_OPS_COMMON : "scc", T0, T1)
#define SK_APPLY2(R, lf, lfl, sm, d, T0, T1, TP) \
    asm volatile(SK_HEAD(R, T0) SK_CPY_Q("%[a0]", "%[b0]") SK_CPY_Q("%[a1]", "%[b1]") SK_SIGN(R) \
                 SK_FMA_Q("%[a0]", "%[b0]", "%[m0]", "-%[d0]", T1, TP) SK_FMA_Q("%[a1]", "%[b1]", "%[m1]", "-%[d1]", T1, TP) "s_branch 6f\n4:\n\t" \
                 SK_FMA_Q("%[a0]", "%[b0]", "%[m0]", "%[d0]", T1, TP) SK_FMA_Q("%[a1]", "%[b1]", "%[m1]", "%[d1]", T1, TP) SK_MID \
                 SK_OWN_Q(1, "%[a0]", "%[b0]", "%[m0]") SK_OWN_Q(2, "%[a1]", "%[b1]", "%[m1]") SK_TAIL \
                 SK_SWP_Q("%[a0]", "%[b0]", TP) SK_SWP_Q("%[a1]", "%[b1]", TP) SK_SWO_Q(1, "%[m0]") SK_SWO_Q(2, "%[m1]") "9:" \
                 : [a0] "+v"(lf[0][R]), [b0] "+v"(lfl[0][R]), [m0] "+s"(sm[0][R]), [a1] "+v"(lf[1][R]), [b1] "+v"(lfl[1][R]), [m1] "+s"(sm[1][R]) \
                 : [d0] "v"(d[0]), [d1] "v"(d[1]), SK_OPS_COMMON : "scc", T0, T1)
#define SK_APPLY3(R, lf, lfl, sm, d, T0, T1, TP) \
    asm volatile(SK_HEAD(R, T0) SK_CPY_Q("%[a0]", "%[b0]") SK_CPY_Q("%[a1]", "%[b1]") SK_CPY_Q("%[a2]", "%[b2]") SK_SIGN(R) \
                 SK_FMA_Q("%[a0]", "%[b0]", "%[m0]", "-%[d0]", T1, TP) SK_FMA_Q("%[a1]", "%[b1]", "%[m1]", "-%[d1]", T1, TP) SK_FMA_Q("%[a2]", "%[b2]", "%[m2]", "-%[d2]", T1, TP) \
                 "s_branch 6f\n4:\n\t" \
                 SK_FMA_Q("%[a0]", "%[b0]", "%[m0]", "%[d0]", T1, TP) SK_FMA_Q("%[a1]", "%[b1]", "%[m1]", "%[d1]", T1, TP) SK_FMA_Q("%[a2]", "%[b2]", "%[m2]", "%[d2]", T1, TP) SK_MID \
                 SK_OWN_Q(1, "%[a0]", "%[b0]", "%[m0]") SK_OWN_Q(2, "%[a1]", "%[b1]", "%[m1]") SK_OWN_Q(3, "%[a2]", "%[b2]", "%[m2]") SK_TAIL \
                 SK_SWP_Q("%[a0]", "%[b0]", TP) SK_SWP_Q("%[a1]", "%[b1]", TP) SK_SWP_Q("%[a2]", "%[b2]", TP) \
                 SK_SWO_Q(1, "%[m0]") SK_SWO_Q(2, "%[m1]") SK_SWO_Q(3, "%[m2]") "9:" \
                 : [a0] "+v"(lf[0][R]), [b0] "+v"(lfl[0][R]), [m0] "+s"(sm[0][R]), [a1] "+v"(lf[1][R]), [b1] "+v"(lfl[1][R]), [m1] "+s"(sm[1][R]), \
                   [a2] "+v"(lf[2][R]), [b2] "+v"(lfl[2][R]), [m2] "+s"(sm[2][R]) \
                 : [d0] "v"(d[0]), [d1] "v"(d[1]), [d2] "v"(d[2]), SK_OPS_COMMON : "scc", T0, T1)
#define SK_APPLY4(R, lf, lfl, sm, d, T0, T1, TP) \
    asm volatile(SK_HEAD(R, T0) SK_CPY_Q("%[a0]", "%[b0]") SK_CPY_Q("%[a1]", "%[b1]") SK_CPY_Q("%[a2]", "%[b2]") SK_CPY_Q("%[a3]", "%[b3]") SK_SIGN(R) \
                 SK_FMA_Q("%[a0]", "%[b0]", "%[m0]", "-%[d0]", T1, TP) SK_FMA_Q("%[a1]", "%[b1]", "%[m1]", "-%[d1]", T1, TP) \
                 SK_FMA_Q("%[a2]", "%[b2]", "%[m2]", "-%[d2]", T1, TP) SK_FMA_Q("%[a3]", "%[b3]", "%[m3]", "-%[d3]", T1, TP) "s_branch 6f\n4:\n\t" \
                 SK_FMA_Q("%[a0]", "%[b0]", "%[m0]", "%[d0]", T1, TP) SK_FMA_Q("%[a1]", "%[b1]", "%[m1]", "%[d1]", T1, TP) \
                 SK_FMA_Q("%[a2]", "%[b2]", "%[m2]", "%[d2]", T1, TP) SK_FMA_Q("%[a3]", "%[b3]", "%[m3]", "%[d3]", T1, TP) SK_MID \
                 SK_OWN_Q(1, "%[a0]", "%[b0]", "%[m0]") SK_OWN_Q(2, "%[a1]", "%[b1]", "%[m1]") SK_OWN_Q(3, "%[a2]", "%[b2]", "%[m2]") \
                 SK_OWN_Q(4, "%[a3]", "%[b3]", "%[m3]") SK_TAIL \
                 SK_SWP_Q("%[a0]", "%[b0]", TP) SK_SWP_Q("%[a1]", "%[b1]", TP) SK_SWP_Q("%[a2]", "%[b2]", TP) SK_SWP_Q("%[a3]", "%[b3]", TP) \
                 SK_SWO_Q(1, "%[m0]") SK_SWO_Q(2, "%[m1]") SK_SWO_Q(3, "%[m2]") SK_SWO_Q(4, "%[m3]") "9:" \
                 : [a0] "+v"(lf[0][R]), [b0] "+v"(lfl[0][R]), [m0] "+s"(sm[0][R]), [a1] "+v"(lf[1][R]), [b1] "+v"(lfl[1][R]), [m1] "+s"(sm[1][R]), \
                   [a2] "+v"(lf[2][R]), [b2] "+v"(lfl[2][R]), [m2] "+s"(sm[2][R]), [a3] "+v"(lf[3][R]), [b3] "+v"(lfl[3][R]), [m3] "+s"(sm[3][R]) \
                 : [d0] "v"(d[0]), [d1] "v"(d[1]), [d2] "v"(d[2]), [d3] "v"(d[3]), SK_OPS_COMMON : "scc", T0, T1)
#define SK_APPLY_ALL(M, lf, lfl, sm, d, SCR) SK_APPLY_ALL_(M, lf, lfl, sm, d, SCR)        /* (SCR expands to the three names below) */
#define SK_APPLY_ALL_(M, lf, lfl, sm, d, T0, T1, TP) \
    do { M(0, lf, lfl, sm, d, T0, T1, TP); M(1, lf, lfl, sm, d, T0, T1, TP); M(2, lf, lfl, sm, d, T0, T1, TP); M(3, lf, lfl, sm, d, T0, T1, TP); \
         M(4, lf, lfl, sm, d, T0, T1, TP); M(5, lf, lfl, sm, d, T0, T1, TP); M(6, lf, lfl, sm, d, T0, T1, TP); M(7, lf, lfl, sm, d, T0, T1, TP); } while (0)
#define SK_APPLY_HALF(M, lf, lfl, sm, d, SCR) SK_APPLY_HALF_(M, lf, lfl, sm, d, SCR)
#define SK_APPLY_HALF_(M, lf, lfl, sm, d, T0, T1, TP) \
    do { M(0, lf, lfl, sm, d, T0, T1, TP); M(1, lf, lfl, sm, d, T0, T1, TP); M(2, lf, lfl, sm, d, T0, T1, TP); M(3, lf, lfl, sm, d, T0, T1, TP); } while (0)
// the statements' scratch pair: the top of the register file the thread count allows (1024 threads: 128 registers per thread)
#define SK_SCRATCH_HI "v254", "v255", "v[254:255]"
#define SK_SCRATCH_LO "v126", "v127", "v[126:127]"

// RB = replicas per workgroup: 8 (one workgroup = one group of the [G][N][8] state) or 4 (two workgroups per group: replicas 4h .. 4h + 3,
// h = blockIdx.x & 1).  Two 4-replica workgroups of 256 threads per CU instead of one 8-replica workgroup of 512: each tests half as many
// replicas per attempt, and one decides while the other applies — the phases of a workgroup are separated by barriers, two
// independent workgroups fill each other's waits.  (The price: each loads its own rows of 4J.)
// BIN: the binary model GraphSK (src/graphs/SK.jl:28-165) through the same kernel.  Its integer cache lfields[i] = sqrt(N) delta_energy(i) and its
// couplings +-1 are exact in Float64, and update_cache! (SK.jl:98-135: lfj + 8 Jsij - 4, i.e. +-4; lfields[move] = -lfm; the same array swap) is
// the Gaussian model's with 4 J = +-4.0: the host hands over the fields as doubles and the +-4.0 matrix, the only difference is the scale of the
// energy — dE = lfields / sqrt(N), the division the reference and the oracle make.
template <int SPT, int NTH, int RB = kSkRB, bool BIN = false>
__global__ __launch_bounds__(NTH) void sk_block_kernel(SkBlockParams P)
{
    static_assert(RB == 8 || RB == 4, "8 or 4 replicas per workgroup");
    constexpr int NH = kSkRB / RB;                         // workgroups per group of 8 replicas
    constexpr int NWV = NTH / 64;                          // wavefronts
    constexpr int RPW = NWV >= RB ? 1 : RB / NWV;          // replicas decided per wavefront
    constexpr int PF = SPT <= 2 ? 4 : 2;                   // attempts per group of the apply phase (two groups of rows in registers)
    constexpr int LGN = NTH == 256 ? 8 : NTH == 512 ? 9 : 10;
    __shared__ double sh_Jw[kSkW * kSkW];
    __shared__ double sh_wf[kSkW][RB], sh_wfl[kSkW][RB], sh_u[kSkW][RB], sh_L[kSkW][RB];
    __shared__ uint32_t sh_acc[2][kSkW];
    __shared__ uint8_t sh_wsp[kSkW], sh_cslot[kSkW];
    __shared__ uint8_t sh_canon[kSkThreads * kSkMaxSPT];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), N = P.N;
    const int grp = blockIdx.x / NH, r8 = (blockIdx.x % NH) * RB, Rp = (gridDim.x / NH) * kSkRB;      // r8: this workgroup's first replica inside its group

    double lf[SPT][RB], lfl[SPT][RB];
    unsigned long long sm[SPT][RB];         // spins as wave-uniform lane masks, one per (q, replica): bit l = the spin of site q * NTH + 64 wave + l
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int j = q * NTH + tid;
        const uint32_t sb = j < N ? P.spins[(size_t)grp * N + j] : 0u;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            lf[q][r] = j < N ? P.lf[((size_t)grp * N + j) * kSkRB + r8 + r] : 0.0;
            lfl[q][r] = j < N ? P.lfl[((size_t)grp * N + j) * kSkRB + r8 + r] : 0.0;
            sm[q][r] = __ballot((sb >> (r8 + r)) & 1u);
        }
    }
    auto spin_byte = [&](int q) {              // the 8 replicas' spins of this lane's site q
        uint32_t v = 0u;
#pragma unroll
        for (int r = 0; r < RB; ++r) v |= (uint32_t)((sm[q][r] >> lane) & 1ull) << r;
        return v;
    };
    for (int j = tid; j < kSkThreads * kSkMaxSPT; j += NTH) sh_canon[j] = 0xffu;
    __syncthreads();
    // per-replica scalar state of the deciding wavefronts (wave-uniform)
    double E_run[RPW];
    int64_t A_run[RPW];
    int32_t mlast[RPW];
#pragma unroll
    for (int a = 0; a < RPW; ++a) {
        const int r = wv + a * NWV;
        const bool on = r < RB;
        E_run[a] = on ? P.E_cur[grp * kSkRB + r8 + r] : 0.0;
        A_run[a] = on ? P.acc_cur[grp * kSkRB + r8 + r] : 0;
        mlast[a] = on ? P.move_last[grp * kSkRB + r8 + r] : -1;
    }
    int64_t ns[RPW], next_sample[RPW];
#pragma unroll
    for (int a = 0; a < RPW; ++a) { ns[a] = P.it_base / P.step; next_sample[a] = (P.it_base / P.step + 1) * P.step; }
    double expc[17];
    uint32_t c_one = 0x3ff00000u, c_mone = 0xbff00000u;          // high words of +-1.0 (SK_FMA_Q), kept in vector registers across the loop
    asm volatile("" : "+v"(c_one), "+v"(c_mone));
    sk_exp_constants(expc);

    const int64_t nblk = (P.iters + kSkW - 1) / kSkW;
#ifdef RRRMC_SKB_STAMPS
    uint64_t st[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    // block 0: sites, uniforms, coupling sub-matrix, canonical window slot of every attempted site
    auto draw_uniforms = [&](int64_t b) {
        for (int idx = tid; idx < kSkW * RB; idx += NTH) {
            const int l = idx / RB, r = idx % RB;
            const double u = rand53(P.k0, P.k1, P.g0 + (uint64_t)(b * kSkW + l + 1), P.replica0 + (uint32_t)(grp * kSkRB + r8 + r));
            sh_u[l][r] = u;
            sh_L[l][r] = log(u);                           // state independent: the verdict's filter (see accept_verdict)
        }
    };
    uint32_t sv = P.blkSites[lane];                        // sites of the current block, lane = attempt
    if (nblk > 0) {
        for (int e = tid; e < kSkW * kSkW; e += NTH) sh_Jw[e] = P.blkJw[e];
        draw_uniforms(0);
        if (wv == 0) {
            sh_canon[sv] = (uint8_t)lane;                  // any one of the lanes that attempt this site wins
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (the read below must see the winner, not this lane's own store)
            sh_cslot[lane] = sh_canon[sv];
            sh_acc[0][lane] = 0u;
        }
    }
    // rows of 4J, G attempts at a time and one group ahead: while the moves of one group are applied from one register set, the rows of
    // the next group load into the other (all of a group's loads are issued at its top, unconditionally — padded rows — and waited for
    // at the top of the next group: the compiler's vmcnt bookkeeping is exact for that shape, it gave up on a per-step ring)
    double JA[PF][SPT], JB[PF][SPT];
#pragma unroll
    for (int kk = 0; kk < PF; ++kk) {
        const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)sv, kk);
#pragma unroll
        for (int q = 0; q < SPT; ++q) JA[kk][q] = P.J4[(size_t)st * P.ldJ + (q * NTH + tid)];
    }
    __syncthreads();

    for (int64_t b = 0; b < nblk; ++b) {
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tA = __builtin_amdgcn_s_memtime();
#endif
        const int pb = (int)(b & 1);
        const int nv = (int)(P.iters - b * kSkW < kSkW ? P.iters - b * kSkW : kSkW);
        const uint32_t sv_next = P.blkSites[(b + 1) * kSkW + lane];        // (the buffer has one block of zeros past the end)
        // ---- gather: owners publish the window
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int j = q * NTH + tid;
            const uint32_t c = sh_canon[j];
            if (c != 0xffu) {
#pragma unroll
                for (int r = 0; r < RB; ++r) { sh_wf[c][r] = lf[q][r]; sh_wfl[c][r] = lfl[q][r]; }
                sh_wsp[c] = (uint8_t)spin_byte(q);
            }
        }
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tB = __builtin_amdgcn_s_memtime();
#endif
        sk_lds_barrier();
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tC = __builtin_amdgcn_s_memtime();
#endif
        // ---- decide: one wavefront per replica, lane = attempt
#pragma unroll
        for (int a = 0; a < RPW; ++a) {
            const int r = wv + a * NWV;
            if (r < RB) {
                const uint32_t cs = sh_cslot[lane];
                double f = sh_wf[cs][r], fl = sh_wfl[cs][r];
                uint32_t sp = (sh_wsp[cs] >> r) & 1u;
                const double u = sh_u[lane][r];
                const int64_t it0 = P.it_base + b * kSkW;               // iteration of lane m (call-relative, 1-based) = it0 + m + 1
                uint32_t accw = 0u;
                // accept iff x >= 0 || u < det_exp(x), x = -beta dE (RRRMC.jl:39).  det_exp is 40 dependent Float64 operations on the
                // chain's critical path, and u < exp(x) is ln u < x: with ln u drawn ahead, x outside [ln u - m, ln u + m] (m = 1e-9 +
                // 1e-12 |ln u|, six orders above the error of either function) decides without the exponential; inside (probability
                // 1e-9 per evaluation), for x < -700 and for u = 0 (m = inf) the exponential is evaluated — by every lane, so the
                // verdicts are det_exp's verdicts bit for bit in all cases.
                const double Lu = sh_L[lane][r];
                const double Lm = 1e-9 - 1e-12 * Lu, Lhi = Lu + Lm, Llo = Lu - Lm;
                // (the verdicts of the 64 attempts as a lane mask: kept in scalar registers, never round-tripped through a vector bool)
                const unsigned long long vm = nv >= kSkW ? ~0ull : (1ull << nv) - 1ull;        // the block's valid attempts
                auto verdict = [&](const double x) -> unsigned long long {
                    // one compare per ballot, the masks combined by scalar instructions (a ballot of a compound condition goes through a vector bool)
                    const unsigned long long ge0 = __builtin_amdgcn_ballot_w64(x >= 0.0), hi = __builtin_amdgcn_ballot_w64(x > Lhi);
                    const unsigned long long lo = __builtin_amdgcn_ballot_w64(x < Llo), fin = __builtin_amdgcn_ballot_w64(x > -700.0);
                    const unsigned long long sure_acc = ge0 | hi, sure_rej = lo & fin;
                    unsigned long long m = vm & sure_acc;
                    unsigned long long need = vm & ~(sure_acc | sure_rej);
                    asm volatile("" : "+s"(need));                      // (opaque: "any lane" stays one scalar compare)
                    // a scalar branch the layout treats as cold: the exponential is needed once in ~10^9 evaluations
                    if (__builtin_expect(need != 0ull, 0)) m = vm & (ge0 | __builtin_amdgcn_ballot_w64(u < det_exp_v(x, expc)));
                    return m;
                };
                // sample points inside this block as a block-relative attempt index (a 32-bit scalar compare per accepted move instead of a
                // 64-bit one): the sample of iteration it is taken BEFORE its move, i.e. before attempt m = it - it0 - 1 is applied
                auto sample_rel = [&]() -> int { const int64_t d = next_sample[a] - it0 - 1; return d < (int64_t)kSkW ? (int)d : kSkW; };
                int ks = sample_rel();
                auto xof = [&](double fv) -> double { if constexpr (BIN) return -P.beta * (fv / P.sN); else return -P.beta * fv; };
                unsigned long long B = verdict(xof(f));
                while (B) {
                    const int k = __builtin_ctzll(B);
                    while (__builtin_expect(k >= ks, 0)) {              // sample BEFORE the move (RRRMC.jl:104-108)
                        if (P.Es && lane == 0) P.Es[ns[a] * Rp + grp * kSkRB + r8 + r] = E_run[a];
                        ns[a] += 1; next_sample[a] += P.step;
                        ks = sample_rel();
                    }
                    const double fk = sk_readlane_f64(f, k);
                    const double dE = BIN ? fk / P.sN : fk;             // delta_energy, SK.jl:278-284 (BIN: SK.jl:137-140)
                    const int32_t site_k = __builtin_amdgcn_readlane((int)sv, k);
                    const uint32_t spk = (uint32_t)__builtin_amdgcn_readlane((int)sp, k);
                    const bool swapped = mlast[a] == site_k;            // undo path of update_cache!, SK.jl:247-250
                    E_run[a] += dE; A_run[a] += 1;
                    const bool dup = (int32_t)sv == site_k;
                    // the attempt's word for the apply phase, written into lane k of accw (v_writelane: this compiler has no builtin for it)
                    const uint32_t aw = 1u | (swapped ? 0x100u : 0u) | (spk << 16);
                    // (one SGPR + M0: the constant bus takes one scalar source; M0 is the compiler's, so it is put back)
                    uint32_t m0_keep;
                    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(accw), "=&s"(m0_keep) : "s"(aw), "s"(k));
                    if (__builtin_expect(swapped, 0)) {
                        const double t = f; f = fl; fl = t;
                    } else {
                        const double d = sh_Jw[k * kSkW + lane];
                        const uint32_t neg = sp ^ spk ^ 1u;             // s_j != s_i (after the flip): sigma = -1
                        const double dl = __longlong_as_double(__double_as_longlong(d) ^ ((long long)neg << 63));
                        fl = f;
                        const double fn = f + dl;                       // lfields[j] = lfj + 4 sigma J, SK.jl:256-262
                        f = dup ? -fl : fn;                             // lfields[move] = -lfm, SK.jl:263-264
                        mlast[a] = site_k;
                    }
                    sp ^= dup ? 1u : 0u;
                    B = verdict(xof(f)) & ((~0ull << k) << 1);
                }
                while (next_sample[a] <= it0 + nv) {                    // the samples behind the block's last accepted move
                    if (P.Es && lane == 0) P.Es[ns[a] * Rp + grp * kSkRB + r8 + r] = E_run[a];
                    ns[a] += 1; next_sample[a] += P.step;
                }
                if (accw) atomicOr(&sh_acc[pb][lane], accw << r);       // bit r accepted, bit 8 + r swapped, bit 16 + r spin before the flip
            }
        }
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tD = __builtin_amdgcn_s_memtime();
#endif
        sk_lds_barrier();
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tE = __builtin_amdgcn_s_memtime();
#endif
        // ---- apply: the block's accepted moves on the registers; stage the next block meanwhile
        const uint32_t accv = sh_acc[pb][lane];
        const bool more = b + 1 < nblk;
        double jw_next[(kSkW * kSkW + NTH - 1) / NTH];
        if (more) {
#pragma unroll
            for (int e = 0; e < (kSkW * kSkW + NTH - 1) / NTH; ++e) jw_next[e] = P.blkJw[(b + 1) * (kSkW * kSkW) + e * NTH + tid];
            if (wv == 0) {
                sh_canon[sv] = 0xffu;                      // retire this block's window slots, then claim the next block's
                asm volatile("" ::: "memory");
                sh_canon[sv_next] = (uint8_t)lane;         // (a wavefront's LDS operations execute in order)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                sh_cslot[lane] = sh_canon[sv_next];
                sh_acc[pb ^ 1][lane] = 0u;
            }
            draw_uniforms(b + 1);
        }
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tF = __builtin_amdgcn_s_memtime();
#endif
        auto apply_step = [&](int k, const double (&d4)[SPT]) {
#ifdef RRRMC_SKB_ABL_W0
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)accv, k) & 0x80000000u;      // timing experiment: nobody accepts
#else
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)accv, k);
#endif
            const uint32_t site = (uint32_t)__builtin_amdgcn_readlane((int)sv, k);
            const uint32_t own = (int)((site & (NTH - 1)) >> 6) == wv ? (site >> LGN) + 1u : 0u;
            const unsigned long long lb = 1ull << (site & 63u);
#ifdef RRRMC_SKB_ABL_NOASM
            asm volatile("" :: "s"(w), "s"(own), "s"(lb), "v"(d4[0]));           // timing experiment: the step's prelude and row loads only
            if constexpr (SPT == 99)
#endif
            if constexpr (NTH == 1024) {          // (RB == 8, one or two sites per thread)
                if constexpr (SPT == 1) { SK_APPLY_ALL(SK_APPLY1, lf, lfl, sm, d4, SK_SCRATCH_LO); }
                else { SK_APPLY_ALL(SK_APPLY2, lf, lfl, sm, d4, SK_SCRATCH_LO); }
            } else if constexpr (RB == 8) {
                if constexpr (SPT == 1) { SK_APPLY_ALL(SK_APPLY1, lf, lfl, sm, d4, SK_SCRATCH_HI); }
                else if constexpr (SPT == 2) { SK_APPLY_ALL(SK_APPLY2, lf, lfl, sm, d4, SK_SCRATCH_HI); }
                else if constexpr (SPT == 3) { SK_APPLY_ALL(SK_APPLY3, lf, lfl, sm, d4, SK_SCRATCH_HI); }
                else { SK_APPLY_ALL(SK_APPLY4, lf, lfl, sm, d4, SK_SCRATCH_HI); }
            } else {
                if constexpr (SPT == 1) { SK_APPLY_HALF(SK_APPLY1, lf, lfl, sm, d4, SK_SCRATCH_HI); }
                else if constexpr (SPT == 2) { SK_APPLY_HALF(SK_APPLY2, lf, lfl, sm, d4, SK_SCRATCH_HI); }
                else if constexpr (SPT == 3) { SK_APPLY_HALF(SK_APPLY3, lf, lfl, sm, d4, SK_SCRATCH_HI); }
                else { SK_APPLY_HALF(SK_APPLY4, lf, lfl, sm, d4, SK_SCRATCH_HI); }
            }
        };
        for (int k0 = 0; k0 < kSkW; k0 += 2 * PF) {
#pragma unroll
            for (int kk = 0; kk < PF; ++kk) {           // rows of attempts k0 + PF .. k0 + 2 PF - 1
                const uint32_t stn = (uint32_t)__builtin_amdgcn_readlane((int)sv, k0 + PF + kk);
#pragma unroll
#if defined(RRRMC_SKB_HALFROW)
                for (int q = 0; q < SPT; ++q) JB[kk][q] = q < (SPT + 1) / 2 ? P.J4[(size_t)stn * P.ldJ + (q * NTH + tid)] : JB[kk][q - (SPT + 1) / 2];   // timing experiment: half the row traffic (wrong couplings)
#elif !defined(RRRMC_SKB_NOLOAD)
                for (int q = 0; q < SPT; ++q) JB[kk][q] = P.J4[(size_t)stn * P.ldJ + (q * NTH + tid)];
#else
                for (int q = 0; q < SPT; ++q) JB[kk][q] = P.J4[(size_t)(stn & 7u) * P.ldJ + (q * NTH + tid)];      // timing experiment: always-hot rows
#endif
            }
#pragma unroll
            for (int kk = 0; kk < PF; ++kk) apply_step(k0 + kk, JA[kk]);
            const uint32_t svn = k0 + 2 * PF < kSkW ? sv : sv_next;       // (the next block's first attempts past the end of this one)
#pragma unroll
            for (int kk = 0; kk < PF; ++kk) {           // rows of attempts k0 + 2 PF .. k0 + 3 PF - 1
                const uint32_t stn = (uint32_t)__builtin_amdgcn_readlane((int)svn, (k0 + 2 * PF + kk) & 63);
#pragma unroll
#if defined(RRRMC_SKB_HALFROW)
                for (int q = 0; q < SPT; ++q) JA[kk][q] = q < (SPT + 1) / 2 ? P.J4[(size_t)stn * P.ldJ + (q * NTH + tid)] : JA[kk][q - (SPT + 1) / 2];
#elif !defined(RRRMC_SKB_NOLOAD)
                for (int q = 0; q < SPT; ++q) JA[kk][q] = P.J4[(size_t)stn * P.ldJ + (q * NTH + tid)];
#else
                for (int q = 0; q < SPT; ++q) JA[kk][q] = P.J4[(size_t)(stn & 7u) * P.ldJ + (q * NTH + tid)];
#endif
            }
#pragma unroll
            for (int kk = 0; kk < PF; ++kk) apply_step(k0 + PF + kk, JB[kk]);
        }
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tG = __builtin_amdgcn_s_memtime();
#endif
        if (more) {
#pragma unroll
            for (int e = 0; e < (kSkW * kSkW + NTH - 1) / NTH; ++e) sh_Jw[e * NTH + tid] = jw_next[e];
        }
        sv = sv_next;
        sk_lds_barrier();
#ifdef RRRMC_SKB_STAMPS
        { const uint64_t tH = __builtin_amdgcn_s_memtime();
          st[0] += tB - tA; st[1] += tC - tB; st[2] += tD - tC; st[3] += tE - tD; st[4] += tF - tE; st[5] += tG - tF; st[6] += tH - tG; }
#endif
    }
#ifdef RRRMC_SKB_STAMPS
    if (grp == 1 && lane == 0 && (wv == 0 || wv == 5))
        printf("skb stamps wave %d blocks %lld: gather %llu bar1 %llu decide %llu bar2 %llu stage %llu apply %llu tail+bar3 %llu (memtime ticks)\n", wv, (long long)nblk,
               (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2], (unsigned long long)st[3], (unsigned long long)st[4],
               (unsigned long long)st[5], (unsigned long long)st[6]);
#endif

#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int j = q * NTH + tid;
        if (j < N) {
            if constexpr (RB == 8) {
                P.spins[(size_t)grp * N + j] = (uint8_t)spin_byte(q);
            } else {
                // the byte is shared with the group's other workgroup: change this one's four bits only (nobody else writes them, so the
                // byte in memory still holds their value from before the launch), atomically on the enclosing word
                const size_t off = (size_t)grp * N + j;
                const uint32_t was = (P.spins[off] >> r8) & 0xfu, sh = 8u * (uint32_t)(off & 3u) + (uint32_t)r8;
                atomicXor(reinterpret_cast<uint32_t*>(P.spins + (off & ~(size_t)3)), (was ^ spin_byte(q)) << sh);
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                P.lf[((size_t)grp * N + j) * kSkRB + r8 + r] = lf[q][r];
                P.lfl[((size_t)grp * N + j) * kSkRB + r8 + r] = lfl[q][r];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < RPW; ++a) {
        const int r = wv + a * NWV;
        if (r < RB && lane == 0) { P.E_cur[grp * kSkRB + r8 + r] = E_run[a]; P.acc_cur[grp * kSkRB + r8 + r] = A_run[a]; P.move_last[grp * kSkRB + r8 + r] = mlast[a]; }
    }
}

// ---- the binary model's state for sk_block_kernel<.., BIN = true>: int32 fields <-> doubles, bit rows -> the +-4.0 matrix (zero diagonal,
//      rows zero-padded to ldJ) ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void skb_fields_to_f64_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b, double* __restrict__ fa,
                                                                 double* __restrict__ fb, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { fa[i] = (double)a[i]; fb[i] = (double)b[i]; }
}
__global__ __launch_bounds__(256) void skb_fields_from_f64_kernel(const double* __restrict__ fa, const double* __restrict__ fb, int32_t* __restrict__ a,
                                                                   int32_t* __restrict__ b, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { a[i] = (int32_t)fa[i]; b[i] = (int32_t)fb[i]; }
}
__global__ __launch_bounds__(256) void skb_dense4_kernel(const uint32_t* __restrict__ Jbits, double* __restrict__ J4, int N, int NW, int ldJ)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= ldJ) return;
    double v = 0.0;
    if (j < N && j != i) v = ((Jbits[(size_t)i * NW + (j >> 5)] >> (j & 31)) & 1u) ? 4.0 : -4.0;
    J4[(size_t)i * ldJ + j] = v;
}

}  // namespace rrrmc
