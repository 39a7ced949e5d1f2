// Float64 helpers shared by every kernel that evaluates accept (src/RRRMC.jl:39) on a Float64 energy difference: the ACCEPT_F64 stream's
// uniform and exp / log1p with a fixed operation order (the oracle evaluates the same sequences, so `rand() < exp(x)` agrees bit for bit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"

namespace rrrmc {

constexpr uint32_t TAG_ACCEPT_F64 = 9;
constexpr uint32_t TAG_GAUSS = 6;

// rand() of replica `replica` at global iteration g: 53-bit uniform of the ACCEPT_F64 stream
RRRMC_HD double rand53(uint32_t k0, uint32_t k1, uint64_t g, uint32_t replica)
{
    const uint64_t blk = g >> 1;
    const Philox4 o = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), replica, TAG_ACCEPT_F64, k0, k1);
    const uint64_t u = (g & 1u) ? (((uint64_t)o.w[2] << 32) | o.w[3]) : (((uint64_t)o.w[0] << 32) | o.w[1]);
    return (double)(u >> 11) * 0x1.0p-53;
}

// exp(x) with a fixed operation order and no fused multiply-add: Cody-Waite reduction, degree-13 Taylor polynomial in
// Horner form, exact scaling.  The oracle evaluates the same sequence, so `rand() < exp(x)` agrees bit for bit.
__device__ __forceinline__ double det_exp(double x)
{
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double c[14] = {1.0, 1.0, 0.5, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
                          1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0};
    if (x != x) return x;
    if (x < -745.2) return 0.0;
    if (x > 709.7) return __builtin_inf();
    const double k = floor(__dadd_rn(__dmul_rn(x, LOG2E), 0.5));
    const double r = __dadd_rn(__dadd_rn(x, -__dmul_rn(k, LN2_HI)), -__dmul_rn(k, LN2_LO));
    double p = c[13];
#pragma unroll
    for (int n = 12; n >= 0; --n) p = __dadd_rn(__dmul_rn(p, r), c[n]);
    return ldexp(p, (int)k);
}

// det_exp with its constants handed in (c[0..13] the Taylor coefficients, c[14] = log2(e), c[15], c[16] = ln 2 high / low): the same
// operations in the same order.  A Float64 constant is two scalar moves at every use; a kernel that evaluates det_exp on its
// critical path once per step keeps them in vector registers instead (sk_exp_constants: an empty asm hides the values from constant
// propagation).
__device__ __forceinline__ void sk_exp_constants(double (&c)[17])
{
    const double v[17] = {1.0, 1.0, 0.5, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
                          1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0,
                          1.44269504088896338700e+00, 6.93147180369123816490e-01, 1.90821492927058770002e-10};
#pragma unroll
    for (int i = 0; i < 17; ++i) { c[i] = v[i]; asm volatile("" : "+v"(c[i])); }
}
// The same constants for a path that runs once in a million evaluations: each is moved into its registers by the asm statement itself, so
// the moves cannot be hoisted out of the enclosing loops (the compiler otherwise materialises all seventeen in the loop preheader and keeps
// them — or spills them — across the hot path).
template <unsigned long long B>
__device__ __forceinline__ double sk_const_here()
{
    uint32_t lo, hi;
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(lo), "=&v"(hi) : "i"((uint32_t)(B & 0xffffffffull)), "i"((uint32_t)(B >> 32)));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#define SK_CONST_HERE(x) sk_const_here<__builtin_bit_cast(unsigned long long, (double)(x))>()
__device__ __forceinline__ void sk_exp_constants_here(double (&c)[17])
{
    c[0] = SK_CONST_HERE(1.0); c[1] = SK_CONST_HERE(1.0); c[2] = SK_CONST_HERE(0.5); c[3] = SK_CONST_HERE(1.0 / 6); c[4] = SK_CONST_HERE(1.0 / 24);
    c[5] = SK_CONST_HERE(1.0 / 120); c[6] = SK_CONST_HERE(1.0 / 720); c[7] = SK_CONST_HERE(1.0 / 5040); c[8] = SK_CONST_HERE(1.0 / 40320);
    c[9] = SK_CONST_HERE(1.0 / 362880); c[10] = SK_CONST_HERE(1.0 / 3628800); c[11] = SK_CONST_HERE(1.0 / 39916800);
    c[12] = SK_CONST_HERE(1.0 / 479001600); c[13] = SK_CONST_HERE(1.0 / 6227020800.0); c[14] = SK_CONST_HERE(1.44269504088896338700e+00);
    c[15] = SK_CONST_HERE(6.93147180369123816490e-01); c[16] = SK_CONST_HERE(1.90821492927058770002e-10);
}
__device__ __forceinline__ double det_exp_c(double x, const double (&c)[17])
{
    if (x != x) return x;
    if (x < -745.2) return 0.0;
    if (x > 709.7) return __builtin_inf();
    const double k = floor(__dadd_rn(__dmul_rn(x, c[14]), 0.5));
    const double r = __dadd_rn(__dadd_rn(x, -__dmul_rn(k, c[15])), -__dmul_rn(k, c[16]));
    double p = c[13];
#pragma unroll
    for (int n = 12; n >= 0; --n) p = __dadd_rn(__dmul_rn(p, r), c[n]);
    return ldexp(p, (int)k);
}

// log1p(y) for -1 < y <= 0 with a fixed operation order (rand_skip of bklMC, src/DeltaE.jl:141-144); same sequence as the oracle's
__device__ __forceinline__ double det_log1p(double y)
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double u = __dadd_rn(1.0, y);
    if (u == 1.0) return y;
    if (!(u > 0.0)) return -__builtin_inf();
    unsigned long long b = (unsigned long long)__double_as_longlong(u);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    b = (b & ((1ull << 52) - 1)) | (1023ull << 52);
    double m = __longlong_as_double((long long)b);
    if (m > 1.4142135623730951) { m = __dmul_rn(m, 0.5); e += 1; }
    const double s = __ddiv_rn(__dadd_rn(m, -1.0), __dadd_rn(m, 1.0));
    const double s2 = __dmul_rn(s, s);
    double p = 1.0 / 21.0;
#pragma unroll
    for (int n = 19; n >= 1; n -= 2) p = __dadd_rn(__dmul_rn(p, s2), 1.0 / (double)n);
    const double lg = __dmul_rn(__dmul_rn(2.0, s), p);
    const double r = __dadd_rn(__dmul_rn((double)e, LN2_HI), __dadd_rn(lg, __dmul_rn((double)e, LN2_LO)));
    return __dadd_rn(r, -__ddiv_rn(__dadd_rn(__dadd_rn(u, -1.0), -y), u));
}

}  // namespace rrrmc
