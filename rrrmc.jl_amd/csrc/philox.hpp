// Philox4x32-10 counter RNG and the random-stream contract of the engine (host + gfx950 device).
//
// The reference draws from Julia's global RNG (src/RRRMC.jl:39,89,113; src/Interface.jl:26), which is
// neither pinned nor portable; every such draw is replaced by an addressed Philox draw so that a chain
// is a pure function of (seed, replica id, iteration) and never of the sharding or the launch shape.
// The contract is specified in DESIGN.md ("Random-stream contract"); oracle/philox_contract.h is an
// independent restatement used only by the tests.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RRRMC_HD __host__ __device__ __forceinline__
#else
#define RRRMC_HD inline
#endif

namespace rrrmc {

enum StreamTag : uint32_t {
    TAG_SITE = 1,      // attempted site of iteration g (shared by all replicas)
    TAG_ACCEPT = 2,    // bit planes of the acceptance uniforms of a 32-replica group
    TAG_INIT = 3,      // initial spins
    TAG_GRAPH = 4,     // random-regular-graph pairing
    TAG_COUPLING = 5,  // +-J couplings
};

struct Philox4 { uint32_t w[4]; };

// a ^ b ^ c in one VALU instruction: gfx950's v_bitop3_b32 with truth table 0x96 (the compiler keeps two v_xor_b32 when it is
// given the plain expression; a Philox round has two such sums next to its two v_mad_u64_u32, so this is a fifth of the block)
RRRMC_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

// any three-input bitwise function in one instruction: bit (a << 2 | b << 1 | c) of the truth table TT is f(a, b, c)
template <uint32_t TT>
RRRMC_HD uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
    uint32_t r = 0u;
    for (uint32_t m = 0; m < 8u; ++m)
        if ((TT >> m) & 1u) r |= ((m & 4u) ? a : ~a) & ((m & 2u) ? b : ~b) & ((m & 1u) ? c : ~c);
    return r;
#endif
}

RRRMC_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

RRRMC_HD uint64_t mulhi64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// n-th 64-bit word of a sequential stream: words (2h, 2h+1) of counter (lo(n>>1), hi(n>>1), 0, tag), h = n & 1
RRRMC_HD uint64_t stream_u64(uint32_t k0, uint32_t k1, uint32_t tag, uint64_t n)
{
    const uint64_t blk = n >> 1;
    const Philox4 o = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), 0u, tag, k0, k1);
    return (n & 1u) ? (((uint64_t)o.w[2] << 32) | o.w[3]) : (((uint64_t)o.w[0] << 32) | o.w[1]);
}

// site attempted at global iteration g (1-based): floor(u64 * N / 2^64)   [rand(1:N), src/RRRMC.jl:113]
RRRMC_HD uint32_t site_of(uint32_t k0, uint32_t k1, uint64_t g, uint32_t N)
{
    return (uint32_t)mulhi64(stream_u64(k0, k1, TAG_SITE, g), (uint64_t)N);
}

// four consecutive bit planes (4*pb .. 4*pb+3, plane 0 = MSB) of the acceptance uniforms of replica group
// `group` at iteration g: bit b of each word belongs to replica 32*group + b.
RRRMC_HD Philox4 accept_planes(uint32_t k0, uint32_t k1, uint64_t g, uint32_t group, uint32_t pb)
{
    return philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), group, (uint32_t)TAG_ACCEPT | (pb << 8), k0, k1);
}

// initial spin word (32 replicas of `group`) of site x
RRRMC_HD uint32_t init_spin_word(uint32_t k0, uint32_t k1, uint32_t group, uint64_t x)
{
    const Philox4 o = philox4x32_10((uint32_t)(x >> 2), 0u, group, TAG_INIT, k0, k1);
    return o.w[x & 3];
}

}  // namespace rrrmc
