// Observables on device-resident configurations:
//   per-replica overlaps pm1dot(a, b) = N - 2|a xor b| between stored snapshots (scripts/scripts.jl:283-295),
//   GraphQuant's integer parts of transverse_mag / overlaps / Qenergy (src/graphs/QT.jl:113-122, 213-268).
// A snapshot is a copy of the model's native spin buffer, so each layout has its own popcount kernel:
//   sparse +-J : [G][N] uint32, bit r&31 of word (r>>5, x)        -> 32x32 bit transpose + popcount
//   SK         : [G8][N] uint8, bit r&7 of byte (r>>3, x)         -> 8 counters per thread
//   GraphQuant : [R][W] uint32 in BitVector chunk order           -> plain popcount
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sparse_kernels.hpp"

namespace rrrmc {

// grid (G, npairs), block 256.  a/b: snapshot bases already offset by the host (device pointer tables).
__global__ __launch_bounds__(256) void overlap_bs32_kernel(const uint32_t* const* __restrict__ srcA, const uint32_t* const* __restrict__ srcB,
                                                           int N, int Rpad, int32_t* __restrict__ out)
{
    __shared__ uint32_t part[4][32];
    const int g = blockIdx.x, p = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t* a = srcA[p] + (size_t)g * N;
    const uint32_t* b = srcB[p] + (size_t)g * N;
    TransposeConsts tc;
    tc.init(lane);
    uint32_t cnt = 0u;
    for (int x0 = 0; x0 < N; x0 += 256) {          // uniform trip count: the transpose exchanges data across lanes
        const int x = x0 + (int)threadIdx.x;
        const uint32_t w = x < N ? (a[x] ^ b[x]) : 0u;
        cnt += (uint32_t)__popc(transpose32(w, tc));
    }
    cnt += (uint32_t)__shfl_xor((int)cnt, 32);
    if (lane < 32) part[wave][lane] = cnt;
    __syncthreads();
    if (threadIdx.x < 32) {
        const uint32_t tot = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        out[(size_t)p * Rpad + g * 32 + threadIdx.x] = N - 2 * (int32_t)tot;
    }
}

// grid (G8, npairs), block 256
__global__ __launch_bounds__(256) void overlap_b8_kernel(const uint8_t* const* __restrict__ srcA, const uint8_t* const* __restrict__ srcB,
                                                         int N, int Rpad, int32_t* __restrict__ out)
{
    __shared__ int32_t tot[8];
    const int g = blockIdx.x, p = blockIdx.y;
    const uint8_t* a = srcA[p] + (size_t)g * N;
    const uint8_t* b = srcB[p] + (size_t)g * N;
    if (threadIdx.x < 8) tot[threadIdx.x] = 0;
    __syncthreads();
    int32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int x = threadIdx.x; x < N; x += 256) {
        const uint32_t w = (uint32_t)(a[x] ^ b[x]);
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] += (int32_t)((w >> j) & 1u);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int32_t v = c[j];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) atomicAdd(&tot[j], v);
    }
    __syncthreads();
    if (threadIdx.x < 8) out[(size_t)p * Rpad + g * 8 + threadIdx.x] = N - 2 * tot[threadIdx.x];
}

// grid (R, npairs), block 64
__global__ __launch_bounds__(64) void overlap_chunks_kernel(const uint32_t* const* __restrict__ srcA, const uint32_t* const* __restrict__ srcB,
                                                            int N, int W, int Rpad, int32_t* __restrict__ out)
{
    const int r = blockIdx.x, p = blockIdx.y;
    const uint32_t* a = srcA[p] + (size_t)r * W;
    const uint32_t* b = srcB[p] + (size_t)r * W;
    int32_t v = 0;
    for (int w = threadIdx.x; w < W; w += 64) v += __popc(a[w] ^ b[w]);
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (threadIdx.x == 0) out[(size_t)p * Rpad + r] = N - 2 * v;
}

// GraphQuant integer observables of every replica.  grid R, block 256, dynamic LDS = (M*WS + M + M/2 + 1) words.
//   e0[r]            = energy0(X0, C) = -sum_j sigma_j sigma_{j+Nk}                       (QT.jl:68-82)
//   Eslice[r][k]     = energy(X1[k], C1[k]) (RRG.jl:164-189), or its integer n for GraphSK slices: E = n / sqrt(Nk) (SK.jl:62-96)
//   ovs_raw[r][d-1]  = sum over slice pairs at ring distance d of pm1dot(slice k1, slice k2)   (QT.jl:213-233)
__global__ __launch_bounds__(256) void quant_observables_kernel(const uint32_t* __restrict__ spins, const int32_t* __restrict__ A,
                                                                const int8_t* __restrict__ J, const uint32_t* __restrict__ Jb, int Wk,
                                                                int Nk, int M, int K, int W,
                                                                int32_t* __restrict__ e0, int32_t* __restrict__ Eslice,
                                                                int32_t* __restrict__ ovs_raw)
{
    extern __shared__ uint32_t qo_lds[];
    const int WS = (Nk + 31) >> 5;
    uint32_t* S = qo_lds;                        // [M][WS] slice-aligned words
    int32_t* es = (int32_t*)(S + M * WS);        // [M]
    int32_t* ov = es + M;                        // [M/2]
    int32_t* pe0 = ov + (M >> 1);                // [1]
    const int r = blockIdx.x;
    const uint32_t* sp = spins + (size_t)r * W;
    for (int t = threadIdx.x; t < M * WS; t += 256) {
        const int k = t / WS, w = t - k * WS;
        const int b = k * Nk + 32 * w;
        const int nb = Nk - 32 * w < 32 ? Nk - 32 * w : 32;
        const int q = b >> 5, sh = b & 31;
        uint32_t v = sp[q] >> sh;
        if (sh && q + 1 < W) v |= sp[q + 1] << (32 - sh);
        if (nb < 32) v &= (1u << nb) - 1u;
        S[t] = v;
    }
    for (int t = threadIdx.x; t < M + (M >> 1) + 1; t += 256) es[t] = 0;
    __syncthreads();
    // slice pair overlaps and the Trotter ring
    for (int t = threadIdx.x; t < M * M; t += 256) {
        const int k1 = t / M, k2 = t - k1 * M;
        const bool pair = k2 > k1;
        const bool ring = k2 == (k1 + M - 1) % M;      // (slice k1, its predecessor), every k1 once — also when M == 2
        if (!pair && !ring) continue;
        int32_t c = 0;
        for (int w = 0; w < WS; ++w) c += __popc(S[k1 * WS + w] ^ S[k2 * WS + w]);
        const int32_t o = Nk - 2 * c;
        if (pair) {
            const int d = k2 - k1 < M + k1 - k2 ? k2 - k1 : M + k1 - k2;
            atomicAdd(&ov[d - 1], o);
        }
        if (ring) atomicAdd(pe0, -o);
    }
    // slice energies: -(1/2) sum_x sum_j J[x][j] sigma_x sigma_y
    for (int t = threadIdx.x; t < M * Nk; t += 256) {
        const int k = t / Nk, x = t - k * Nk;
        const uint32_t* Sk = S + k * WS;
        const int sx = (int)((Sk[x >> 5] >> (x & 31)) & 1u);
        int32_t acc = 0;
        if (Jb) {                                // binary GraphSK slices (SK.jl:62-96): sum_{y != x} (2 J_xy - 1) sigma_x sigma_y by popcounts
            int sc = 0;
            for (int w = 0; w < WS; ++w) sc += __popc(Sk[w] ^ Jb[(size_t)x * Wk + w]);
            acc = (2 * sx - 1) * (Nk - 1 - 2 * (sc - sx));
        }
        for (int j = 0; j < K; ++j) {
            const int y = A[(size_t)x * K + j];
            const int sy = (int)((Sk[y >> 5] >> (y & 31)) & 1u);
            acc += (int32_t)J[(size_t)x * K + j] * (1 - 2 * (sx ^ sy));
        }
        if (acc) atomicAdd(&es[k], -acc);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < M; t += 256) Eslice[(size_t)r * M + t] = es[t] / 2;
    for (int t = threadIdx.x; t < (M >> 1); t += 256) ovs_raw[(size_t)r * (M >> 1) + t] = ov[t];
    if (threadIdx.x == 0) e0[r] = *pe0;
}

}  // namespace rrrmc
