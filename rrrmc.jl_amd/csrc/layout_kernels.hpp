// Conversion between the caller's Config layout (R x ceil(N/64) BitVector chunks, src/Interface.jl:21-29, src/Common.jl:15-23)
// and the models' bit-sliced device layouts word[group][site] (bit b of the word = replica group * W + b), on the device:
// one wavefront per (64-site chunk, replica group); the 64-lane ballot of bit b IS chunk c of replica group * W + b.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rrrmc {

// native[g][x] (W replicas per word) -> chunks[r][c]
template <typename WordT>
__global__ __launch_bounds__(64) void spins_pack_kernel(const WordT* __restrict__ native, unsigned long long* __restrict__ chunks, int N, int nch, int R)
{
    constexpr int W = (int)sizeof(WordT) * 8;
    const int c = (int)blockIdx.x, g = (int)blockIdx.y, lane = (int)threadIdx.x;
    const int x = c * 64 + lane;
    const WordT w = x < N ? native[(size_t)g * (size_t)N + (size_t)x] : (WordT)0;
    unsigned long long mine = 0ull;
#pragma unroll
    for (int b = 0; b < W; ++b) {
        const unsigned long long m = __ballot((int)((w >> b) & (WordT)1));
        if (lane == b) mine = m;
    }
    const int r = g * W + lane;
    if (lane < W && r < R) chunks[(size_t)r * (size_t)nch + (size_t)c] = mine;
}

// chunks[r][c] -> native[g][x]; replicas beyond R (padding of the last group) read as 0
template <typename WordT>
__global__ __launch_bounds__(64) void spins_unpack_kernel(const unsigned long long* __restrict__ chunks, WordT* __restrict__ native, int N, int nch, int R)
{
    constexpr int W = (int)sizeof(WordT) * 8;
    const int c = (int)blockIdx.x, g = (int)blockIdx.y, lane = (int)threadIdx.x;
    const int x = c * 64 + lane;
    WordT w = 0;
#pragma unroll
    for (int b = 0; b < W; ++b) {
        const int r = g * W + b;
        const unsigned long long m = r < R ? chunks[(size_t)r * (size_t)nch + (size_t)c] : 0ull;      // wave-uniform load
        w |= (WordT)((m >> lane) & 1ull) << b;
    }
    if (x < N) native[(size_t)g * (size_t)N + (size_t)x] = w;
}

}  // namespace rrrmc
