// Parameters of the kernels for the Float64-coupling sparse models (spf_kernels.hpp, spf_team_kernel.hpp); no device code here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rrrmc {

constexpr int kSpfMaxK = 8;
#ifndef RRRMC_SPF_DEPTH
#define RRRMC_SPF_DEPTH 4
#define RRRMC_SPF_NB 2
#endif
constexpr int kSpfDepth = RRRMC_SPF_DEPTH;          // prefetch distance of the attempted site's field, in iterations
constexpr int kSpfNb = RRRMC_SPF_NB;             // prefetch distance of the neighbour fields / spin words

struct SpfParams {
    const int32_t* A;       // [N][K]
    const double* J;        // [N][K]
    const int32_t* sites;   // [iters + 2 kSpfDepth] sites of this launch's iterations and of the 2 kSpfDepth following ones
    unsigned long long* spins;   // [W][N]
    double* lf;             // [W][N][64]
    double* undo;           // [W][K+1][64]: saved neighbour fields (slot k) and own field (slot K) of the last accepted move
    int32_t* move_last;     // [Rpad], -1 = none
    double* E_cur;          // [Rpad]
    int64_t* acc_cur;       // [Rpad]
    double* Es;             // [nsamples][Rpad]; may be null
    double beta;
    uint64_t g0;            // iterations already consumed from the streams
    int64_t iters, step, sample0;
    int64_t it_off;         // iterations of this sampling call done by earlier launches (samples are taken at call-relative k*step)
    uint32_t k0, k1, replica0;
    int N, Rpad;
};

}  // namespace rrrmc
