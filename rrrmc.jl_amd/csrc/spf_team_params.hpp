// Parameters and LDS geometry of spf_team_kernel (spf_team_kernel.hpp), shared by the kernel's translation unit (spf_team_tu.hip) and the host
// code that launches it through spf_team_api.hpp; no device code here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spf_params.hpp"

namespace rrrmc {

constexpr int kSpfTeamWindow = 64;          // dependency window of spf_team_plan_kernel: >= the slots of every build (attempts in flight)

// The attempts of one launch, state-independent, one record of 4 + 3 K dwords per iteration `it` (record 0 and the records behind the last
// iteration are padding, so that an executing wavefront fetches the two records of a pair with one load):
//   [0] site   [1] same   [2] [3] conflicts   [4 .. 4 + K) the neighbours   [4 + K .. 4 + 3 K) their couplings (Float64, low word first)
// conflicts: bit c - 1 set <=> the closed neighbourhood of iteration it - c (c = 1 .. 64, the window) meets this one's, i.e. the two sites are at
// distance <= 2: the two attempts do not commute.  same = the latest earlier iteration within the window at the SAME site (0 = none).
__host__ __device__ constexpr int spf_plan_stride(int K) { return 4 + 3 * K; }

struct SpfTeamParams {
    SpfParams S;
    const uint32_t* plan;       // [iters + 2][4 + 3 K]  (spf_plan_stride)
    int32_t* status;            // one word per context, may be null: set to 1 by a workgroup whose wait ran into kSpfTeamSpinLimit (a protocol
                                // failure: the launch then runs to its end without waiting and its results are void)
};
constexpr int32_t kSpfTeamSpinLimit = 1 << 22;      // polls of one wait (each >= 64 cycles asleep): ~ 0.3 s, a thousand times the longest legitimate wait

// record areas: [0, M) the slots, M + x the KEEP of executing wavefront x, M + NX the undo records the launch starts with.  Per area and replica:
// the K neighbour fields before the move, the own field (+0.0 for a replica that did not accept) and the attempt's tag (its launch-relative
// iteration, 0 for a replica that did not accept)
__host__ __device__ constexpr int spf_team_areas(int NW, int M) { return M + (NW - 1) + 1; }
__host__ __device__ constexpr size_t spf_team_lds_bytes(int K, int NW, int M, int TW)
{
    return (sizeof(double) * (size_t)(K + 1) * TW + sizeof(uint32_t) * TW) * (size_t)spf_team_areas(NW, M) + sizeof(int32_t) * (size_t)(TW + 2 * M + 4);      // tl, done, ev, prefix + abort flag
}
// M = slots = attempts in flight at most.  Sixteen-wavefront teams own a compute unit: as many slots as its 160 KiB of LDS hold beside the keep
// areas, at most 60 (one wavefront read brings all flags; the planner's window is 64) — measured 3-4 % over one pair per wavefront.
// Eight-wavefront teams (many groups) stay at one pair per executing wavefront: small teams leave room for three workgroups per compute unit.
__host__ __device__ constexpr int spf_team_slots(int K, int NW, int TW)
{
    int m = NW == 16 ? 60 : (TW < 64 ? 42 : 2 * (NW - 1));          // (eight wavefronts narrower than a group: the fused K = 7, 8 builds, one team per compute unit)
    while (m > 2 * (NW - 1) && spf_team_lds_bytes(K, NW, m, TW) > (size_t)160 * 1024) --m;
    return m;
}

}  // namespace rrrmc
