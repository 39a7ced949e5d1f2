// Launchers of spf_team_kernel (spf_team_kernel.hpp), which lives in a translation unit of its own (spf_team_tu.hip: thirty-two builds of a
// long kernel, compiled beside the rest of the library instead of behind it).  Host-side interface only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spf_team_params.hpp"

namespace rrrmc {

// the builds: (wavefronts per team, replicas per team) = (16, 64) while the records fit the LDS (K <= 4), (16, 32), (16, 16), (8, 64)
// dynamic LDS of the build, 0 if there is none
size_t spf_team_build_lds(int K, int nw, int tw);
// spf_team_plan_kernel: the state-independent records of the n iterations of one launch
hipError_t spf_team_plan_launch(const int32_t* A, const double* J, const int32_t* sites, uint32_t* plan, int64_t n, int K, hipStream_t st);
// one launch: W groups of 64 replicas = W * (64 / tw) teams.  hipErrorInvalidValue if there is no such build
hipError_t spf_team_launch(int K, int nw, int tw, int W, hipStream_t st, const SpfTeamParams& TP);

}  // namespace rrrmc
