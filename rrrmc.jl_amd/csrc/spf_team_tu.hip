// Translation unit of spf_team_kernel (spf_team_kernel.hpp) and its launchers (spf_team_api.hpp).
#include <mutex>
#include <set>

#include "spf_team_api.hpp"
#include "spf_team_kernel.hpp"

namespace rrrmc {

namespace {

constexpr size_t kLdsMax = (size_t)160 * 1024;

template <int K, int NW, int TW>
constexpr bool build_fits() { return spf_team_lds_bytes(K, NW, spf_team_slots(K, NW, TW), TW) <= kLdsMax; }

template <int K, int NW, int TW>
hipError_t launch_build(int W, hipStream_t st, const SpfTeamParams& TP)
{
    if constexpr (build_fits<K, NW, TW>()) {
        constexpr int M = spf_team_slots(K, NW, TW);
        constexpr size_t lds = spf_team_lds_bytes(K, NW, M, TW);
        auto fn = spf_team_kernel<K, NW, M, TW>;
        // hipFuncSetAttribute acts on the CURRENT device's copy of the kernel (the caller has made the context's device current): the limit is
        // raised once per (device, build), under a lock — contexts on different devices, driven by different host threads, share this code
        {
            static std::mutex mu;
            static std::set<int> raised;
            int dev = 0;
            hipError_t e = hipGetDevice(&dev);
            if (e != hipSuccess) return e;
            std::lock_guard<std::mutex> lock(mu);
            if (!raised.count(dev)) {
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                raised.insert(dev);
            }
        }
        hipLaunchKernelGGL(fn, dim3((unsigned)W * (64 / TW)), dim3(NW * 64), lds, st, TP);
        return hipGetLastError();
    } else {
        (void)W; (void)st; (void)TP;
        return hipErrorInvalidValue;
    }
}

template <int K>
hipError_t launch_K(int nw, int tw, int W, hipStream_t st, const SpfTeamParams& TP)
{
    if (nw == 16 && tw == 64) return launch_build<K, 16, 64>(W, st, TP);
    if (nw == 16 && tw == 32) return launch_build<K, 16, 32>(W, st, TP);
    if (nw == 16 && tw == 16) return launch_build<K, 16, 16>(W, st, TP);
    if (nw == 8 && tw == 64) return launch_build<K, 8, 64>(W, st, TP);
    if constexpr (K >= 7) {
        if (nw == 8 && tw == 32) return launch_build<K, 8, 32>(W, st, TP);
        if (nw == 8 && tw == 16) return launch_build<K, 8, 16>(W, st, TP);
    }
    return hipErrorInvalidValue;
}

template <int K>
size_t lds_K(int nw, int tw)
{
    if (nw == 16 && tw == 64) return build_fits<K, 16, 64>() ? spf_team_lds_bytes(K, 16, spf_team_slots(K, 16, 64), 64) : 0;
    if (nw == 16 && tw == 32) return build_fits<K, 16, 32>() ? spf_team_lds_bytes(K, 16, spf_team_slots(K, 16, 32), 32) : 0;
    if (nw == 16 && tw == 16) return build_fits<K, 16, 16>() ? spf_team_lds_bytes(K, 16, spf_team_slots(K, 16, 16), 16) : 0;
    if (nw == 8 && tw == 64) return build_fits<K, 8, 64>() ? spf_team_lds_bytes(K, 8, spf_team_slots(K, 8, 64), 64) : 0;
    if (K >= 7 && nw == 8 && tw == 32) return spf_team_lds_bytes(K, 8, spf_team_slots(K, 8, 32), 32);
    if (K >= 7 && nw == 8 && tw == 16) return spf_team_lds_bytes(K, 8, spf_team_slots(K, 8, 16), 16);
    return 0;
}

}  // namespace

#define SPF_TEAM_SWITCH_K(K, expr_of_KK)                                                                                  \
    switch (K) {                                                                                                          \
    case 1: { constexpr int KK = 1; return expr_of_KK; }                                                                  \
    case 2: { constexpr int KK = 2; return expr_of_KK; }                                                                  \
    case 3: { constexpr int KK = 3; return expr_of_KK; }                                                                  \
    case 4: { constexpr int KK = 4; return expr_of_KK; }                                                                  \
    case 5: { constexpr int KK = 5; return expr_of_KK; }                                                                  \
    case 6: { constexpr int KK = 6; return expr_of_KK; }                                                                  \
    case 7: { constexpr int KK = 7; return expr_of_KK; }                                                                  \
    case 8: { constexpr int KK = 8; return expr_of_KK; }                                                                  \
    default: break;                                                                                                       \
    }

size_t spf_team_build_lds(int K, int nw, int tw)
{
    SPF_TEAM_SWITCH_K(K, lds_K<KK>(nw, tw))
    return 0;
}

hipError_t spf_team_plan_launch(const int32_t* A, const double* J, const int32_t* sites, uint32_t* plan, int64_t n, int K, hipStream_t st)
{
    hipLaunchKernelGGL(spf_team_plan_kernel, dim3((unsigned)((n + 2 + 255) / 256)), dim3(256), 0, st, A, J, sites, plan, n, K);
    return hipGetLastError();
}

hipError_t spf_team_launch(int K, int nw, int tw, int W, hipStream_t st, const SpfTeamParams& TP)
{
    SPF_TEAM_SWITCH_K(K, launch_K<KK>(nw, tw, W, st, TP))
    return hipErrorInvalidValue;
}

}  // namespace rrrmc
