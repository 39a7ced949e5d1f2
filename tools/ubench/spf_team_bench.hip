// Stand-alone harness for spf_team_kernel (csrc/spf_team_kernel.hpp): runs the same launches through spf_sweep_kernel (one wavefront per
// group, parity-tested against the oracle by tests/test_gpu_spf_parity.py) and through the team kernel, compares EVERYTHING both leave
// behind bit for bit (fields, spins, undo records, move_last, energies, samples, accepted counts) and times both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/spf_team_bench.hip -o tools/ubench/spf_team_bench.out
//   ./tools/ubench/spf_team_bench.out [K = 3 | 4 | 6] [N = 4096] [R = 8192] [iters = 16384] [beta = 1.0] [launches = 3] [NW = 16 | 8] [step = 4096] [M = slots | 0] [TW = 64 | 32 | 16]
// K = 3: ring + random perfect matching; K = 4 / 6: periodic square / cubic lattice with N = L^2 / L^3 sites.  Gaussian couplings.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../rrrmc.jl_amd/csrc/spf_kernels.hpp"
#include "../../rrrmc.jl_amd/csrc/spf_team_kernel.hpp"

using namespace rrrmc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct State {
    unsigned long long* spins; double *lf, *undo, *E, *Es; int32_t* ml; int64_t* acc;
};

static size_t g_trace_bytes = 0;
static int alloc_state(State& s, int W, int N, int K, size_t nsamp)
{
    CK(hipMalloc(&s.spins, sizeof(unsigned long long) * W * N));
    CK(hipMalloc(&s.lf, sizeof(double) * (size_t)W * N * 64));
    CK(hipMalloc(&s.undo, sizeof(double) * (size_t)W * (K + 1) * 64));
    CK(hipMalloc(&s.E, sizeof(double) * W * 64));
    // (the trace and stamp builds write their clocks into the sample buffer: 8 words per iteration of the longest launch)
    const size_t es_bytes = sizeof(double) * (nsamp + 1) * W * 64 + (size_t)g_trace_bytes;
    CK(hipMalloc(&s.Es, es_bytes));
    CK(hipMalloc(&s.ml, sizeof(int32_t) * W * 64));
    CK(hipMalloc(&s.acc, sizeof(int64_t) * W * 64));
    CK(hipMemset(s.undo, 0, sizeof(double) * (size_t)W * (K + 1) * 64));
    CK(hipMemset(s.acc, 0, sizeof(int64_t) * W * 64));
    CK(hipMemset(s.Es, 0, es_bytes));
    return 0;
}

static size_t g_N = 1;
template <typename T>
static long long diff(const T* a, const T* b, size_t n, const char* what)
{
    std::vector<T> ha(n), hb(n);
    if (hipMemcpy(ha.data(), a, sizeof(T) * n, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hb.data(), b, sizeof(T) * n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    long long bad = 0; size_t first = 0;
    for (size_t i = 0; i < n; ++i) if (memcmp(&ha[i], &hb[i], sizeof(T)) != 0) { if (!bad) first = i; ++bad; }
    if (bad) {
        printf("  MISMATCH %s: %lld of %zu differ, first at %zu\n", what, bad, n, first);
        int shown = 0;
        for (size_t i = first; i < n && shown < 6; ++i)
            if (memcmp(&ha[i], &hb[i], sizeof(T)) != 0) {
                unsigned long long ua = 0, ub = 0;
                memcpy(&ua, &ha[i], sizeof(T)); memcpy(&ub, &hb[i], sizeof(T));
                printf("    [%zu = group %zu site %zu lane %zu] %016llx vs %016llx", i, i / ((size_t)g_N * 64), i / 64 % g_N, i % 64, ua, ub);
                if (sizeof(T) == 8 && what[0] != 's') { double da, db; memcpy(&da, &ha[i], 8); memcpy(&db, &hb[i], 8); printf("  (%.17g vs %.17g)", da, db); }
                printf("\n");
                ++shown;
            }
    }
    return bad;
}

typedef void (*team_fn)(SpfTeamParams);
typedef void (*sweep_fn)(SpfParams);

int main(int argc, char** argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 3, N = argc > 2 ? atoi(argv[2]) : 4096, R = argc > 3 ? atoi(argv[3]) : 8192;
    const long long iters = argc > 4 ? atoll(argv[4]) : 16384;
    const double beta = argc > 5 ? atof(argv[5]) : 1.0;
    const int launches = argc > 6 ? atoi(argv[6]) : 3, NW = argc > 7 ? atoi(argv[7]) : 16;
    const long long step = argc > 8 ? atoll(argv[8]) : 4096;
    const int W = (R + 63) / 64;
    g_N = (size_t)N;
    srand48(4242 + N + K);

    std::vector<int32_t> A((size_t)N * K);
    std::vector<double> J((size_t)N * K);
    auto gauss = []() { const double u1 = drand48() + 1e-300, u2 = drand48(); return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2); };
    if (K == 3) {
        if (N % 2 || N < 6) { fprintf(stderr, "K = 3 needs an even N >= 6\n"); return 1; }
        std::vector<int> mate(N);
        for (;;) {
            std::vector<int> perm(N);
            for (int i = 0; i < N; ++i) perm[i] = i;
            for (int i = N - 1; i > 0; --i) { const int j = (int)(lrand48() % (i + 1)); std::swap(perm[i], perm[j]); }
            bool ok = true;
            for (int a = 0; a < N; a += 2) {
                const int u = perm[a], v = perm[a + 1], d = (u - v + N) % N;
                if (d == 1 || d == N - 1) ok = false;
                mate[u] = v; mate[v] = u;
            }
            if (ok) break;
        }
        std::vector<double> Jr(N), Jm(N);
        for (int i = 0; i < N; ++i) Jr[i] = gauss();                     // bond i -- i+1
        for (int i = 0; i < N; ++i) if (mate[i] > i) { Jm[i] = gauss(); Jm[mate[i]] = Jm[i]; }
        for (int i = 0; i < N; ++i) {
            int nb[3] = {(i + N - 1) % N, (i + 1) % N, mate[i]};
            double jj[3] = {Jr[(i + N - 1) % N], Jr[i], Jm[i]};
            for (int a = 0; a < 3; ++a) for (int b = a + 1; b < 3; ++b) if (nb[b] < nb[a]) { std::swap(nb[a], nb[b]); std::swap(jj[a], jj[b]); }
            for (int k = 0; k < 3; ++k) { A[(size_t)i * 3 + k] = nb[k]; J[(size_t)i * 3 + k] = jj[k]; }
        }
    } else if (K == 4 || K == 6) {
        const int D = K / 2;
        int L = (int)std::lround(std::pow((double)N, 1.0 / D));
        if ((D == 2 ? L * L : L * L * L) != N || L < 3) { fprintf(stderr, "N must be L^%d with L >= 3\n", D); return 1; }
        std::vector<double> Jb((size_t)N * D);
        for (auto& v : Jb) v = gauss();                                   // bond from site i in direction +d
        for (int i = 0; i < N; ++i) {
            int c[3] = {i % L, (i / L) % L, i / (L * L)};
            int k = 0;
            for (int d = 0; d < D; ++d) {
                int cm[3] = {c[0], c[1], c[2]}, cp[3] = {c[0], c[1], c[2]};
                cm[d] = (c[d] + L - 1) % L; cp[d] = (c[d] + 1) % L;
                const int im = cm[0] + L * (cm[1] + L * cm[2]), ip = cp[0] + L * (cp[1] + L * cp[2]);
                A[(size_t)i * K + k] = im; J[(size_t)i * K + k] = Jb[(size_t)im * D + d]; ++k;
                A[(size_t)i * K + k] = ip; J[(size_t)i * K + k] = Jb[(size_t)i * D + d]; ++k;
            }
        }
    } else { fprintf(stderr, "K must be 3, 4 or 6\n"); return 1; }

    int32_t *dA, *sites; uint32_t* deps; double* dJ;
    const long long maxit = iters + 8;
    CK(hipMalloc(&dA, sizeof(int32_t) * A.size())); CK(hipMalloc(&dJ, sizeof(double) * J.size()));
    CK(hipMalloc(&sites, sizeof(int32_t) * (maxit + 2 * kSpfDepth))); CK(hipMalloc(&deps, sizeof(uint32_t) * (size_t)(maxit + 2) * spf_plan_stride(K)));
    CK(hipMemcpy(dA, A.data(), sizeof(int32_t) * A.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dJ, J.data(), sizeof(double) * J.size(), hipMemcpyHostToDevice));

    const size_t nsamp = (size_t)((iters + 8) * launches / step + 2);
    State sa, sb;
#if defined(SPF_TEAM_TRACE) || defined(SPF_TEAM_STAMPS)
    g_trace_bytes = (size_t)(iters + launches + 8) * 64 + 4096;
#endif
    if (alloc_state(sa, W, N, K, nsamp) || alloc_state(sb, W, N, K, nsamp)) return 1;
    std::vector<unsigned long long> hs((size_t)W * N);
    for (auto& v : hs) v = ((unsigned long long)lrand48() << 42) ^ ((unsigned long long)lrand48() << 21) ^ (unsigned long long)lrand48();
    CK(hipMemcpy(sa.spins, hs.data(), sizeof(unsigned long long) * hs.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(sb.spins, hs.data(), sizeof(unsigned long long) * hs.size(), hipMemcpyHostToDevice));

    sweep_fn sweep = K == 3 ? spf_sweep_kernel<3> : K == 4 ? spf_sweep_kernel<4> : spf_sweep_kernel<6>;
    sweep_fn energy = K == 3 ? spf_energy_kernel<3> : K == 4 ? spf_energy_kernel<4> : spf_energy_kernel<6>;
    // M = slots (0: the build's default, spf_team_slots); TW = replicas per team (64 | 32 | 16)
    int M = argc > 9 ? atoi(argv[9]) : 0;
    const int TW = argc > 10 ? atoi(argv[10]) : 64;
    team_fn team = nullptr;
#define TEAM_B(KK, NWW, MM, TT) (K == KK && NW == NWW && M == MM && TW == TT) team = spf_team_kernel<KK, NWW, MM, TT>
#define TEAM_D(KK, NWW, TT) TEAM_B(KK, NWW, spf_team_slots(KK, NWW, TT), TT)
    if (M == 0) M = spf_team_slots(K, NW, TW);
    if TEAM_D(3, 16, 64); else if TEAM_D(4, 16, 64); else if TEAM_D(3, 8, 64); else if TEAM_D(4, 8, 64); else if TEAM_D(6, 8, 64);
    else if TEAM_D(3, 16, 32); else if TEAM_D(4, 16, 32); else if TEAM_D(6, 16, 32);
    else if TEAM_D(3, 16, 16); else if TEAM_D(4, 16, 16); else if TEAM_D(6, 16, 16);
    else if TEAM_B(3, 16, 47, 64); else if TEAM_B(3, 8, 28, 64); else if TEAM_B(3, 16, 60, 32); else if TEAM_B(3, 16, 45, 32);
    if (!team) { fprintf(stderr, "no such build\n"); return 1; }
    const size_t lds = spf_team_lds_bytes(K, NW, M, TW);
    printf("NW %d M %d TW %d: %zu bytes of LDS\n", NW, M, TW, lds);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(team), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    auto params = [&](State& s) {
        SpfParams P{};
        P.A = dA; P.J = dJ; P.sites = sites; P.spins = s.spins; P.lf = s.lf; P.undo = s.undo; P.move_last = s.ml; P.E_cur = s.E; P.acc_cur = s.acc; P.Es = s.Es;
        P.beta = beta; P.k0 = 0x5EED; P.k1 = 7; P.replica0 = 0; P.N = N; P.Rpad = W * 64; P.step = step;
        return P;
    };
    hipLaunchKernelGGL(energy, dim3(W), dim3(64), 0, 0, params(sa));
    hipLaunchKernelGGL(energy, dim3(W), dim3(64), 0, 0, params(sb));
    CK(hipDeviceSynchronize());

    int32_t* d_status; CK(hipMalloc(&d_status, sizeof(int32_t))); CK(hipMemset(d_status, 0, sizeof(int32_t)));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    uint64_t g0 = 0;
    long long off = 0, bad = 0;
    for (int l = 0; l < launches; ++l) {
        const long long n = iters + (l % 3);                 // odd and even lengths, odd and even stream offsets
        hipLaunchKernelGGL(spf_sites_kernel, dim3((unsigned)((n + 2 * kSpfDepth + 255) / 256)), dim3(256), 0, 0, sites, n + 2 * kSpfDepth, g0, 0x5EEDu, 7u, (uint32_t)N);
        hipLaunchKernelGGL(spf_team_plan_kernel, dim3((unsigned)((n + 2 + 255) / 256)), dim3(256), 0, 0, dA, dJ, sites, deps, n, K);
        SpfParams Pa = params(sa), Pb = params(sb);
        Pa.g0 = Pb.g0 = g0; Pa.iters = Pb.iters = n; Pa.it_off = Pb.it_off = off; Pa.sample0 = Pb.sample0 = off / step;
        SpfTeamParams TP{Pb, deps, d_status};
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(sweep, dim3(W), dim3(64), 0, 0, Pa);
        CK(hipEventRecord(e1, 0));
        hipLaunchKernelGGL(team, dim3(W * (64 / TW)), dim3(NW * 64), lds, 0, TP);
        CK(hipEventRecord(e2, 0));
        CK(hipDeviceSynchronize());
        float ma = 0.f, mb = 0.f;
        CK(hipEventElapsedTime(&ma, e0, e1)); CK(hipEventElapsedTime(&mb, e1, e2));
        g0 += (uint64_t)n; off += n;
        printf("launch %d: %lld iterations x %d replicas  sweep %.3f ms (%.3e attempts/s)  team %.3f ms (%.3e attempts/s)\n", l, n, W * 64, ma,
               (double)W * 64 * n / (ma * 1e-3), mb, (double)W * 64 * n / (mb * 1e-3));
        { int32_t hs_ = 0; CK(hipMemcpy(&hs_, d_status, sizeof hs_, hipMemcpyDeviceToHost)); if (hs_) { printf("  ABORTED: a wait ran into its limit\n"); ++bad; } }
        bad += diff(sa.lf, sb.lf, (size_t)W * N * 64, "lfields");
        bad += diff(sa.spins, sb.spins, (size_t)W * N, "spins");
        bad += diff(sa.undo, sb.undo, (size_t)W * (K + 1) * 64, "undo");
        bad += diff(sa.ml, sb.ml, (size_t)W * 64, "move_last");
        bad += diff(sa.E, sb.E, (size_t)W * 64, "E");
        bad += diff(sa.acc, sb.acc, (size_t)W * 64, "accepted");
#ifdef SPF_TEAM_TRACE
        if (l == launches - 1) {
            std::vector<unsigned long long> tr((size_t)(n + 1) * 8);
            CK(hipMemcpy(tr.data(), sb.Es, sizeof(unsigned long long) * tr.size(), hipMemcpyDeviceToHost));
            FILE* f = fopen("gpurun_out/spf_trace.txt", "w");
            const long long a0 = n / 2, a1 = a0 + 600;
            if (f) {
                fprintf(f, "# it wave request work_off decided undo_checked slot_free stores_issued reported retired   (cycles from the first line's request)\n");
                const unsigned long long t0 = tr[(size_t)a0 * 8];
                for (long long t = a0; t < a1 && t <= n; ++t) {
                    fprintf(f, "%lld %d", t, (int)((((unsigned long long)t + (g0 - (uint64_t)n)) >> 1) - ((g0 - (uint64_t)n + 1) >> 1)) % (NW - 1));
                    for (int q = 0; q < 8; ++q) fprintf(f, " %lld", (long long)(tr[(size_t)t * 8 + q] - t0));
                    fprintf(f, "\n");
                }
                fclose(f);
            }
        }
#elif defined(SPF_TEAM_STAMPS)
        {
            std::vector<unsigned long long> hst(32 * 8);
            CK(hipMemcpy(hst.data(), sb.Es, sizeof(unsigned long long) * hst.size(), hipMemcpyDeviceToHost));
            const char* nm[5] = {"prep", "dep/slot wait", "loads+decide", "update", "store ack+report"};
            printf("  retire: %llu looks at the flags, %llu found nothing, %llu cycles: flag checks (idle) %llu, flag checks (productive) %llu, data reads %llu, processing %llu\n", hst[(NW - 1) * 8], hst[(NW - 1) * 8 + 1], hst[(NW - 1) * 8 + 2], hst[(NW - 1) * 8 + 3], hst[(NW - 1) * 8 + 4], hst[(NW - 1) * 8 + 5], hst[(NW - 1) * 8 + 6]);
            for (int xw = 0; xw < NW - 1; xw += (getenv("SPF_ALL_WAVES") ? 1 : NW - 2)) {
                printf("  wave %d (%llu attempts), cycles per attempt:", xw, hst[xw * 8 + 5]);
                for (int q = 0; q < 5; ++q) printf("  %s %.1f", nm[q], (double)hst[xw * 8 + q] / (double)hst[xw * 8 + 5]);
                printf("\n");
                const unsigned long long* g = &hst[(NW + xw) * 8];
                printf("    gate waits: window %llu (%.0f cycles each), conflict %llu (%.0f each); slot waits %llu (%.0f each)\n", g[0], g[0] ? (double)g[1] / g[0] : 0.0,
                       g[2], g[2] ? (double)g[3] / g[2] : 0.0, g[4], g[4] ? (double)g[5] / g[4] : 0.0);
            }
        }
#else
        bad += diff(sa.Es, sb.Es, (size_t)(off / step) * W * 64, "samples");
#endif
    }
    std::vector<int64_t> a((size_t)W * 64);
    CK(hipMemcpy(a.data(), sa.acc, sizeof(int64_t) * a.size(), hipMemcpyDeviceToHost));
    double s = 0; for (auto v : a) s += (double)v;
    printf("acceptance %.4f   %s\n", s / a.size() / (double)off, bad ? "FAILED" : "identical");
    return bad ? 2 : 0;
}
