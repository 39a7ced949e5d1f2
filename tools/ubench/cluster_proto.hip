// TIMING-ONLY prototype (wrong dynamics) of the site-partitioned cluster for random-site standardMC on a large lattice (DESIGN.md 4g / 8.6;
// VERDICT r4 item 6): config 4's GraphEA(64, 3) at one GPU's share, 512 replicas = 16 groups of 32 replicas; a group's 262 144 sites are divided
// over S = 16 workgroups (slabs of 4 lattice planes: 16 384 words of 32 replica bits = 64 KiB of LDS, plus two halo planes), so that one
// 32-bit instruction decides 32 attempts.  A chunk of C attempts is cut into dependency LEVELS by the (real) planner rule — two attempts
// conflict when one site is the other's or its neighbour — and every level ends with a barrier of the group's 16 workgroups through a
// counter in global memory, behind which the flips of boundary planes are handed to the two neighbouring slabs.
//   per level and workgroup: its share of the level's attempts, one per thread and pass: record (8 words) + masks (4 words) streamed from
//   global memory, 7 scattered LDS reads, ~40 bit operations, one LDS atomic XOR, boundary flips appended to an out-box in global memory.
// What is timed: µs per chunk with (a) everything, (b) barriers only, (c) work only — for chunks of 32 768 and 262 144 attempts.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/cluster_proto.hip -o tools/ubench/cluster_proto.out
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int L = 64, NS = L * L * L, S = 16, PLANES = L / S, SLAB = PLANES * L * L, HALO = L * L, G = 16, NT = 1024;
constexpr int kSpinLimit = 1 << 22;

struct Rec { uint32_t site, nb[6], pad; };          // local word indices into the slab image [0, SLAB + 2 HALO)

struct Params {
    const Rec* recs;            // all attempts of the chunk sorted by (level, slab, index)
    const uint32_t* start;      // [levels][S + 1] offsets into recs
    const uint4* masks;         // [C] pseudo acceptance masks (shared by the groups: timing only)
    uint32_t* counter;          // [G] monotonic barrier counters
    unsigned long long* outbox; // [G][S][2][cap] boundary flips (word index in the neighbour's halo | mask << 32)
    uint32_t* outcnt;           // [G][S][2][levels * chunks] entries written per (direction, level)
    uint32_t* sink;
    int levels, chunks, cap, mode;          // mode 0 everything, 1 barriers only, 2 work only
};

__global__ __launch_bounds__(NT) void cluster_kernel(Params P)
{
    extern __shared__ uint32_t img[];            // [SLAB + 2 HALO]
    __shared__ uint32_t ocnt[2];
    const int g = blockIdx.x / S, s = blockIdx.x % S, tid = threadIdx.x;
    for (int i = tid; i < SLAB + 2 * HALO; i += NT) img[i] = 0x9e3779b9u * (uint32_t)(i + s * 7919 + g);
    __syncthreads();
    uint32_t acc = 0, epoch = 0;
    for (int c = 0; c < P.chunks; ++c) {
        for (int lv = 0; lv < P.levels; ++lv, ++epoch) {
            if (tid < 2) ocnt[tid] = 0;
            __syncthreads();
            if (P.mode != 1) {
                const uint32_t a0 = P.start[lv * (S + 1) + s], a1 = P.start[lv * (S + 1) + s + 1];
                for (uint32_t a = a0 + tid; a < a1; a += NT) {
                    const Rec r = P.recs[a];
                    const uint4 m = P.masks[a];
                    const uint32_t si = img[r.site];
                    uint32_t n[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) n[k] = img[r.nb[k]] ^ si;
                    // bit-sliced count of unsatisfied bonds (3 planes) and a pseudo decision per class
                    uint32_t c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
                    for (int k = 0; k < 6; ++k) { const uint32_t x = n[k], t0 = c0 & x; c0 ^= x; const uint32_t t1 = c1 & t0; c1 ^= t0; c2 ^= t1; }
                    const uint32_t flip = (~c2 & ~c1) | (c2 & m.x) | (c1 & ~c2 & m.y) | (c0 & m.z & m.w);
                    atomicXor(&img[r.site], flip);
                    acc += __popc(flip);
                    const uint32_t plane = r.site / HALO;                       // boundary planes feed the neighbours' halos
                    if (flip && (plane == 0 || plane == PLANES - 1)) {
                        const int dir = plane == 0 ? 0 : 1;
                        const uint32_t slot = atomicAdd(&ocnt[dir], 1u);
                        if (slot < (uint32_t)P.cap)         // write-through (sc1) store: no L2 write-back needed before the counter (cdna_hip_programming.md G16, R1)
                            __hip_atomic_store(&P.outbox[(((size_t)g * S + s) * 2 + dir) * P.cap + slot], (unsigned long long)(r.site % HALO) | ((unsigned long long)flip << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            __syncthreads();
            if (P.mode != 2) {
                if (tid == 0) {
                    __hip_atomic_store(&P.outcnt[(((size_t)g * S + s) * 2 + 0) * 64 + (epoch & 63)], ocnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&P.outcnt[(((size_t)g * S + s) * 2 + 1) * 64 + (epoch & 63)], ocnt[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (every wave's out-box stores were drained by the barrier above)
#ifdef PROTO_FENCES
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                    __hip_atomic_fetch_add(&P.counter[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t want = (epoch + 1) * S;
                    int spins = 0;
                    while (__hip_atomic_load(&P.counter[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < kSpinLimit) __builtin_amdgcn_s_sleep(1);
#ifdef PROTO_FENCES
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                }
                __syncthreads();
                // the neighbours' boundary flips of this level into the halo planes
                if (tid < 2) {
                    const int dir = tid, nbr = dir == 0 ? (s + S - 1) % S : (s + 1) % S, ndir = 1 - dir;        // its upper (dir 1) / lower (dir 0) boundary faces this slab
                    ocnt[dir] = __hip_atomic_load(&P.outcnt[(((size_t)g * S + nbr) * 2 + ndir) * 64 + (epoch & 63)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
#pragma unroll
                for (int dir = 0; dir < 2; ++dir) {
                    const int nbr = dir == 0 ? (s + S - 1) % S : (s + 1) % S, ndir = 1 - dir;
                    const uint32_t cnt = ocnt[dir];
                    for (uint32_t e = tid; e < cnt && e < (uint32_t)P.cap; e += NT) {
                        const unsigned long long f = __hip_atomic_load(&P.outbox[(((size_t)g * S + nbr) * 2 + ndir) * P.cap + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        atomicXor(&img[SLAB + dir * HALO + (uint32_t)f], (uint32_t)(f >> 32));
                    }
                }
                __syncthreads();
            }
        }
    }
    if (acc == 0x12345u) P.sink[0] = acc;
}

int main(int argc, char** argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 32768, chunks = argc > 2 ? atoi(argv[2]) : 8;
    srand48(7);
    std::vector<int> site(C), lv(C);
    std::vector<int> lw(NS, 0), lr(NS, 0);
    auto nb = [](int x, int k) { int c[3] = {x % L, (x / L) % L, x / (L * L)}; const int d = k / 2; c[d] = (c[d] + (k & 1 ? 1 : L - 1)) % L; return c[0] + L * (c[1] + L * c[2]); };
    int levels = 0;
    for (int t = 0; t < C; ++t) {
        const int i = (int)(lrand48() % NS);
        site[t] = i;
        int l = std::max(lw[i], lr[i]);
        for (int k = 0; k < 6; ++k) l = std::max(l, lw[nb(i, k)]);
        lv[t] = ++l;
        lw[i] = l; lr[i] = std::max(lr[i], l);
        for (int k = 0; k < 6; ++k) lr[nb(i, k)] = std::max(lr[nb(i, k)], l);
        levels = std::max(levels, l);
    }
    std::vector<uint32_t> start((size_t)levels * (S + 1), 0);
    std::vector<int> order(C);
    for (int t = 0; t < C; ++t) order[t] = t;
    auto slab_of = [](int x) { return (x / (L * L)) / PLANES; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lv[a] != lv[b] ? lv[a] < lv[b] : slab_of(site[a]) < slab_of(site[b]); });
    std::vector<Rec> recs(C);
    size_t maxshare = 0;
    {
        size_t p = 0;
        for (int l = 1; l <= levels; ++l)
            for (int s = 0; s <= S; ++s) {
                start[(size_t)(l - 1) * (S + 1) + s] = (uint32_t)p;
                if (s == S) break;
                const size_t p0 = p;
                while (p < (size_t)C && lv[order[p]] == l && slab_of(site[order[p]]) == s) ++p;
                maxshare = std::max(maxshare, p - p0);
            }
    }
    for (int q = 0; q < C; ++q) {
        const int x = site[order[q]], s = slab_of(x), z0 = s * PLANES;
        auto local = [&](int y) { const int z = y / (L * L), dz = (z - z0 + L) % L; return dz < PLANES ? dz * HALO + y % HALO : dz == PLANES ? SLAB + HALO + y % HALO : SLAB + y % HALO; };
        recs[q].site = (uint32_t)local(x);
        for (int k = 0; k < 6; ++k) recs[q].nb[k] = (uint32_t)local(nb(x, k));
        recs[q].pad = 0;
    }
    std::vector<uint4> masks(C);
    for (auto& m : masks) m = make_uint4((uint32_t)mrand48(), (uint32_t)mrand48() & (uint32_t)mrand48(), (uint32_t)mrand48(), (uint32_t)mrand48());
    printf("chunk of %d attempts: %d levels, largest share of a level for one slab %zu attempts\n", C, levels, maxshare);

    Params P{};
    Rec* d_recs; uint32_t *d_start, *d_counter, *d_outcnt, *d_sink; uint4* d_masks; unsigned long long* d_out;
    const int cap = 8192;
    CK(hipMalloc(&d_recs, sizeof(Rec) * C)); CK(hipMalloc(&d_start, sizeof(uint32_t) * start.size())); CK(hipMalloc(&d_masks, sizeof(uint4) * C));
    CK(hipMalloc(&d_counter, sizeof(uint32_t) * G)); CK(hipMalloc(&d_outcnt, sizeof(uint32_t) * G * S * 2 * 64)); CK(hipMalloc(&d_sink, 4));
    CK(hipMalloc(&d_out, sizeof(unsigned long long) * (size_t)G * S * 2 * cap));
    CK(hipMemcpy(d_recs, recs.data(), sizeof(Rec) * C, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_start, start.data(), sizeof(uint32_t) * start.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_masks, masks.data(), sizeof(uint4) * C, hipMemcpyHostToDevice));
    P.recs = d_recs; P.start = d_start; P.masks = d_masks; P.counter = d_counter; P.outbox = d_out; P.outcnt = d_outcnt; P.sink = d_sink;
    P.levels = levels; P.chunks = chunks; P.cap = cap;
    const size_t lds = sizeof(uint32_t) * (SLAB + 2 * HALO);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(cluster_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"everything", "barriers only", "work only"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            P.mode = mode;
            CK(hipMemset(d_counter, 0, sizeof(uint32_t) * G)); CK(hipMemset(d_outcnt, 0, sizeof(uint32_t) * G * S * 2 * 64));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(cluster_kernel, dim3(G * S), dim3(NT), lds, 0, P);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us_chunk = ms * 1e3 / chunks;
            if (rep) printf("  %-14s %8.1f us per chunk  (%.2f us per level)  -> %.3e attempts/s at 512 replicas\n", names[mode], us_chunk, us_chunk / levels,
                            (double)C * 32 * G / (us_chunk * 1e-6));
        }
    return 0;
}
