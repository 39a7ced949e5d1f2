// Host-only sanitizer build (g++ -fsanitize=address,undefined; SURVEY.md §5: sanitizers on the CPU build only) of the HIP-free host code
// of the library — the chunk planner and batch cutter of a sampling call, the acceptance thresholds, the shard bounds of a multi-device
// context, the Philox streams — driven over a sweep of call shapes with the invariants the kernels rely on checked for every shape.
// Exit status 0 = every invariant holds and neither sanitizer fired (they abort the process).
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../rrrmc.jl_amd/csrc/philox.hpp"
#include "../rrrmc.jl_amd/csrc/host_plan.hpp"

using namespace rrrmc;

#define REQUIRE(cond)                                                                                   \
    do {                                                                                                \
        if (!(cond)) { std::fprintf(stderr, "%s:%d: invariant violated: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
    } while (0)

static uint64_t lcg(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return s >> 11; }

static void check_plan(int64_t iters, int64_t step, int C, int task, int64_t slots_max, int64_t chunks_max, int64_t first_chunks)
{
    std::vector<ChunkDesc> ch;
    std::vector<ChunkBatch> bt;
    plan_chunk_list(iters, step, C, task, slots_max, chunks_max, first_chunks, ch, bt);
    // the chunks tile 1..iters in order, none longer than C, none crossing a sample point; the sample flag marks the chunks that start at one
    int64_t cur = 1, samples = 0;
    for (const ChunkDesc& c : ch) {
        REQUIRE((int64_t)c.g0 == cur);
        REQUIRE(c.count >= 1 && (int64_t)c.count <= C);
        const int64_t last = cur + c.count - 1;
        for (int64_t it = cur + 1; it <= last; ++it) REQUIRE(it % step != 0);          // a sample point is always a chunk's FIRST iteration
        REQUIRE(((c.flags & kChunkSampleBefore) != 0) == (cur % step == 0));
        samples += (c.flags & kChunkSampleBefore) ? 1 : 0;
        cur = last + 1;
    }
    REQUIRE(cur == iters + 1);
    REQUIRE(samples == iters / step);
    // the batches tile the chunk list; inside a batch slot_base is the running sum of counts and stays inside the plan buffers
    size_t at = 0;
    int64_t smp = 0;
    for (size_t b = 0; b < bt.size(); ++b) {
        REQUIRE(bt[b].first == at && bt[b].n >= 1);
        REQUIRE(bt[b].sample0 == smp);
        REQUIRE((int64_t)bt[b].n <= ((b == 0 && first_chunks > 0) ? first_chunks : chunks_max));
        int64_t slots = 0;
        for (size_t c = at; c < at + bt[b].n; ++c) {
            REQUIRE(ch[c].slot_base == (uint32_t)slots);
            slots += ch[c].count;
            smp += (ch[c].flags & kChunkSampleBefore) ? 1 : 0;
        }
        REQUIRE(slots <= slots_max || bt[b].n == 1);
        at += bt[b].n;
    }
    REQUIRE(at == ch.size());
}

int main()
{
    // ---- the chunk planner over call shapes: the bench's, the tests', ragged and degenerate ones
    const int64_t fixed[][3] = {{1 << 22, 1 << 12, 1408}, {4096, 4096, 1472}, {1, 1, 64}, {5, 7, 64}, {64, 1, 64}, {100000, 3, 832}, {8000, 250, 1472},
                                {1 << 16, 1 << 16, 4096}, {12345, 12346, 1472}, {4097, 4096, 1408}};
    for (const auto& f : fixed) {
        check_plan(f[0], f[1], (int)f[2], 64, 1 << 20, 1 << 16, 0);
        check_plan(f[0], f[1], (int)f[2], 64, 4 * f[2], 7, 2);
    }
    check_plan(0, 1, 64, 64, 1 << 20, 1 << 16, 0);                         // an empty call has no chunks
    uint64_t s = 20261003;
    for (int t = 0; t < 3000; ++t) {
        const int64_t iters = 1 + (int64_t)(lcg(s) % 20000), step = 1 + (int64_t)(lcg(s) % (t % 3 ? 3000 : 40));
        const int C = 64 * (1 + (int)(lcg(s) % 64));
        check_plan(iters, step, C, 64, C * (1 + (int64_t)(lcg(s) % 9)), 1 + (int64_t)(lcg(s) % 50), (int64_t)(lcg(s) % 4));
    }
    // ---- acceptance thresholds: u < ceil(p 2^64)  <=>  u 2^-64 < p, checked at the rounding edges
    bool always = false;
    REQUIRE(threshold64(0.0, &always) == 0 && !always);
    REQUIRE(threshold64(-1.0, &always) == 0 && !always);
    REQUIRE(threshold64(std::nan(""), &always) == 0 && !always);
    REQUIRE(threshold64(1.0, &always) == ~0ull && always);
    REQUIRE(threshold64(0.5, &always) == 1ull << 63 && !always);
    REQUIRE(threshold64(0x1.0p-64, &always) == 1);
    REQUIRE(threshold64(0x1.8p-64, &always) == 2);                         // 1.5 -> ceil = 2
    REQUIRE(threshold64(0x1.0p-1074, &always) == 1);                       // the smallest subnormal still accepts u = 0
    REQUIRE(threshold64(0x1.fffffffffffffp-1, &always) == 0xfffffffffffff800ull);
    for (int t = 0; t < 100000; ++t) {
        const double p = std::ldexp((double)(lcg(s) | 1), -53 - (int)(lcg(s) % 40));
        const uint64_t T = threshold64(p, &always);
        if (p >= 1.0) { REQUIRE(always); continue; }
        // T - 1 < p 2^64 <= T, evaluated exactly in long double (64-bit mantissa covers the cases drawn here)
        const long double x = (long double)p * 18446744073709551616.0L;
        REQUIRE((long double)T >= x && (T == 0 || (long double)(T - 1) < x));
    }
    // ---- shards of a multi-device context: whole 32-replica groups, in order, covering 0..R exactly
    for (int64_t R = 1; R <= 700; R += (R < 70 ? 1 : 37))
        for (int32_t nd = 1; nd <= 9; ++nd) {
            int64_t at = 0;
            for (int32_t d = 0; d < nd; ++d) {
                int64_t b0, b1;
                shard_bounds(R, nd, d, &b0, &b1);
                if (b1 <= b0) continue;
                REQUIRE(b0 == at && b0 % 32 == 0 && b1 <= R && (b1 % 32 == 0 || b1 == R));
                at = b1;
            }
            REQUIRE(at == R);
        }
    // ---- Philox4x32-10: the published known-answer vectors (Random123 kat_vectors), and the stream helpers' ranges
    {
        Philox4 o = philox4x32_10(0, 0, 0, 0, 0, 0);
        REQUIRE(o.w[0] == 0x6627e8d5u && o.w[1] == 0xe169c58du && o.w[2] == 0xbc57ac4cu && o.w[3] == 0x9b00dbd8u);
        o = philox4x32_10(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
        REQUIRE(o.w[0] == 0x408f276du && o.w[1] == 0x41c83b0eu && o.w[2] == 0xa20bc7c6u && o.w[3] == 0x6d5451fdu);
        o = philox4x32_10(0x243f6a88u, 0x85a308d3u, 0x13198a2eu, 0x03707344u, 0xa4093822u, 0x299f31d0u);
        REQUIRE(o.w[0] == 0xd16cfe09u && o.w[1] == 0x94fdccebu && o.w[2] == 0x5001e420u && o.w[3] == 0x24126ea1u);
        for (uint64_t g = 1; g < 5000; ++g)
            for (uint32_t N : {1u, 2u, 3u, 10u, 4096u, 262144u, 1048576u}) REQUIRE(site_of(7, 9, g, N) < N);
    }
    std::printf("host_sanitize: all invariants hold\n");
    return 0;
}
