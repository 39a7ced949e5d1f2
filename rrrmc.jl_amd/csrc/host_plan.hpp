// Host-side planning of a sampling call, free of any HIP dependency: the same code is compiled into the library (rrrmc_hip.hip) and,
// with g++ -fsanitize=address,undefined, into tests/host_sanitize.cpp (sanitizers on the CPU build only).
//   ChunkDesc / plan_chunk_list   how the iterations of a standardMC call are cut into chunks and batches
//   threshold64                   ceil(p 2^64): the acceptance thresholds of the bit-plane comparison
//   shard_bounds                  the replica shards of a multi-device context
#pragma once
#include <stdint.h>

#include <cstring>
#include <vector>

namespace rrrmc {

constexpr uint32_t kChunkSampleBefore = 1u;       // chunk flag: an energy sample is due before its first move

struct ChunkDesc {
    uint64_t g0;         // iteration (1-based, RELATIVE to the sampling call: the kernels add the call's base `gbase` = iterations done before it)
                         // of the chunk's first attempt — the list depends on (iters, step, C) only, so back-to-back calls share one upload
    uint32_t count;      // attempts in the chunk (<= C)
    uint32_t slot_base;  // offset of the chunk's slots / vector table in the plan buffers
    uint32_t nvec;       // number of consumer batches, written by plan_kernel
    uint32_t flags;
};

struct ChunkBatch { size_t first, n; int64_t sample0; };      // chunks [first, first + n) are planned / swept together; sample0 = samples before them

// Chunk list of one call: cuts at every multiple of `step` (a sample precedes the move of iteration k*step, src/RRRMC.jl:104-108) and
// every C moves.  Every step of the sweep kernel costs about a microsecond whatever its chunk holds, so the iterations between two cuts
// (sample points, the end of the call) are divided into the FEWEST chunks of at most C, made equally long in whole `task`-slot producer
// tasks where the segment allows (4096 = 1408 + 1344 + 1344 is 64 tasks of 64, three times 1366 would be 66).
// Batches are bounded by the plan buffers (batch_slots_max slots, batch_chunks_max chunks; a shorter first batch where asked for);
// slot_base restarts in every batch.
inline void plan_chunk_list(int64_t iters, int64_t step, int C, int task, int64_t batch_slots_max, int64_t batch_chunks_max,
                            int64_t batch_first_chunks, std::vector<ChunkDesc>& chunks, std::vector<ChunkBatch>& batches)
{
    const int64_t nsamp = iters / step;
    chunks.clear();
    batches.clear();
    chunks.reserve((size_t)(iters / C + nsamp + 2));
    for (int64_t cur = 1; cur <= iters;) {
        const int64_t next_sample = (cur / step + 1) * step;
        int64_t seg_end = next_sample;                        // exclusive end of the segment that may be chunked freely
        if (seg_end > iters + 1) seg_end = iters + 1;
        const int64_t seg = seg_end - cur, nch = (seg + C - 1) / C;
        int64_t len = (seg + nch - 1) / nch;                  // ceil(seg / nch) <= C: the remaining chunks re-balance themselves
        if (nch > 1) {
            const int64_t up = (len + task - 1) / task * task, down = len / task * task;
            if (up <= C) len = up;
            else if (down > 0 && seg - down <= (nch - 1) * (int64_t)C) len = down;
        }
        int64_t end = cur + len;
        if (end > seg_end) end = seg_end;
        ChunkDesc cd{};
        cd.g0 = (uint64_t)cur;
        cd.count = (uint32_t)(end - cur);
        cd.flags = (cur % step == 0) ? kChunkSampleBefore : 0u;
        chunks.push_back(cd);
        cur = end;
    }
    const size_t nch_all = chunks.size();
    size_t first = 0;
    int64_t slots = 0, samples = 0, sample0 = 0;
    for (size_t c = 0; c < nch_all; ++c) {
        // (a short first batch where the plan of a batch is expensive: the sweep starts after plan(0), every later plan overlaps a sweep)
        const int64_t cmax = (first == 0 && batch_first_chunks > 0) ? batch_first_chunks : batch_chunks_max;
        if (c > first && (slots + chunks[c].count > batch_slots_max || (int64_t)(c - first) >= cmax)) {
            batches.push_back({first, c - first, sample0});
            first = c; slots = 0; sample0 = samples;
        }
        chunks[c].slot_base = (uint32_t)slots;
        slots += chunks[c].count;
        if (chunks[c].flags & kChunkSampleBefore) samples += 1;
    }
    if (nch_all > first) batches.push_back({first, nch_all - first, sample0});
}

// ceil(p * 2^64) for 0 < p < 1; *always when p >= 1.  accept(x) = x >= 0 || rand() < exp(x), src/RRRMC.jl:39: with a 64-bit uniform u,
// u 2^-64 < p  <=>  u < ceil(p 2^64)
inline uint64_t threshold64(double p, bool* always)
{
    *always = false;
    if (!(p > 0.0)) return 0;
    if (p >= 1.0) { *always = true; return ~0ull; }
    uint64_t bits;
    std::memcpy(&bits, &p, 8);
    const int bexp = (int)((bits >> 52) & 0x7ff);
    uint64_t man = bits & ((1ull << 52) - 1);
    int e;
    if (bexp == 0) e = -1074; else { man |= 1ull << 52; e = bexp - 1075; }
    const int sh = e + 64;
    if (sh >= 0) return man << sh;
    const int s = -sh;
    if (s >= 64) return 1;
    return (man + ((1ull << s) - 1)) >> s;
}

// shard d of ndev over R replicas: whole 32-replica groups in global-id order (the last shard takes the ragged tail); b1 <= b0 = empty
inline void shard_bounds(int64_t R, int32_t ndev, int32_t d, int64_t* b0, int64_t* b1)
{
    const int64_t groups = (R + 31) / 32;
    *b0 = 32 * (groups * d / ndev);
    *b1 = d + 1 == ndev ? R : 32 * (groups * (d + 1) / ndev);
    if (*b1 > R) *b1 = R;
    if (*b0 > R) *b0 = R;
}

}  // namespace rrrmc
