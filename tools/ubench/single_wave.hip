// Micro-benchmark: how fast can ONE wavefront issue?  Cycles per wave64 instruction for a single wave on its SIMD
// (independent / dependent VALU streams, LDS reads and writes), measured with s_memtime around long unrolled sequences.
// The sweep kernel's consumer is a single wave: this is its ceiling.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP> __global__ __launch_bounds__(64) void k(unsigned long long* out, uint32_t* sink, uint32_t seed)
{
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i * 7 + seed;
    __syncthreads();
    uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 * 3 + 1;
    uint32_t b0 = a0 >> 3;
    uint32_t addr = (threadIdx.x * 4u) & 0x3ffcu;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (OP == 0) { REP64(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
    if (OP == 1) { REP64(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(b0));) }
    if (OP == 2) { REP64(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(addr));) }
    if (OP == 3) { REP64(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_and_b32 %1, 0x3ffc, %0" : "=v"(a0), "+v"(addr));) }
    if (OP == 4) {
        const uint32_t wa = (threadIdx.x * 16u) & 0x3ff0u;
        REP64(asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:8\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:1032" :: "v"(wa), "v"(t0));)
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    if (OP == 5) {
        uint4 q;
        REP64(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr)); addr = q.x & 0x3ff0u;)
        a0 ^= q.y;
    }
    if (OP == 6) {
        const uint32_t wa = (threadIdx.x * 16u) & 0x3ff0u;
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u q = {a0, a1, a2, a3};
        REP64(asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072" :: "v"(wa), "v"(q));)
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    if (OP == 7) {
        const uint32_t wa = (threadIdx.x * 16u) & 0x3ff0u;
        REP64(asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %1 offset:2048\n ds_write_b32 %0, %1 offset:3072" :: "v"(wa), "v"(a0));)
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    if (OP == 8) {   // ds_read_b32 issue rate: 4 reads in flight, consumed late
        REP64(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(addr));)
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    if (OP == 9) {   // ds_read_b128 issue rate
        uint4 q0, q1;
        const uint32_t ra = (threadIdx.x * 16u) & 0x3ff0u;
        REP64(asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n ds_read_b128 %0, %2 offset:2048\n ds_read_b128 %1, %2 offset:3072" : "=v"(q0), "=v"(q1) : "v"(ra));)
        asm volatile("s_waitcnt lgkmcnt(0)");
        a0 ^= q0.x ^ q1.y;
    }
    if (OP == 10) {  // random-address gathers (bank conflicts as in the consumer)
        uint32_t r0 = (a0 >> 7) & 0x3ffcu, r1 = (a1 >> 9) & 0x3ffcu, r2 = (a2 >> 5) & 0x3ffcu, r3 = (a3 >> 11) & 0x3ffcu;
        REP64(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));)
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ addr;
}
template <int OP> void run(const char* name, int ninstr, unsigned long long* d, uint32_t* s)
{
    k<OP><<<256, 64>>>(d, s, 1);
    hipDeviceSynchronize();
    k<OP><<<256, 64>>>(d, s, 2);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < 256; ++i) sum += (double)h[i];
    printf("%-44s %8.2f counter ticks per instruction (%d instr)\n", name, sum / 256 / ninstr, ninstr);
}
int main()
{
    unsigned long long* d; uint32_t* s;
    hipMalloc(&d, 256 * 8); hipMalloc(&s, 256 * 64 * 4);
    run<0>("4 independent v_xor streams", 256, d, s);
    run<1>("1 dependent v_xor stream", 256, d, s);
    run<2>("4 ds_read_b32 + wait (per group of 4)", 64, d, s);
    run<3>("dependent ds_read_b32 chain (latency)", 64, d, s);
    run<4>("ds_write_b64 stream (per instruction)", 256, d, s);
    run<5>("dependent ds_read_b128 chain (latency)", 64, d, s);
    run<6>("ds_write_b128 stream (per instruction)", 256, d, s);
    run<7>("ds_write_b32 stream (per instruction)", 256, d, s);
    run<8>("ds_read_b32 stream, linear addresses", 256, d, s);
    run<9>("ds_read_b128 stream", 256, d, s);
    run<10>("ds_read_b32 stream, random addresses", 256, d, s);
    // the cycle counter's unit
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("(ticks of __builtin_readcyclecounter = s_memtime; on this part they are shader clocks: LDS latency reads 64)\n");
    return 0;
}
