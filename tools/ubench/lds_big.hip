// Does a workgroup really get all the dynamic LDS it asks for above 128 KiB?  Every dword of the allocation is written with its own index by
// 1024 threads and read back by the NEXT wavefront after a barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned* out, int ndw)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned* w = reinterpret_cast<unsigned*>(lds);
    for (int i = threadIdx.x; i < ndw; i += 1024) w[i] = 0x5a000000u + (unsigned)i;
    __syncthreads();
    unsigned bad = 0, firstbad = 0xffffffffu;
    for (int i = (threadIdx.x + 64) % 1024; i < ndw; i += 1024) if (w[i] != 0x5a000000u + (unsigned)i) { ++bad; if (firstbad == 0xffffffffu) firstbad = (unsigned)i; }
    atomicAdd(out, bad);
    if (bad) atomicMin(out + 1, firstbad);
}
int main()
{
    unsigned* d; hipMalloc(&d, 8);
    for (int bytes : {65536, 98304, 131072, 138816, 147456, 161928, 163840}) {
        unsigned init[2] = {0, 0xffffffffu}; hipMemcpy(d, init, 8, hipMemcpyHostToDevice);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), bytes, 0, d, bytes / 4);
        hipError_t e2 = hipDeviceSynchronize();
        unsigned h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("%7d bytes: attr %s, run %s, bad dwords %u (first %u)\n", bytes, hipGetErrorString(e), hipGetErrorString(e2), h[0], h[1]);
    }
    return 0;
}
