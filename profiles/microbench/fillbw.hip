// microbench: per-CU ingest bandwidth of L2-resident data via LDS-DMA vs plain VGPR loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void glds16(const uint4* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
// each workgroup re-reads its own `kb` KiB region `iters` times
template <int MODE>
__global__ void k(const uint4* src, int kb, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const uint4* base = src + (size_t)blockIdx.x * kb * 64;   // kb KiB = kb*64 uint4
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            for (int j = wave; j < kb; j += nw) glds16(base + j * 64 + lane, smem + j * 1024);
            __syncthreads();
            acc.x += ((uint4*)smem)[threadIdx.x].x;
            __syncthreads();
        } else {
            for (int j = wave; j < kb; j += nw) { uint4 v = base[j * 64 + lane]; acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w; }
        }
    }
    if (acc.x == 0x12345 && acc.y == 7) out[0] = acc.z + acc.w;
}
int main() {
    const int NCU = 256;
    uint4* d; float* o;
    hipMalloc(&d, (size_t)4096 * 64 * 1024); hipMalloc(&o, 4);
    hipMemset(d, 1, (size_t)4096 * 64 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int wgs : {256, 512, 768})
            for (int threads : {256, 512})
                for (int kb : {32, 64}) {
                    if (mode == 0 && (size_t)kb * 1024 * (wgs / 256) > 160 * 1024) continue;
                    const int iters = 200;
                    auto kern = mode == 0 ? k<0> : k<1>;
                    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    size_t lds = mode == 0 ? (size_t)kb * 1024 : 0;
                    hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), lds, 0, d, kb, 5, o);
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), lds, 0, d, kb, iters, o);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    double bytes = (double)wgs * kb * 1024 * iters;
                    printf("%s wgs %d thr %d kb %d: %.1f GB/s per CU, %.2f TB/s chip (%s)\n", mode == 0 ? "lds-dma" : "vgpr   ", wgs, threads, kb,
                           bytes / (ms * 1e-3) / NCU / 1e9, bytes / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
                }
    return 0;
}
