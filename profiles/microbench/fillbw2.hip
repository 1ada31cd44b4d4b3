// microbench: LDS-DMA streaming from HBM with the conv's access shape: per step a WG fetches `nrow` runs of 1216 B
// (76 px) laid out with a given stride; 1 WG per CU, steps walk forward through a large buffer (no reuse).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void glds16(const uint4* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
__global__ void k(const uint4* src, long long nvec, int nrow, long long row_stride_vec, long long plane_stride_vec, int nplane, int steps, int depth, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // per step: nplane planes x nrow rows x 76 px; px index p in [0, nrow*76) per plane -> 64-px slots
    const int npx = nrow * 76, nslot = (npx + 63) / 64;
    unsigned acc = 0;
    for (int s = 0; s < steps; ++s) {
        // tile base walks through the buffer: different WGs + steps touch different rows
        long long tb = ((long long)(blockIdx.x + (long long)s * gridDim.x) * 5 * row_stride_vec) % (plane_stride_vec - (long long)(nrow + 1) * row_stride_vec);
        for (int pl = 0; pl < nplane; ++pl)
            for (int sl = wave; sl < nslot; sl += 4) {
                int p = sl * 64 + lane;
                if (p < npx) {
                    int r = p / 76, c = p - r * 76;
                    glds16(src + pl * plane_stride_vec + tb + r * row_stride_vec + c, lds0 + ((s % depth) * nplane * nslot + pl * nslot + sl) * 1024);
                }
            }
        if (s + 1 >= depth) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); acc += smem[threadIdx.x * 16]; __syncthreads(); }
    }
    if (acc == 0x12345678) out[0] = acc;
}
int main() {
    const long long plane_vec = 150LL * 150 * 32;           // one channel group plane for batch 32 (vectors)
    const int NPLANES = 48 * 2;
    uint4* d; float* o;
    hipMalloc(&d, plane_vec * NPLANES * 16); hipMalloc(&o, 4);
    hipMemset(d, 1, plane_vec * NPLANES * 16);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { const char* name; long long rs, ps; int depth; } cfgs[] = {
        {"conv-like: rows 2400B apart, 4 planes 11.5MB apart, depth1", 150, plane_vec, 1},
        {"conv-like depth2", 150, plane_vec, 2},
        {"rows contiguous (1216B), planes 11.5MB apart, depth1", 76, plane_vec, 1},
        {"rows contiguous, planes adjacent (fully contiguous 34KB), depth1", 76, 76 * 7, 1},
        {"fully contiguous, depth2", 76, 76 * 7, 2},
    };
    for (auto& c : cfgs) {
        const int nrow = 7, nplane = 4, steps = 400;
        size_t lds = (size_t)c.depth * nplane * ((nrow * 76 + 63) / 64) * 1024;
        hipLaunchKernelGGL(k, dim3(256), dim3(256), lds, 0, d, plane_vec * NPLANES, nrow, c.rs, c.ps == plane_vec ? plane_vec : (long long)c.ps, nplane, 20, c.depth, o);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), lds, 0, d, plane_vec * NPLANES, nrow, c.rs, c.ps == plane_vec ? plane_vec : (long long)c.ps, nplane, steps, c.depth, o);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = 256.0 * steps * nplane * nrow * 76 * 16;
        printf("%-70s: %.2f us/step, %.1f GB/s per CU, %.2f TB/s chip  (%s)\n", c.name, ms * 1e3 / steps, bytes / (ms * 1e-3) / 256 / 1e9, bytes / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
