// microbench: HBM write bandwidth of the conv epilogue's store pattern vs dense 16-byte stores
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PAT>
__global__ void k(uint4* out, long long plane_vec, int runs_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < runs_per_wg; ++r) {
        long long base = ((long long)blockIdx.x * 4 + wave) * 64 + (long long)(r % 8) * plane_vec + (long long)(r / 8) * 150;
        if (PAT == 0) {
            uint2 v = make_uint2(lane, r);
            ((uint2*)(out + base + (lane & 31)))[lane >> 5] = v;
            ((uint2*)(out + base + 32 + (lane & 31)))[lane >> 5] = v;
        } else if (PAT == 1) {
            out[base + lane] = make_uint4(lane, r, 1, 2);
        } else {
            out[base + (lane & 31) + (lane >> 5) * plane_vec * 8] = make_uint4(lane, r, 1, 2);
            out[base + 32 + (lane & 31) + (lane >> 5) * plane_vec * 8] = make_uint4(lane, r, 1, 2);
        }
    }
}
int main() {
    uint4* d; size_t bytes = (size_t)8 << 30; hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long long plane_vec = 150 * 150 * 32;
    for (int pat = 0; pat < 3; ++pat) for (int rep = 0; rep < 2; ++rep) {
        const int wgs = 1920, runs = 40;
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) {
            if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, d, plane_vec, runs);
            if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, d, plane_vec, runs);
            if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, d, plane_vec, runs);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double b = 20.0 * wgs * 4 * runs * 1024 * (pat == 2 ? 2 : 1);
        printf("pattern %d: %.2f TB/s (%.1f us per launch, %.0f MB) %s\n", pat, b / (ms * 1e-3) / 1e12, ms * 1e3 / 20, b / 20 / 1e6, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
