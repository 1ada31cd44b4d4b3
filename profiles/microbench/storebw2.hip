// microbench: streaming HBM write bandwidth (footprint >> MALL) for 8-B-interleaved vs 16-B dense stores, + read
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PAT>
__global__ void k(uint4* out, long long nvec) {
    const long long gw = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // global wave id
    const int lane = threadIdx.x & 63;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long base = gw * 64; base + 64 <= nvec; base += nwaves * 64) {
        if (PAT == 0) {
            uint2 v = make_uint2(lane, (unsigned)base);
            ((uint2*)(out + base + (lane & 31)))[lane >> 5] = v;
            ((uint2*)(out + base + 32 + (lane & 31)))[lane >> 5] = v;
        } else if (PAT == 1) {
            out[base + lane] = make_uint4(lane, (unsigned)base, 1, 2);
        } else {
            uint4 v = out[base + lane];
            if (v.x == 0xdeadbeef) out[0] = v;
        }
    }
}
int main() {
    const long long nvec = (4LL << 30) / 16;
    uint4* d; hipMalloc(&d, nvec * 16); hipMemset(d, 0, nvec * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 3; ++pat) for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, d, nvec);
        if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, d, nvec);
        if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, d, nvec);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("pattern %d (%s): %.2f TB/s\n", pat, pat == 0 ? "8B interleaved stores" : pat == 1 ? "16B dense stores" : "16B reads", nvec * 16.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
