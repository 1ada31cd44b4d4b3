// Is v_mfma_f32_32x32x8_bf16_1k half the cost of v_mfma_f32_32x32x16_bf16 on gfx950?  (A tail K chunk with ONE real channel group — the
// latent-input configs, DESIGN 8 — could then run at half price.)   hipcc --offload-arch=gfx950 -O3 mfma_k8_vs_k16.hip -o /tmp/k8 && /tmp/k8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int MODE>
__global__ void k(float* out, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    bf16x8 a8, b8;
    s16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(threadIdx.x * 0.002f - i); }
    for (int i = 0; i < 4; ++i) { a4[i] = (short)(threadIdx.x * 37 + i); b4[i] = (short)(threadIdx.x * 91 - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = 1024.0 * 4 * iters * 4;     // MFMAs
        printf("%s: %.2f ms, %.2f ns per MFMA per wave-slot, %.1f TFLOP/s\n", mode == 0 ? "32x32x16_bf16" : "32x32x8_bf16_1k", ms, ms * 1e6 / (iters * 4.0 * 4),
               n * 2.0 * 32 * 32 * (mode == 0 ? 16 : 8) / ms / 1e9);
    }
    return 0;
}
