// Sustained MFMA rate and shader clock under load: every wave issues `iters` x 8 independent 32x32x16 MFMAs from registers.
//   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, int rnd) {
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    uint4 av = make_uint4(threadIdx.x, 0x3f803f80, 0x3f803f80, 0x3f803f80), bv = make_uint4(0x3f803f80, threadIdx.x * 3, 0x3f803f80, 1);
    uint4 av2 = av, bv2 = bv;
    if (rnd) {   // random finite bf16/f16 bit patterns (sign/mantissa random, small exponents): realistic operand toggling
        unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
        auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (h & 0x83ff83ffu) | 0x3c003c00u; };
        av = make_uint4(nx(), nx(), nx(), nx()); bv = make_uint4(nx(), nx(), nx(), nx());
        av2 = make_uint4(nx(), nx(), nx(), nx()); bv2 = make_uint4(nx(), nx(), nx(), nx());
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint4 aa = (j & 1) ? av2 : av, bb = (j & 2) ? bv2 : bv;
            if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aa), __builtin_bit_cast(bf16x8, bb), acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aa), __builtin_bit_cast(f16x8, bb), acc[j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int wgs, int iters, int rnd = 0) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, wgs * 256 * 4); hipMalloc(&cyc, wgs * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<wgs, 256>>>(out, cyc, iters / 10, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<wgs, 256>>>(out, cyc, iters, rnd);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[4]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double flop = (double)wgs * 4 * iters * 8 * 32768.0;
    printf("%-5s rnd %d wgs %5d: %.2f ms  %.1f TFLOP/s  wave cycles %llu -> clock %.2f GHz, %.1f cyc/MFMA/SIMD-wave\n", name, rnd, wgs, ms, flop / ms / 1e9,
           c[0], c[0] / (ms * 1e6), (double)c[0] / (iters * 8));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int i = 0; i < 3; ++i) run<0>("bf16", 256, 2000000, 0);
    for (int i = 0; i < 8; ++i) run<0>("bf16", 256, 2000000, 1);   // random operands: power-limited clock?
    for (int i = 0; i < 4; ++i) run<1>("f16", 256, 2000000, 1);
    for (int i = 0; i < 3; ++i) run<0>("bf16", 512, 1000000, 1);
    return 0;
}
