// How much clock (power budget) do the operand feeds cost?  Every wave runs 9-MFMA groups (32x32x16 bf16, random operands);
// per group it additionally issues NLDS ds_read_b128 (operands really come from LDS when NLDS > 0) and NDMA 1-KiB
// global_load_lds copies from an L2-resident buffer.  Reports sustained TFLOP/s and the shader clock.
//   hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power && ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void glds16(const uint4* src, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_dst) : "memory");
}

template <int NLDS, int NDMA, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, unsigned long long* cyc, const uint4* src, int iters) {
    __shared__ uint4 lds[4096];   // 64 KiB
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (h & 0x83ff83ffu) | 0x3c003c00u; };
    for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = make_uint4(nx(), nx(), nx(), nx());
    __syncthreads();
    f32x16 acc[3];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    uint4 a0 = make_uint4(nx(), nx(), nx(), nx()), a1 = make_uint4(nx(), nx(), nx(), nx());
    uint4 b[3][2];
    for (int j = 0; j < 3; ++j) { b[j][0] = make_uint4(nx(), nx(), nx(), nx()); b[j][1] = make_uint4(nx(), nx(), nx(), nx()); }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int base = ((it * 8 + wave * 67) & 31) * 64 + lane;
        if (NLDS >= 2) { a0 = lds[base]; a1 = lds[base + 2048]; }
        if (NLDS >= 4) { b[0][0] = lds[base + 64]; b[0][1] = lds[base + 2048 + 64]; }
        if (NLDS >= 6) { b[1][0] = lds[base + 128]; b[1][1] = lds[base + 2048 + 128]; }
        if (NLDS >= 8) { b[2][0] = lds[base + 192]; b[2][1] = lds[base + 2048 + 192]; }
#pragma unroll
        for (int d = 0; d < NDMA; ++d) glds16(src + ((it * NDMA + d) & 1023) * 64 + lane, lds0 + 32768 + ((wave * NDMA + d) & 31) * 1024);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b[j][0]), acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b[j][1]), acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b[j][0]), acc[j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NLDS, int NDMA, int WAVES>
void run(int wgs_per_cu, int iters) {
    const int wgs = 256 * wgs_per_cu;
    float* out; unsigned long long* cyc; uint4* src;
    (void)hipMalloc(&out, wgs * WAVES * 64 * 4); (void)hipMalloc(&cyc, wgs * 8); (void)hipMalloc(&src, 1024 * 1024);
    (void)hipMemset(src, 0x3c, 1024 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 0, ms = 0;
    unsigned long long c = 0;
    for (int rep = 0; rep < 4; ++rep) {     // ~0.2 s per rep: long enough for the power management to settle
        (void)hipEventRecord(e0);
        k<NLDS, NDMA, WAVES><<<wgs, WAVES * 64>>>(out, cyc, src, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        best = ms;
    }
    const double flop = (double)wgs * WAVES * iters * 9 * 32768.0;
    printf("lds_reads/9mfma %d  dma/9mfma %d  waves/WG %d WG/CU %d: %.1f ms  %.0f TFLOP/s  wave-cycles/iter %.0f  clock(if 1 wave/SIMD) %.2f GHz\n", NLDS, NDMA,
           WAVES, wgs_per_cu, best, flop / best / 1e9, (double)c / iters, c / (best * 1e6));
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(src);
}

int main() {
    const int N = 1200000;
    run<0, 0, 4>(1, N);
    run<2, 0, 4>(1, N);
    run<4, 0, 4>(1, N);
    run<8, 0, 4>(1, N);
    run<8, 1, 4>(1, N);
    run<8, 2, 4>(1, N);
    run<0, 2, 4>(1, N);
    run<8, 0, 4>(2, N / 2);
    run<8, 2, 4>(2, N / 2);
    run<8, 2, 4>(3, N / 3);
    return 0;
}
