// How should a LONE 4-wave workgroup per CU (the 52x52 training crops: one tile per CU) bring its K-chunk into LDS while it multiplies the
// previous one?  Per chunk every wave moves NOPS x 1 KiB from a MALL-resident buffer into LDS and issues NMFMA 32x32x16 bf16 MFMAs whose
// operands are read back from LDS at the conv kernel's rate (8 ds_read_b128 per 9 MFMAs).
//   MODE 0  LDS-DMA (global_load_lds_dwordx4), all copies of chunk c+1 issued in front of the MFMAs of chunk c  — esr_conv.hip NST = 2
//   MODE 1  register staging: global_load_dwordx4 of chunk c+1 issued in front of the MFMAs of chunk c, ds_write_b128 after them
//   MODE 2  register staging, the loads issued one by one between the MFMA groups
//   MODE 3  LDS-DMA issued one by one between the MFMA groups
// Reports shader cycles per chunk (wave 0 of every workgroup, mean) and wall time.
//   hipcc --offload-arch=gfx950 -O3 ingest_paths.hip -o bin/ingest_paths && bin/ingest_paths
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void glds16(const uint4* src, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void gload16(u32x4& dst, const uint4* src) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory"); }
__device__ __forceinline__ void pin(u32x4& f) { asm volatile("" : "+v"(f)); }

__global__ void fill(uint4* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + 12345u;
    auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (h & 0x83ff83ffu) | 0x3c003c00u; };
    p[i] = make_uint4(nx(), nx(), nx(), nx());
}

constexpr int STAGE_VEC = 20 * 4 * 64;      // up to 20 ops x 4 waves x 64 lanes (16 B each) per stage = 80 KiB

template <int MODE, int NOPS, int NMFMA>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, const uint4* src, size_t region_vec, int iters) {
    extern __shared__ uint4 lds[];           // 2 stages
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint4* const base = src + (size_t)blockIdx.x * region_vec;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds;
    for (int i = threadIdx.x; i < 2 * STAGE_VEC; i += 256) lds[i] = base[i % region_vec];
    __syncthreads();
    f32x16 acc[3];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    u32x4 r[(MODE == 1 || MODE == 2) ? NOPS : 1];
    constexpr int NG = NMFMA / 9;            // groups of 9 MFMAs (one tap of the split MT1 kernel: 3 column tiles x 3 terms)
    size_t pos = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int st = it & 1;
        const uint4* const sp = base + pos + wave * 64 + lane;           // this wave's pieces: sp + op * 256
        pos += (size_t)NOPS * 256;
        if (pos + (size_t)NOPS * 256 > region_vec) pos = 0;
        if (MODE == 0) {
#pragma unroll
            for (int op = 0; op < NOPS; ++op) glds16(sp + op * 256, lds0 + ((st ^ 1) * STAGE_VEC + (op * 4 + wave) * 64) * 16);
        }
        if (MODE == 1) {
#pragma unroll
            for (int op = 0; op < NOPS; ++op) gload16(r[op], sp + op * 256);
        }
        const uint4* const s = lds + st * STAGE_VEC + lane;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int o = ((g * 5 + wave) % (NOPS * 4 - 8)) * 64;
            const uint4 a0 = s[o], a1 = s[o + 64], b00 = s[o + 128], b01 = s[o + 192], b10 = s[o + 256], b11 = s[o + 320], b20 = s[o + 384], b21 = s[o + 448];
            const uint4 bb[3][2] = {{b00, b01}, {b10, b11}, {b20, b21}};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bb[j][0]), acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bb[j][1]), acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bb[j][0]), acc[j], 0, 0, 0);
            }
            if (MODE == 2) {
#pragma unroll
                for (int op = 0; op < NOPS; ++op)
                    if (op * NG / NOPS == g) gload16(r[op], sp + op * 256);
            }
            if (MODE == 3) {
#pragma unroll
                for (int op = 0; op < NOPS; ++op)
                    if (op * NG / NOPS == g) glds16(sp + op * 256, lds0 + ((st ^ 1) * STAGE_VEC + (op * 4 + wave) * 64) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int op = 0; op < NOPS; ++op) {
                pin(r[op]);
                ((u32x4*)lds)[(st ^ 1) * STAGE_VEC + (op * 4 + wave) * 64 + lane] = r[op];
            }
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 16; ++i) sum += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NOPS, int NMFMA>
void run(const uint4* src, size_t total_vec, int iters) {
    const int wgs = 256;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, wgs * 256 * 4); (void)hipMalloc(&cyc, wgs * 8);
    auto kern = k<MODE, NOPS, NMFMA>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_VEC * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    unsigned long long c[256];
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        kern<<<wgs, 256, 2 * STAGE_VEC * 16>>>(out, cyc, src, total_vec / wgs, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 256; ++i) mean += (double)c[i] / 256;
    static const char* names[] = {"lds-dma, up front", "registers, up front", "registers, interleaved", "lds-dma, interleaved"};
    printf("%-24s  %2d KiB/wave %3d MFMA/wave per chunk: %6.0f cycles/chunk (MFMA floor %4d)  %5.1f B/cycle/CU  %.2f us/chunk  clock %.2f GHz\n", names[MODE], NOPS,
           NMFMA, mean / iters, NMFMA * 32, NOPS * 4096.0 / (mean / iters), ms * 1e3 / iters, mean / iters / (ms * 1e3 / iters) / 1e3);
    (void)hipFree(out); (void)hipFree(cyc);
}

template <int NOPS, int NMFMA>
void all(const uint4* src, size_t total_vec, int iters) {
    run<0, NOPS, NMFMA>(src, total_vec, iters);
    run<3, NOPS, NMFMA>(src, total_vec, iters);
    run<1, NOPS, NMFMA>(src, total_vec, iters);
    run<2, NOPS, NMFMA>(src, total_vec, iters);
}

int main() {
    const size_t total_vec = (size_t)48 * 1024 * 1024 / 16;      // 48 MB: what a 128-channel split input of the 32 x 52x52 batch occupies
    uint4* src;
    (void)hipMalloc(&src, total_vec * 16);
    fill<<<(unsigned)((total_vec + 255) / 256), 256>>>(src, total_vec);
    (void)hipDeviceSynchronize();
    const int iters = 20000;
    printf("-- split MT1 (81 MFMAs per chunk): 17 KiB per wave as issued today, 13 without the redundant slots\n");
    all<17, 81>(src, total_vec, iters);
    all<13, 81>(src, total_vec, iters);
    printf("-- bf16 MT1 (27 MFMAs per chunk): 9 KiB today, 7 without the redundant slots\n");
    all<9, 27>(src, total_vec, iters);
    all<7, 27>(src, total_vec, iters);
    printf("-- copies only\n");
    run<0, 13, 0>(src, total_vec, iters);
    run<1, 13, 0>(src, total_vec, iters);
    return 0;
}
