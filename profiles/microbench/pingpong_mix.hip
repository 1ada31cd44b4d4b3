// Round 4, VERDICT r3 item 1: what can an 8-wave PING-PONG workgroup (two waves per SIMD in antiphase, shared weight fragments) reach against
// today's arrangement (two unsynchronised 4-wave workgroups per CU, one LDS stage each) on the instruction mix of the split (bf16x3) 32-channel
// conv kernel at configs[1]?  Per K chunk a wave issues 81 v_mfma_f32_32x32x16_bf16 (9 taps x 3 column tiles x 3 terms) whose operands come
// from LDS at the kernel's rate (8 ds_read_b128 per 9 MFMAs), and the workgroup copies the chunk's operands with 1-KiB global_load_lds pieces:
//   ARR 0  "2wg"       4 waves, 2 workgroups per CU (512 workgroups): issue NOPS pieces per wave -> vmcnt(0) -> barrier -> MFMAs -> barrier
//   ARR 1  "pingpong"  8 waves, 1 workgroup per CU: in phase p half (p & 1) multiplies chunk p / 2 of ITS tile out of its own activation stage and
//                      the shared weight stage, the other half copies its next activation chunk (9 pieces per wave) and half of the next weight
//                      chunk (2-3 pieces per wave); one barrier per phase
//   PRIO   0 none; 1 s_setprio 1 while a wave multiplies, 0 while it copies; 2 static s_setprio 1 for waves 4-7 (MI355X_MICROARCH.md)
//   BUF    0 global_load_lds_dwordx4 (64-bit lane addresses); 1 buffer_load_dwordx4 ... offen lds through one SRD (uniform base and bounds in
//          SGPRs, one 32-bit lane offset computed once, the piece's position as the SGPR offset: no per-lane address arithmetic per copy)
// Source: a region per workgroup walked cyclically — 48 MB in total (Infinity-Cache resident) or 2 GB (streamed from HBM).
//   hipcc --offload-arch=gfx950 -O3 pingpong_mix.hip -o bin/pingpong_mix && bin/pingpong_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void glds16(const uint4* src, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ i32x4 make_srd(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
    r.z = 0x7fffffff;
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void glds16b(i32x4 srd, unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 : : "v"(voff), "s"(srd), "s"(__builtin_amdgcn_readfirstlane(soff)), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

__global__ void fill(uint4* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + 12345u;
    auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (h & 0x83ff83ffu) | 0x3c003c00u; };
    p[i] = make_uint4(nx(), nx(), nx(), nx());
}

// 81 MFMAs out of `s` (this lane's pointer into an LDS stage of >= NV 16-byte vectors per 64-lane slot row)
__device__ __forceinline__ void mfma81(f32x16 (&acc)[3], const uint4* s, const uint4* w, int wave, int nslot) {
#pragma unroll
    for (int g = 0; g < 9; ++g) {
        const int o = ((g * 5 + wave) % (nslot - 6)) * 64;
        const uint4 a0 = w[(2 * g) * 64], a1 = w[(2 * g + 1) * 64];
        const uint4 b00 = s[o], b01 = s[o + 64], b10 = s[o + 128], b11 = s[o + 192], b20 = s[o + 256], b21 = s[o + 320];
        const uint4 bb[3][2] = {{b00, b01}, {b10, b11}, {b20, b21}};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bb[j][0]), acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bb[j][1]), acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bb[j][0]), acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

constexpr int ACT_SLOTS = 36, W_SLOTS = 18;              // 1-KiB pieces per chunk: 4 planes x 9 slots of activations, 18 weight fragments

template <int PRIO, int BUF = 0>
__global__ __launch_bounds__(256, 2) void k_2wg(float* out, unsigned long long* cyc, const uint4* src, size_t region_vec, int iters) {
    extern __shared__ uint4 lds[];           // one stage: ACT_SLOTS + W_SLOTS pieces
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint4* const base = src + (size_t)blockIdx.x * region_vec;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds;
    f32x16 acc[3];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    constexpr int NP = ACT_SLOTS + W_SLOTS;   // 54 pieces, 13-14 per wave
    size_t pos = 0;
    const i32x4 srd = make_srd(base);
    const unsigned vlane = lane * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const uint4* const sp = base + pos + lane;
        const unsigned spos = (unsigned)(pos * 16);
        pos += (size_t)NP * 64;
        if (pos + (size_t)NP * 64 > region_vec) pos = 0;
        if (BUF) { for (int p = wave; p < NP; p += 4) glds16b(srd, vlane, spos + p * 1024, lds0 + p * 1024); }
        else for (int p = wave; p < NP; p += 4) glds16(sp + p * 64, lds0 + p * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
        mfma81(acc, lds + lane, lds + ACT_SLOTS * 64 + lane, wave, ACT_SLOTS);
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 16; ++i) sum += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int PRIO, int WSHARE, int BUF = 0>
__global__ __launch_bounds__(512, 1) void k_pp(float* out, unsigned long long* cyc, const uint4* src, size_t region_vec, int iters) {
    extern __shared__ uint4 lds[];           // ACT[2] (36 pieces each), W[2] (18 pieces each)
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int half = wave_all >> 2, wave = wave_all & 3;
    const uint4* const base = src + (size_t)blockIdx.x * region_vec;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds;
    const unsigned act0 = lds0 + half * ACT_SLOTS * 1024, w0 = lds0 + 2 * ACT_SLOTS * 1024;
    f32x16 acc[3];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    if (PRIO == 2 && half == 1) __builtin_amdgcn_s_setprio(1);
    size_t pos = (size_t)half * ACT_SLOTS * 64;
    const size_t step = (size_t)(2 * ACT_SLOTS + W_SLOTS) * 64;
    const i32x4 srd = make_srd(base);
    const unsigned vlane = lane * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    // phase p: half (p & 1) multiplies global chunk g = p >> 1; the other half copies: its activations of the chunk it multiplies next, and its
    // half of the weights of chunk g + 1 (stage (g + 1) & 1: last read in phase 2 g - 1)
    for (int p = 0; p < 2 * iters; ++p) {
        const int g = p >> 1;
        if ((p & 1) == half) {
            if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
            mfma81(acc, lds + half * ACT_SLOTS * 64 + lane, lds + (2 * ACT_SLOTS + (g & 1) * W_SLOTS) * 64 + lane, wave, ACT_SLOTS);
            if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        } else {
            const uint4* const sp = base + pos + lane;
            const unsigned spos = (unsigned)(pos * 16);
            pos += step;
            if (pos + step > region_vec) pos = (size_t)half * ACT_SLOTS * 64;
            if (BUF) { for (int q = wave; q < ACT_SLOTS; q += 4) glds16b(srd, vlane, spos + q * 1024, act0 + q * 1024); }
            else for (int q = wave; q < ACT_SLOTS; q += 4) glds16(sp + q * 64, act0 + q * 1024);
            // weights of chunk g + 1: WSHARE = 1: this half copies its half of them (shared stage); 0: all 18 (as if nothing were shared)
            const unsigned wd = w0 + ((g + 1) & 1) * W_SLOTS * 1024;
            const uint4* const wp = base + (size_t)2 * ACT_SLOTS * 64 + lane;
            if (WSHARE && BUF) { for (int q = half * (W_SLOTS / 2) + wave; q < (half + 1) * (W_SLOTS / 2); q += 4) glds16b(srd, vlane, (2 * ACT_SLOTS + q) * 1024, wd + q * 1024); }
            else if (WSHARE) { for (int q = half * (W_SLOTS / 2) + wave; q < (half + 1) * (W_SLOTS / 2); q += 4) glds16(wp + q * 64, wd + q * 1024); }
            else if (half == 0) { for (int q = wave; q < W_SLOTS; q += 4) glds16(wp + q * 64, wd + q * 1024); }
            else { for (int q = wave; q < W_SLOTS; q += 4) glds16(wp + q * 64, wd + q * 1024); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

template <typename K>
void run(const char* name, K kern, int wgs, int threads, size_t lds, const uint4* src, size_t total_vec, int iters, double mfma_per_simd_per_iter, double bytes_per_cu_per_iter) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, (size_t)wgs * threads * 4); (void)hipMalloc(&cyc, wgs * 8);
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    static unsigned long long c[512];
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), lds, 0, out, cyc, src, total_vec / wgs, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipMemcpy(c, cyc, wgs * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < wgs; ++i) mean += (double)c[i] / wgs;
    const double us = ms * 1e3 / iters, cyc_it = mean / iters;
    const double pf = mfma_per_simd_per_iter * 1024 * 32768.0 / (us * 1e-6) / 1e15;
    printf("%-34s %7.0f cycles/iter  busy %.3f  %5.1f B/cycle/CU  %.3f us/iter  clock %.2f GHz  %.3f PF bf16 issue\n", name, cyc_it,
           mfma_per_simd_per_iter * 32 / cyc_it, bytes_per_cu_per_iter / cyc_it, us, cyc_it / us / 1e3, pf);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main(int argc, char** argv) {
    const int iters = 4000;
    for (int big = 0; big < 2; ++big) {
        const size_t total_vec = (big ? (size_t)2048 : (size_t)48) * 1024 * 1024 / 16;
        uint4* src;
        if (hipMalloc(&src, total_vec * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
        fill<<<(unsigned)((total_vec + 255) / 256), 256>>>(src, total_vec);
        (void)hipDeviceSynchronize();
        printf("-- source region %s\n", big ? "2 GB (HBM stream)" : "48 MB (Infinity Cache)");
        // per iteration a 2wg workgroup does one chunk (81 MFMAs per wave; two workgroups per CU -> 162 per SIMD), copies 54 KiB (108 per CU)
        run("2wg", k_2wg<0>, 512, 256, 54 * 1024, src, total_vec, iters, 162, 108 * 1024);
        run("2wg, setprio while multiplying", k_2wg<1>, 512, 256, 54 * 1024, src, total_vec, iters, 162, 108 * 1024);
        run("2wg, buffer_load lds", k_2wg<0, 1>, 512, 256, 54 * 1024, src, total_vec, iters, 162, 108 * 1024);
        // per iteration (two phases) the ping-pong workgroup does one chunk per half (162 MFMAs per SIMD), copies 2 x 36 + 18 KiB
        run("pingpong, shared weights", k_pp<0, 1>, 256, 512, 108 * 1024, src, total_vec, iters, 162, 90 * 1024);
        run("pingpong, shared, prio multiply", k_pp<1, 1>, 256, 512, 108 * 1024, src, total_vec, iters, 162, 90 * 1024);
        run("pingpong, shared, prio waves 4-7", k_pp<2, 1>, 256, 512, 108 * 1024, src, total_vec, iters, 162, 90 * 1024);
        run("pingpong, shared, buffer_load lds", k_pp<0, 1, 1>, 256, 512, 108 * 1024, src, total_vec, iters, 162, 90 * 1024);
        run("pingpong, weights per half", k_pp<0, 0>, 256, 512, 108 * 1024, src, total_vec, iters, 162, 108 * 1024);
        (void)hipFree(src);
    }
    return 0;
}
