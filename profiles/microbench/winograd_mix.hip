// Would Winograd F(2x2, 3x3) pay for the cout-64 convs of the generator in the fp32-class 'split' arithmetic on this part?  
// An instruction-mix benchmark at the conv kernel's own granularity — what ONE 4-wave workgroup does per K chunk (16 input channels) for 384
// output pixels and 64 output channels, two workgroups resident per CU, operands really coming out of LDS, the LDS-DMA stream running:
//   direct   (csrc/esr_conv.hip today): 9 taps x 2 M tiles x 3 pixel tiles x 3 split terms = 162 MFMAs per wave, 90 ds_read_b128, 13 copies
//   winograd : 96 tiles of 2x2 outputs.  Input transform V = B^T d B in fp32 on the VALU — per lane one (tile, 8-channel group): 32 ds_read_b128
//              (4x4 patch, hi + lo planes; as 64 ds_read_b64 here: two halves of 4 channels keep the transform in 128 registers), hi+lo -> fp32,
//              32 adds per channel, re-split to bf16 hi / lo (v_cvt_pk_bf16_f32, subtract, convert), 32 ds_write_b128 (64 b64) — 192 such lane tasks per chunk = 3/4 of a wave-task per wave; then 16 positions x 2 M tiles x 3 tile
//              blocks x 3 terms / 4 waves = 72 MFMAs per wave with their fragment reads (16 x 2 A planes per M tile + 16 x 2 B planes per tile
//              block), and 20 copies (the transformed weights are 16 taps instead of 9).  The output transform (once per tile, not per K chunk) and
//              the halo overhead of 4x4 patches are left OUT — in Winograd's favour.
// Both loops run on random operand bits (the power draw of MFMAs depends on the data).  Reported: ms per 100k chunk-iterations.
//   hipcc --offload-arch=gfx950 -O3 winograd_mix.hip -o bin/winograd_mix && ./bin/winograd_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ void glds16(const uint4* src, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ f32x16 mf(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pk(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}

constexpr int LDS_VEC = 6144;      // 96 KiB of LDS per workgroup?  no: 6144 x 16 B = 96 KiB would stop two from being resident; see below
constexpr int NV = 3584;           // 56 KiB: two workgroups per CU fit (the conv kernel's stage is ~52 KiB)

// MODE 0: direct mix; MODE 1: winograd mix
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, const uint4* src, int iters) {
    __shared__ uint4 lds[NV];
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (h & 0x83ff83ffu) | 0x3c003c00u; };
    for (int i = threadIdx.x; i < NV; i += 256) lds[i] = make_uint4(nx(), nx(), nx(), nx());
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds;
    f32x16 acc[6];
    for (int j = 0; j < 6; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int base = ((it * 5 + wave * 67) & 15) * 64 + lane;       // fragment base inside the first 2048 vectors
        // ---- copies of the chunk's operands (input tile + weights) from an L2-resident buffer
        constexpr int NDMA = MODE == 0 ? 13 : 20;
#pragma unroll
        for (int d = 0; d < NDMA; ++d) glds16(src + ((it * NDMA + d) & 1023) * 64 + lane, lds0 + (2048 + ((wave * NDMA + d) % 24) * 64) * 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (MODE == 0) {
            // 9 taps: per tap 2 M tiles x 2 planes of A, 3 pixel tiles x 2 planes of B, 18 MFMAs
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                uint4 a[2][2], b[3][2];
#pragma unroll
                for (int m = 0; m < 2; ++m) { a[m][0] = lds[base + (t * 4 + m * 2) * 16 % 1024]; a[m][1] = lds[base + (t * 4 + m * 2 + 1) * 16 % 1024]; }
#pragma unroll
                for (int r = 0; r < 3; ++r) { b[r][0] = lds[1024 + (base + t + r * 32) % 1024]; b[r][1] = lds[1024 + (base + t + r * 32 + 17) % 1024]; }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        acc[m * 3 + r] = mf(a[m][1], b[r][0], acc[m * 3 + r]);
                        acc[m * 3 + r] = mf(a[m][0], b[r][1], acc[m * 3 + r]);
                        acc[m * 3 + r] = mf(a[m][0], b[r][0], acc[m * 3 + r]);
                    }
            }
        } else {
            // ---- input transform: 3 of 4 iterations carry a lane task (192 tasks / 256 lanes)
            if ((it & 3) != 3) {
                // two halves of 4 channels each (64 + 64 live floats instead of 256: no spills)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float d[4][4][4];
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        const uint2 vh = *(const uint2*)((const char*)&lds[(base + p * 37) % 1024] + half * 8);
                        const uint2 vl = *(const uint2*)((const char*)&lds[1024 + (base + p * 37) % 1024] + half * 8);
                        const uint32_t wh[2] = {vh.x, vh.y}, wl[2] = {vl.x, vl.y};
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            d[p / 4][p % 4][2 * e] = __uint_as_float(wh[e] << 16) + __uint_as_float(wl[e] << 16);
                            d[p / 4][p % 4][2 * e + 1] = __uint_as_float(wh[e] & 0xffff0000u) + __uint_as_float(wl[e] & 0xffff0000u);
                        }
                    }
                    float v[4][4][4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t[4][4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {           // B^T d  (rows)
                            t[0][c] = d[0][c][e] - d[2][c][e];
                            t[1][c] = d[1][c][e] + d[2][c][e];
                            t[2][c] = d[2][c][e] - d[1][c][e];
                            t[3][c] = d[1][c][e] - d[3][c][e];
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {           // (.) B  (columns)
                            v[r][0][e] = t[r][0] - t[r][2];
                            v[r][1][e] = t[r][1] + t[r][2];
                            v[r][2][e] = t[r][2] - t[r][1];
                            v[r][3][e] = t[r][1] - t[r][3];
                        }
                    }
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        uint32_t hi[2], lo[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float x0 = v[p / 4][p % 4][2 * e], x1 = v[p / 4][p % 4][2 * e + 1];
                            hi[e] = pk(x0, x1);
                            lo[e] = pk(x0 - __uint_as_float(hi[e] << 16), x1 - __uint_as_float(hi[e] & 0xffff0000u));
                        }
                        *(uint2*)((char*)&lds[2560 + ((lane + p * 64 + wave * 16) & 511)] + half * 8) = make_uint2(hi[0], hi[1]);
                        *(uint2*)((char*)&lds[3072 + ((lane + p * 64 + wave * 16) & 511)] + half * 8) = make_uint2(lo[0], lo[1]);
                    }
                }
            }
            __syncthreads();
            // ---- 16 positions: this wave's M tile (2 planes of A), 3 tile blocks... 72 MFMAs per wave = 16 positions x (1.5 tile blocks) x 3 terms
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const uint4 a0 = lds[(base + p * 32) % 1024], a1 = lds[(base + p * 32 + 16) % 1024];
                uint4 b[2][2];
                b[0][0] = lds[2560 + ((lane + p * 64) & 511)]; b[0][1] = lds[3072 + ((lane + p * 64) & 511)];
                acc[0] = mf(a1, b[0][0], acc[0]);
                acc[0] = mf(a0, b[0][1], acc[0]);
                acc[0] = mf(a0, b[0][0], acc[0]);
                if (p & 1) {                                 // the other half tile block: every second position
                    b[1][0] = lds[2560 + ((lane + p * 64 + 32) & 511)]; b[1][1] = lds[3072 + ((lane + p * 64 + 32) & 511)];
                    acc[1] = mf(a1, b[1][0], acc[1]);
                    acc[1] = mf(a0, b[1][1], acc[1]);
                    acc[1] = mf(a0, b[1][0], acc[1]);
                }
            }
        }
        __syncthreads();
    }
    float s = 0;
    for (int j = 0; j < 6; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
float run(int iters) {
    const int wgs = 512;
    float* out; uint4* src;
    (void)hipMalloc(&out, wgs * 256 * 4); (void)hipMalloc(&src, 1024 * 1024);
    (void)hipMemset(src, 0x3c, 1024 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {          // >= 0.2 s per repetition: the power management settles
        (void)hipEventRecord(e0);
        k<MODE><<<wgs, 256>>>(out, src, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipFree(out); (void)hipFree(src);
    return ms;
}

int main() {
    const int N = 100000;
    const float d = run<0>(N), w = run<1>(N);
    const double mf_d = 512.0 * 4 * N * 162, mf_w = 512.0 * 4 * N * 72;
    printf("direct   : %.1f ms per %d chunk-iterations  (162 MFMA / wave / chunk: %.0f TFLOP/s of bf16 issue)\n", d, N, mf_d * 32768 / d / 1e9);
    printf("winograd : %.1f ms per %d chunk-iterations  ( 72 MFMA / wave / chunk + input transform: %.0f TFLOP/s of bf16 issue)\n", w, N, mf_w * 32768 / w / 1e9);
    printf("speed-up of the K loop alone (output transform and 4x4-patch halo NOT charged): %.2fx  (2.25x fewer MFMAs)\n", d / w);
    return 0;
}
