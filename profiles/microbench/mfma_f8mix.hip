// Can an fp32-class conv spend fewer MFMA cycles than split-bf16's three terms?  Candidate ("f16f8"): the main term Whi*Xhi on the fp16
// pipe, the two correction terms Whi*Xlo + Wlo*Xhi as ONE fp8 (e4m3) K=64 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, 2x the bf16 rate):
// per pair of taps of a 2-group chunk, 2 f16 MFMAs + 1 fp8 MFMA = 128 pipe cycles instead of 6 bf16 MFMAs = 192.
// Part 1 pins the instruction's semantics (operand rows/columns, the e8m0 scales).  Part 2 measures, with random operands, LDS operand
// reads and an LDS-DMA stream at the conv kernel's rates, what the chip sustains under its power cap for the three instruction mixes.
//   hipcc --offload-arch=gfx950 -O3 mfma_f8mix.hip -o mfma_f8mix && ./mfma_f8mix
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---------------------------------------------------------------- part 1: semantics
__global__ void probe(const i32x8* a, const i32x8* b, float* out, int sa, int sb) {
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = c[i];
}
static unsigned char e4m3(float v) {          // exact for the small values used here
    if (v == 0) return 0;
    unsigned char s = v < 0 ? 0x80 : 0;
    v = fabsf(v);
    int e; float m = frexpf(v, &e);            // v = m * 2^e, m in [0.5, 1)
    int E = e - 1 + 7;                         // biased exponent of 1.xxx * 2^(e-1)
    if (E <= 0) { int q = (int)lrintf(v * 512.f); return s | (unsigned char)q; }    // subnormal: q * 2^-9
    int q = (int)lrintf((m * 2 - 1) * 8);
    if (q == 8) { q = 0; ++E; }
    return s | (unsigned char)((E << 3) | q);
}
static int run_probe() {
    std::vector<float> A(32 * 64), B(64 * 32);
    srand(3);
    for (auto& x : A) x = (float)(rand() % 9 - 4) * 0.25f;
    for (auto& x : B) x = (float)(rand() % 13 - 6) * 0.5f;
    // hypothesis: lane l holds row (A) / column (B) l%32 and the 32 consecutive K values of block l/32, byte j = K index 32*(l/32)+j
    std::vector<unsigned char> ha(64 * 32), hb(64 * 32);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            const int k = 32 * (l / 32) + j;
            ha[l * 32 + j] = e4m3(A[(l % 32) * 64 + k]);
            hb[l * 32 + j] = e4m3(B[k * 32 + (l % 32)]);
        }
    i32x8 *da, *db; float* dout;
    (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dout, 64 * 16 * 4);
    (void)hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
    int bad_total = 0;
    const int scales[4][2] = {{127, 127}, {126, 127}, {127, 120}, {106, 127}};
    for (int t = 0; t < 4; ++t) {
        probe<<<1, 64>>>(da, db, dout, scales[t][0], scales[t][1]);
        std::vector<float> out(64 * 16);
        (void)hipMemcpy(out.data(), dout, 64 * 16 * 4, hipMemcpyDeviceToHost);
        const double sc = ldexp(1.0, scales[t][0] - 127 + scales[t][1] - 127);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 16; ++i) {
                const int n = l % 32, m = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
                double ref = 0;
                for (int k = 0; k < 64; ++k) ref += (double)A[m * 64 + k] * B[k * 32 + n];
                if (fabs(out[l * 16 + i] - ref * sc) > 1e-6 * fabs(ref * sc) + 1e-12) ++bad;
            }
        printf("probe scale_a=%d scale_b=%d (x%g): %d / 1024 mismatches\n", scales[t][0], scales[t][1], sc, bad);
        bad_total += bad;
    }
    return bad_total;
}

// ---------------------------------------------------------------- part 2: sustained rate under the power cap
__device__ __forceinline__ void glds16(const uint4* src, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_dst) : "memory");
}
// MODE 0: split-bf16, 18 MFMAs per iteration (2 taps x 3 column tiles x 3 terms)      16 LDS reads
// MODE 1: f16f8, 6 f16 MFMAs + 3 fp8 K=64 MFMAs per iteration                          16 LDS reads
// MODE 2: one-term f16, 6 MFMAs per iteration                                           8 LDS reads
// MODE 3: f16f8 with the fp8 MFMAs into a second accumulator set (no scale needed)
template <int MODE, int NDMA, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void k(float* out, unsigned long long* cyc, const uint4* src, int iters) {
    __shared__ uint4 lds[4096];   // 64 KiB
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return h; };
    // random bits that are finite in every format used: f16/bf16 exponent field kept small, fp8 bytes never 0x7F/0xFF
    auto rv = [&]() { const unsigned r = nx(); return (r & 0x83ff83ffu) | 0x38003800u; };
    auto rv8 = [&]() { return nx() & 0xBFBFBFBFu; };          // e4m3 bytes with |v| < 2: never NaN
    for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = (i & 2048) ? make_uint4(rv8(), rv8(), rv8(), rv8()) : make_uint4(rv(), rv(), rv(), rv());
    __syncthreads();
    f32x16 acc[3], acc2[3];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 16; ++i) { acc[j][i] = 0.f; acc2[j][i] = 0.f; }
    uint4 a[4], b[3][4];
    for (int i = 0; i < 4; ++i) a[i] = (i >= 2 && MODE != 0) ? make_uint4(rv8(), rv8(), rv8(), rv8()) : make_uint4(rv(), rv(), rv(), rv());
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 4; ++i) b[j][i] = (i >= 2 && MODE != 0) ? make_uint4(rv8(), rv8(), rv8(), rv8()) : make_uint4(rv(), rv(), rv(), rv());
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds;
    const int sc_a = 106, sc_b = 127;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int base = ((it * 8 + wave * 67) & 15) * 64 + lane;
        constexpr int NA = MODE == 2 ? 2 : 4;
#pragma unroll
        for (int i = 0; i < NA; ++i) a[i] = lds[base + i * 1024];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < NA; ++i) b[j][i] = lds[base + 64 * (j + 1) + i * 1024];
#pragma unroll
        for (int d = 0; d < NDMA; ++d) glds16(src + ((it * NDMA + d) & 1023) * 64 + lane, lds0 + 32768 + ((wave * NDMA + d) & 31) * 1024);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (MODE == 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[2 * t + 1]), __builtin_bit_cast(bf16x8, b[j][2 * t]), acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[2 * t]), __builtin_bit_cast(bf16x8, b[j][2 * t + 1]), acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[2 * t]), __builtin_bit_cast(bf16x8, b[j][2 * t]), acc[j], 0, 0, 0);
                }
            } else {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[j][0]), acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[j][1]), acc[j], 0, 0, 0);
                if (MODE == 1 || MODE == 3) {
                    i32x8 fa, fb;
                    fa[0] = a[2].x; fa[1] = a[2].y; fa[2] = a[2].z; fa[3] = a[2].w; fa[4] = a[3].x; fa[5] = a[3].y; fa[6] = a[3].z; fa[7] = a[3].w;
                    fb[0] = b[j][2].x; fb[1] = b[j][2].y; fb[2] = b[j][2].z; fb[3] = b[j][2].w; fb[4] = b[j][3].x; fb[5] = b[j][3].y; fb[6] = b[j][3].z; fb[7] = b[j][3].w;
                    if (MODE == 1) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc[j], 0, 0, 0, sc_a, 0, sc_b);
                    else acc2[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc2[j], 0, 0, 0, 127, 0, 127);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i] + acc2[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NDMA, int WAVES>
void run(int wgs_per_cu, int iters) {
    const int wgs = 256 * wgs_per_cu;
    float* out; unsigned long long* cyc; uint4* src;
    (void)hipMalloc(&out, wgs * WAVES * 64 * 4); (void)hipMalloc(&cyc, wgs * 8); (void)hipMalloc(&src, 1024 * 1024);
    (void)hipMemset(src, 0x3c, 1024 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    unsigned long long c = 0;
    for (int rep = 0; rep < 4; ++rep) {     // ~0.2 s per rep: long enough for the power management to settle
        (void)hipEventRecord(e0);
        k<MODE, NDMA, WAVES><<<wgs, WAVES * 64>>>(out, cyc, src, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    }
    static const char* names[] = {"split-bf16 (18 bf16 MFMA)", "f16f8 (6 f16 + 3 fp8x64, one acc)", "one-term f16 (6 MFMA)", "f16f8 (6 f16 + 3 fp8x64, two accs)"};
    // one iteration = 2 taps x 3 column tiles of a 2-group chunk for one M-tile = 2*3*32*32*16 MACs of the convolution
    const double conv_flop = (double)wgs * WAVES * iters * 6 * 32768.0;
    printf("%-36s dma/iter %d  WG/CU %d: %.1f ms  ns per tap-pair group %.1f  conv-equivalent %.0f TFLOP/s  wave-cycles/iter %.0f  clock %.2f GHz\n", names[MODE], NDMA,
           wgs_per_cu, ms, ms * 1e6 / iters, conv_flop / ms / 1e9, (double)c / iters, c / (ms * 1e6));
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(src);
}

int main() {
    if (run_probe()) printf("PROBE FAILED\n");
    const int N = 400000;
    run<0, 0, 4>(2, N); run<1, 0, 4>(2, N); run<3, 0, 4>(2, N); run<2, 0, 4>(2, N);
    run<0, 4, 4>(2, N); run<1, 4, 4>(2, N); run<3, 4, 4>(2, N); run<2, 2, 4>(2, N);
    return 0;
}
