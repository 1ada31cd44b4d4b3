// conv3x3 (+bias +LeakyReLU +scaled residuals, fused nearest-upsample of the input) as an implicit GEMM on
// the gfx950 bf16 MFMA pipe, fp32 accumulate, "split-bf16" operands for fp32-class accuracy.
//
// Replaces codes/models/modules/block.py:129-146,230-235,262-270,85-97,293-309 of the reference
// (see include/esr_hip.h).  Design notes live in DESIGN.md §conv3x3; the short version:
//
//   GEMM view      D[cout][pixel] = sum_{tap, cin}  W[cout][tap,cin] * X[tap,cin][pixel]
//                  A = weights (M = cout, 32 per MFMA), B = activations (N = 32 consecutive pixels),
//                  K = 16 = two 8-channel groups at one tap  ->  v_mfma_f32_32x32x16_bf16
//   activations    [B][CG][H+2][W+2][8] bf16 hi (+lo) planes with a zero border in memory: one B fragment is one
//                  aligned ds_read_b128 of an LDS pixel vector, a tap shift is +-16 B, padding needs no branches
//   tile           TH x TW output pixels per 256-thread workgroup, flattened with pitch P = TW+2 so that the
//                  32-pixel MFMA columns are 32 CONSECUTIVE LDS vectors for every tap (conflict-free b128 reads);
//                  the 2 pitch-padding columns compute garbage that is masked at the store
//   K loop         chunk = 2 channel groups x 9 taps; input tile double-buffered in LDS, next chunk prefetched into
//                  registers during the MFMAs (issue-early / write-late); weight fragments stream from L2 in
//                  fragment order (one coalesced 1 KiB load per wave per fragment)
//   split-bf16     x = hi + lo (both bf16).  acc += Wlo*Xhi + Whi*Xlo + Whi*Xhi  (3 MFMAs, lo*lo dropped: 2^-16)
//   epilogue       bias, LeakyReLU, alpha*y + beta1*r1 + beta2*r2, optional act' mask (data-gradient use), re-split to
//                  hi/lo and 8-byte stores that tile 512 contiguous bytes per wave instruction
#include "esr_common.h"

namespace {

constexpr int NW = 4;          // waves per workgroup
constexpr int NTHREADS = 256;
constexpr int MAXS = 3;        // activation DMA slots per wave per plane: NPIX_L <= MAXS*NW*64

struct ConvArgs {
    DView in0, in1;
    int ups, Win_p;                 // input upsample factor; padded input row pitch (W_in + 2)
    const uint4* wpack;
    const float* bias;
    int cout, H, W;                 // output interior
    int TH, TW, P, NPIX_T, NPIX_L, tiles_x, tiles_y, ncp;
    float act_slope, alpha, beta1, beta2;
    DView res1, res2, out, out2, mask;
    float* out_nchw;
    int mask_cg0, mask_cg1;
    float mask_slope;
};

__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// async global -> LDS copy of 16 bytes per lane; LDS destination = (wave-uniform) dst + lane*16
__device__ __forceinline__ void glds16(const uint4* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// base pointer (hi or lo) of input channel group g for image b; groups past the end alias group 0 of in1
// (their packed weights are zero, the data only has to be finite)
__device__ __forceinline__ const uint4* in_plane(const ConvArgs& a, int g, int b, bool lo) {
    if (g < a.in0.ncg) return (lo ? a.in0.lo : a.in0.hi) + b * a.in0.bs + g * a.in0.cs;
    int g1 = g - a.in0.ncg;
    if (g1 >= a.in1.ncg) g1 = 0;
    return (lo ? a.in1.lo : a.in1.hi) + b * a.in1.bs + g1 * a.in1.cs;
}

// One workgroup = one TH x TW output tile, all output channels.  K loop over chunks of 2 channel groups x 9 taps:
//   DMA (global_load_lds) the chunk's input tile and its 9*MT weight fragments into LDS -> barrier -> 9*MT*R*(3|1) MFMAs
//   per wave straight out of LDS -> barrier.  The LDS stage is single-buffered on purpose: 2-3 workgroups share a CU
//   (launch bounds + LDS budget), so one workgroup's DMA wait is covered by its neighbours' MFMAs, no staging
//   registers or ds_writes exist, and every global access of the loop is an asynchronous 1 KiB-per-wave DMA.
template <int NPL, int MT, int R>
__global__ __launch_bounds__(NTHREADS, MT == 1 ? 3 : 2) void conv3x3_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int x0 = tx * a.TW, y0 = ty * a.TH;   // tile origin: output interior coords == padded coords of the halo origin
    const int P = a.P;
    const int plane_bytes = a.NPIX_L * 16;
    unsigned char* const s_act = smem;                              // [2 groups][NPL][NPIX_L] pixel vectors
    unsigned char* const s_w = smem + 2 * NPL * plane_bytes;        // [9 taps][MT][NPL] fragments of 1 KiB
    constexpr int NWI = 9 * MT * NPL;                               // weight DMA instructions per chunk

    // ---- per-lane source offsets of the activation DMA slots this wave issues (chunk independent).
    // slot s covers LDS pixels [(wave + s*NW)*64, +64); out-of-image pixels read the plane's (0,0) border vector (zero).
    int soff[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        const int p = (wave + s * NW) * 64 + lane;
        const int rr = p / P, cc = p - rr * P;
        const int Yp = y0 + rr, Xp = x0 + cc;
        const bool inb = (p < a.NPIX_T) && (Yp < a.H + 2) && (Xp < a.W + 2);
        int sy = Yp, sx = Xp;
        if (a.ups == 2) { sy = (Yp + 1) >> 1; sx = (Xp + 1) >> 1; }
        else if (a.ups > 2) { sy = (Yp - 1 + a.ups) / a.ups; sx = (Xp - 1 + a.ups) / a.ups; }
        soff[s] = (p < a.NPIX_L) ? (inb ? sy * a.Win_p + sx : 0) : -1;
    }

    f32x16 acc[MT][R];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][r][i] = 0.f;

    // lane's B-fragment base: channel group = lane>>5, pixel column = lane&31, N-tile = wave + r*NW
    const unsigned char* const sb = s_act + (lane >> 5) * NPL * plane_bytes + (wave * 32 + (lane & 31)) * 16;
    const unsigned char* const sa = s_w + lane * 16;

    for (int cp = 0; cp < a.ncp; ++cp) {
        // ---- stage chunk cp
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                const uint4* src = in_plane(a, 2 * cp + cg, b, pl == 1);
                unsigned char* dst = s_act + (cg * NPL + pl) * plane_bytes + wave * 1024;
#pragma unroll
                for (int s = 0; s < MAXS; ++s)
                    if (soff[s] >= 0) glds16(src + soff[s], dst + s * NW * 1024);
            }
        }
        {
            const uint4* wsrc = a.wpack + (size_t)cp * NWI * 64 + lane;
#pragma unroll
            for (int j0 = 0; j0 < NWI; j0 += NW) {
                const int j = j0 + wave;
                if (j < NWI) glds16(wsrc + j * 64, s_w + j * 1024);
            }
        }
        __syncthreads();   // (drains this wave's DMA first: the compiler puts s_waitcnt vmcnt(0) in front of the barrier)

#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tapoff = ((t / 3) * P + (t % 3)) * 16;
            uint4 ah[MT], al[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ah[m] = *(const uint4*)(sa + ((t * MT + m) * NPL) * 1024);
                if (NPL == 2) al[m] = *(const uint4*)(sa + ((t * MT + m) * NPL + 1) * 1024);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned char* pb = sb + r * NW * 512 + tapoff;
                const uint4 bh = *(const uint4*)pb;
                if (NPL == 2) {
                    const uint4 bl = *(const uint4*)(pb + plane_bytes);
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][r] = mfma(al[m], bh, acc[m][r]);
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][r] = mfma(ah[m], bl, acc[m][r]);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m][r] = mfma(ah[m], bh, acc[m][r]);
            }
        }
        __syncthreads();
    }

    // ---- epilogue.  D layout (32x32 MFMA): lane holds pixel column j = lane&31 and, for register i,
    // output row (i&3) + 8*(i>>2) + 4*(lane>>5): i>>2 selects the 8-channel group inside the 32-row tile,
    // (i&3) + 4*(lane>>5) the channel inside the group -> 4 consecutive channels = 8 bytes of bf16.
    const int half = lane >> 5;
    const long long Wp = a.W + 2;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int nt = wave + r * NW;
        const int q = nt * 32 + (lane & 31);
        const int rr = q / P, cc = q - rr * P;
        const int Y = y0 + rr, X = x0 + cc;
        const bool valid = (rr < a.TH) && (cc < a.TW) && (Y < a.H) && (X < a.W);
        if (!valid) continue;
        const long long pix = (long long)(Y + 1) * Wp + (X + 1);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int cg = m * 4 + g4;             // output channel group
                const int ch0 = cg * 8 + half * 4;     // first of this lane's 4 channels
                if (cg * 8 >= a.cout) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[m][r][g4 * 4 + i];
                if (a.bias) {
                    const float4 bz = *(const float4*)(a.bias + ch0);
                    v[0] += bz.x; v[1] += bz.y; v[2] += bz.z; v[3] += bz.w;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = v[i] > 0.f ? v[i] : v[i] * a.act_slope;
                    v[i] *= a.alpha;
                }
                if (a.res1.hi) {
                    const long long o = b * a.res1.bs + cg * a.res1.cs + pix;
                    const uint2 h = ((const uint2*)(a.res1.hi + o))[half];
                    float rv[4] = {bf2f(h.x & 0xFFFF), bf2f(h.x >> 16), bf2f(h.y & 0xFFFF), bf2f(h.y >> 16)};
                    if (a.res1.lo) {
                        const uint2 l = ((const uint2*)(a.res1.lo + o))[half];
                        rv[0] += bf2f(l.x & 0xFFFF); rv[1] += bf2f(l.x >> 16); rv[2] += bf2f(l.y & 0xFFFF); rv[3] += bf2f(l.y >> 16);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaf(a.beta1, rv[i], v[i]);
                }
                if (a.res2.hi) {
                    const long long o = b * a.res2.bs + cg * a.res2.cs + pix;
                    const uint2 h = ((const uint2*)(a.res2.hi + o))[half];
                    float rv[4] = {bf2f(h.x & 0xFFFF), bf2f(h.x >> 16), bf2f(h.y & 0xFFFF), bf2f(h.y >> 16)};
                    if (a.res2.lo) {
                        const uint2 l = ((const uint2*)(a.res2.lo + o))[half];
                        rv[0] += bf2f(l.x & 0xFFFF); rv[1] += bf2f(l.x >> 16); rv[2] += bf2f(l.y & 0xFFFF); rv[3] += bf2f(l.y >> 16);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaf(a.beta2, rv[i], v[i]);
                }
                if (a.mask.hi && cg >= a.mask_cg0 && cg < a.mask_cg1) {
                    const long long o = b * a.mask.bs + (cg - a.mask_cg0) * a.mask.cs + pix;
                    const uint2 h = ((const uint2*)(a.mask.hi + o))[half];
                    // sign of the stored (post-activation) value == sign of the pre-activation (slope > 0)
                    const uint32_t s[4] = {h.x & 0x8000u, h.x & 0x80000000u, h.y & 0x8000u, h.y & 0x80000000u};
                    const uint32_t nz[4] = {h.x & 0x7FFFu, h.x & 0x7FFF0000u, h.y & 0x7FFFu, h.y & 0x7FFF0000u};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (s[i] || !nz[i]) v[i] *= a.mask_slope;   // x <= 0 -> slope (torch: grad of leaky_relu at 0 is slope)
                }
                if (a.out_nchw) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (ch0 + i < a.cout)
                            a.out_nchw[((long long)(b * a.cout + ch0 + i) * a.H + Y) * a.W + X] = v[i];
                }
                if (a.out.hi) {
                    uint32_t hh[4], ll[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (ch0 + i >= a.cout) v[i] = 0.f;
                        split_bf16(v[i], hh[i], ll[i]);
                    }
                    const uint2 hv = make_uint2(hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16));
                    const uint2 lv = make_uint2(ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16));
                    const long long o = b * a.out.bs + cg * a.out.cs + pix;
                    ((uint2*)(a.out.hi + o))[half] = hv;
                    if (a.out.lo) ((uint2*)(a.out.lo + o))[half] = lv;
                    if (a.out2.hi) {
                        const long long o2 = b * a.out2.bs + cg * a.out2.cs + pix;
                        ((uint2*)(a.out2.hi + o2))[half] = hv;
                        if (a.out2.lo) ((uint2*)(a.out2.lo + o2))[half] = lv;
                    }
                }
            }
        }
    }
}

// ---- weight packing: [M][K][3][3] fp32 -> [kstep = cp*9+tap][mtile][hi|lo][lane][8] bf16
__global__ void pack_weights_kernel(const float* __restrict__ w, int dim0, int dim1, const int* __restrict__ kmap, int ncg_in,
                                    const int* __restrict__ mmap, int mtiles, int transposed, int npl, uint4* __restrict__ out, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (kstep, mtile, lane)
    if (idx >= total) return;
    const int lane = idx & 63;
    const int m = (idx >> 6) % mtiles;
    const int ks = (idx >> 6) / mtiles;
    const int cp = ks / 9, t = ks % 9;
    const int cg = 2 * cp + (lane >> 5);
    const int mch = mmap[m * 32 + (lane & 31)];
    uint32_t hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kch = cg < ncg_in ? kmap[cg * 8 + e] : -1;
        float v = 0.f;
        if (kch >= 0 && mch >= 0)
            v = transposed ? w[((long long)kch * dim1 + mch) * 9 + (8 - t)] : w[((long long)mch * dim1 + kch) * 9 + t];
        split_bf16(v, hi[e], lo[e]);
    }
    uint4* o = out + ((size_t)(ks * mtiles + m) * npl) * 64 + lane;
    o[0] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
    if (npl == 2) o[64] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
}

struct TileCfg { int R, TH, TW, P, NPIX_T, NPIX_L, tiles_x, tiles_y; size_t lds; };

// Choose (R, TH, TW): minimise MFMA work = workgroups * R (every wave always runs R column tiles) under the LDS budget that
// keeps `occ` workgroups resident per CU.
TileCfg pick_tile(int H, int W, int npl, int mt, int occ) {
    TileCfg best{};
    double best_cost = -1;
    const size_t budget = (size_t)(160 * 1024) / occ;
    for (int R = 2; R <= 4; ++R) {
        const int max_px = 32 * NW * R;
        for (int ntx = 1; ntx <= W; ++ntx) {
            const int TW = (W + ntx - 1) / ntx;
            const int P = TW + 2;
            if (P > max_px) continue;
            if (ntx > 1 && TW < 6) break;
            int THmax = max_px / P;
            if (THmax > H) THmax = H;
            for (int TH = THmax; TH >= 1 && TH >= THmax - 6; --TH) {
                const int npix_t = (TH + 2) * P;
                int npix_l = max_px + 2 * P + 2;
                if (npix_l < npix_t) npix_l = npix_t;
                if (npix_l > MAXS * NW * 64) continue;
                const size_t lds = (size_t)2 * npl * npix_l * 16 + (size_t)9 * mt * npl * 1024;
                if (lds > budget) continue;
                const int nty = (H + TH - 1) / TH;
                const double cost = (double)ntx * nty * (R + 0.35);
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    best = TileCfg{R, TH, TW, P, npix_t, npix_l, ntx, nty, lds};
                }
            }
        }
    }
    return best;
}

template <int NPL, int MT, int R>
int launch(const ConvArgs& a, int B, size_t lds, hipStream_t s) {
    auto k = conv3x3_kernel<NPL, MT, R>;
    static bool attr_set = false;   // benign race: idempotent
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(k, dim3(a.tiles_x * a.tiles_y * B), dim3(NTHREADS), lds, s, a);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

template <int NPL, int MT>
int launch_r(const ConvArgs& a, int B, const TileCfg& t, hipStream_t s) {
    switch (t.R) {
        case 2: return launch<NPL, MT, 2>(a, B, t.lds, s);
        case 3: return launch<NPL, MT, 3>(a, B, t.lds, s);
        default: return launch<NPL, MT, 4>(a, B, t.lds, s);
    }
}

}  // namespace

extern "C" size_t esr_conv_wpack_bytes(int ncg_in, int cout, int split) {
    const int ncp = (ncg_in + 1) / 2, mt = (cout + 31) / 32;
    return (size_t)ncp * 9 * mt * (split ? 2 : 1) * 64 * 16;
}

extern "C" int esr_pack_conv_weights(const float* w, int cout_w, int cin_w, const int32_t* kmap, int ncg_in, const int32_t* mmap,
                                     int mtiles, int transposed, int split, void* wpack, esr_stream_t stream) {
    if (!w || !kmap || !mmap || !wpack || ncg_in <= 0 || mtiles <= 0) return ESR_E_ARG;
    const int ncp = (ncg_in + 1) / 2;
    const int total = ncp * 9 * mtiles * 64;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, cout_w, cin_w, kmap,
                       ncg_in, mmap, mtiles, transposed, split ? 2 : 1, (uint4*)wpack, total);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_conv3x3(const esr_conv3x3_desc* d, esr_stream_t stream) {
    if (!d || !d->in1.hi || !d->wpack || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->cout <= 0) return ESR_E_ARG;
    if (!d->out.hi && !d->out_nchw) return ESR_E_ARG;
    const int ups = d->upsample <= 0 ? 1 : d->upsample;
    if (ups > 1 && d->in0.hi) return ESR_E_UNSUPPORTED;
    if (d->in1.H * ups != d->H || d->in1.W * ups != d->W) return ESR_E_ARG;
    if (d->in0.hi && (d->in0.H != d->H || d->in0.W != d->W)) return ESR_E_ARG;
    const bool split = d->in1.lo != nullptr;
    if (d->in0.hi && ((d->in0.lo != nullptr) != split)) return ESR_E_ARG;
    const int mt = (d->cout + 31) / 32;
    if (mt > 2) return ESR_E_UNSUPPORTED;   // callers split wider outputs into 64-channel launches
    if (d->out.hi && d->out.ncg * 8 < d->cout) return ESR_E_ARG;

    ConvArgs a{};
    a.in0 = to_dview(d->in0);
    a.in1 = to_dview(d->in1);
    a.ups = ups;
    a.Win_p = d->in1.W + 2;
    a.wpack = (const uint4*)d->wpack;
    a.bias = d->bias;
    a.cout = d->cout;
    a.H = d->H;
    a.W = d->W;
    const int npl = split ? 2 : 1;
    TileCfg t = pick_tile(d->H, d->W, npl, mt, mt == 1 ? 3 : 2);
    if (t.R == 0) t = pick_tile(d->H, d->W, npl, mt, 1);
    if (t.R == 0) return ESR_E_UNSUPPORTED;
    a.TH = t.TH; a.TW = t.TW; a.P = t.P; a.NPIX_T = t.NPIX_T; a.NPIX_L = t.NPIX_L;
    a.tiles_x = t.tiles_x; a.tiles_y = t.tiles_y;
    a.ncp = (a.in0.ncg + a.in1.ncg + 1) / 2;
    a.act_slope = d->act_slope;
    a.alpha = d->alpha;
    a.beta1 = d->beta1;
    a.beta2 = d->beta2;
    a.res1 = to_dview(d->res1);
    a.res2 = to_dview(d->res2);
    a.out = to_dview(d->out);
    a.out2 = to_dview(d->out2);
    a.mask = to_dview(d->mask_src);
    a.out_nchw = d->out_nchw;
    a.mask_cg0 = d->mask_cg0;
    a.mask_cg1 = d->mask_cg1;
    a.mask_slope = d->mask_slope;
    hipStream_t s = (hipStream_t)stream;
    if (split) return mt == 1 ? launch_r<2, 1>(a, d->B, t, s) : launch_r<2, 2>(a, d->B, t, s);
    return mt == 1 ? launch_r<1, 1>(a, d->B, t, s) : launch_r<1, 2>(a, d->B, t, s);
}
