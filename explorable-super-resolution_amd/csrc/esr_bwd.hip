// Backward-pass helpers that are not convolutions: elementwise / pooling ops on the channel-group activation layout and the
// adjoints of the module-boundary packing (replicate padding, latent bilinear /sf).  HBM-bound streaming kernels: one 16-byte
// pixel vector (8 channels) per thread, coalesced along W.
//
// Reference operations whose gradients these implement (autograd in the reference):
//   nearest upsample            codes/models/modules/block.py:293-300 (Upsampler)          -> sum-pool s x s
//   LeakyReLU(0.2)              block.py:18                                                 -> * (x > 0 ? 1 : 0.2)
//   ReplicationPad2d            codes/CEM/CEMnet.py:70-71,286-295                           -> fold the pad ring into the edge pixels
//   bilinear /sf of the latent  codes/models/modules/architecture.py:284                    -> spread each LR gradient over its taps
#include "esr_common.h"
#include <vector>

namespace {

// 16-bit element of either plane format -> fp32 (fmt: ESR_FMT_BF16 / ESR_FMT_F16, uniform per view)
__device__ __forceinline__ float el2f(uint32_t bits, int fmt) { return fmt == ESR_FMT_F16 ? h2f(bits) : bf2f(bits); }

__device__ __forceinline__ void unpack8(const uint4* hi, const uint4* lo, long long o, float (&v)[8], int fmt) {
    const uint4 h = hi[o];
    const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = el2f(hw[e] & 0xFFFF, fmt); v[2 * e + 1] = el2f(hw[e] >> 16, fmt); }
    if (lo) {
        const uint4 l = lo[o];
        const uint32_t lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += el2f(lw[e] & 0xFFFF, fmt); v[2 * e + 1] += el2f(lw[e] >> 16, fmt); }
    }
}

__device__ __forceinline__ void pack8(uint4* hi, uint4* lo, long long o, const float (&v)[8], int fmt) {
    uint32_t h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (fmt == ESR_FMT_F16) { h[e] = f2h(v[e]); l[e] = f2h(v[e] - h2f(h[e])); }
        else split_bf16(v[e], h[e], l[e]);
    }
    hi[o] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    if (lo) lo[o] = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

// out = alpha * A + beta * sumpool_s(Bv), optionally * leaky_relu'(mask).  A / mask / out share out's size; Bv is s x larger.
__global__ void act_combine_kernel(DView A, float alpha, DView Bv, float beta, int s, DView M, float slope, DView out, int H, int W, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, cg, y, x) over the interior
    if (idx >= total) return;
    const int x = (int)(idx % W);
    long long t = idx / W;
    const int y = (int)(t % H);
    t /= H;
    const int cg = (int)(t % out.ncg);
    const int b = (int)(t / out.ncg);
    const long long pix = (long long)(y + 1) * (W + 2) + (x + 1);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (A.hi) {
        float a8[8];
        unpack8(A.hi, A.lo, b * A.bs + cg * A.cs + pix, a8, A.fmt);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = alpha * a8[e];
    }
    if (Bv.hi) {
        const int Wb = W * s + 2;
        for (int dy = 0; dy < s; ++dy)
            for (int dx = 0; dx < s; ++dx) {
                float b8[8];
                unpack8(Bv.hi, Bv.lo, b * Bv.bs + cg * Bv.cs + (long long)(y * s + dy + 1) * Wb + (x * s + dx + 1), b8, Bv.fmt);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(beta, b8[e], v[e]);
            }
    }
    if (M.hi) {
        const uint4 h = M.hi[b * M.bs + cg * M.cs + pix];
        const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t bits = (e & 1) ? (hw[e >> 1] >> 16) : (hw[e >> 1] & 0xFFFF);
            if ((bits & 0x8000u) || !(bits & 0x7FFFu)) v[e] *= slope;   // stored activation <= 0
        }
    }
    pack8((uint4*)out.hi, (uint4*)out.lo, b * out.bs + cg * out.cs + pix, v, out.fmt);
}

// Adjoint of the conv kernel's pixel-shuffle store (esr_hip.h): dst[g*r^2 + s][y][x] = src[g][r*y + s/r][r*x + s%r]; one 16-byte vector per thread
__global__ void pixel_unshuffle_kernel(DView src, int r, DView dst, int H, int W, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, dst group, y, x) over dst's interior
    if (idx >= total) return;
    const int x = (int)(idx % W);
    long long t = idx / W;
    const int y = (int)(t % H);
    t /= H;
    const int cg = (int)(t % dst.ncg);
    const int b = (int)(t / dst.ncg);
    const int g = cg / (r * r), sp = cg % (r * r);
    const long long so = b * src.bs + g * src.cs + (long long)(r * y + sp / r + 1) * (r * W + 2) + (r * x + sp % r + 1);
    const long long d_o = b * dst.bs + cg * dst.cs + (long long)(y + 1) * (W + 2) + (x + 1);
    ((uint4*)dst.hi)[d_o] = src.hi[so];
    if (dst.lo) ((uint4*)dst.lo)[d_o] = src.lo ? src.lo[so] : make_uint4(0, 0, 0, 0);
}

// Adjoint of esr_pack_nchw: act-layout gradient (interior (h+2pad)/down x (w+2pad)/down) -> fp32 NCHW gradient of the
// un-padded source, dst[b][c0+c][y][x] (+)= sum over padded positions that the replicate padding maps onto (y,x) of
//   down == 1 : g[pos]
//   down  > 1 : sum of bilinear tap weights * g[lr pos]   (the same taps the forward used)
// `accumulate` adds into dst instead of overwriting (latent: HR-resolution convs + LR-resolution convs both contribute).
__global__ void unpack_grad_kernel(DView G, float* __restrict__ dst, long long dbs, int C, int h, int w, int c0, int nc, int pad, int down,
                                   int accumulate, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, channel group, y, x) of the un-padded source
    if (idx >= total) return;
    const int x = (int)(idx % w);
    long long t = idx / w;
    const int y = (int)(t % h);
    t /= h;
    const int ncg = (nc + 7) >> 3;
    const int cg = (int)(t % ncg);
    const int b = (int)(t / ncg);
    const int hp = h + 2 * pad, wp = w + 2 * pad;
    // padded-frame positions mapping onto (y, x)
    const int y_lo = y == 0 ? 0 : y + pad, y_hi = y == h - 1 ? hp - 1 : y + pad;
    const int x_lo = x == 0 ? 0 : x + pad, x_hi = x == w - 1 ? wp - 1 : x + pad;
    const int Hd = hp / down, Wd = wp / down;
    const long long base = b * G.bs + cg * G.cs;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto add_at = [&](int yy, int xx, float wgt) {   // acc += wgt * the 8 gradient channels at act-layout interior pixel (yy, xx)
        float g8[8];
        unpack8(G.hi, G.lo, base + (long long)(yy + 1) * (Wd + 2) + (xx + 1), g8, G.fmt);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(wgt, g8[e], acc[e]);
    };
    if (down == 1) {
        for (int yy = y_lo; yy <= y_hi; ++yy)
            for (int xx = x_lo; xx <= x_hi; ++xx) add_at(yy, xx, 1.f);
    } else {
        // forward: lr(i) = (1-l)*hr(y0) + l*hr(y1), src = (i+0.5)*down-0.5, y0 = floor(src), y1 = min(y0+1, hp-1), l = src - y0
        // Even `down` (the x2 / x4 / x8 generators): src = down i + down / 2 - 0.5, so LR pixel i reads exactly the two central rows of its block with
        // weight 0.5 each (l = 0.5 exactly, never clamped) — row yy has ONE contributor (i = yy / down) when yy mod down is down/2 - 1 or down/2 and none
        // otherwise: the same (index, weight) the search below finds, without its ~20 candidate evaluations per pixel (the Z gradient calls of a configs[3]
        // iteration: 5.7 -> 3.6 ms of this kernel per iteration).
        const bool even = (down & 1) == 0;
        for (int yy = y_lo; yy <= y_hi; ++yy) {
            // LR pixels whose taps include padded-HR row yy: at most two
            float wy[2]; int iy[2]; int ny = 0;
            if (even) {
                const int i = yy / down, r = yy - i * down;
                if (r == down / 2 - 1 || r == down / 2) { wy[0] = 0.5f; iy[0] = i; ny = 1; }
                if (ny == 0) continue;
            } else
            for (int i = (yy - 1) / down - 1; i <= yy / down + 1; ++i) {
                if (i < 0 || i >= Hd) continue;
                const float src = fmaxf((i + 0.5f) * (float)down - 0.5f, 0.f);
                const int a0 = (int)src, a1 = a0 + (a0 < hp - 1 ? 1 : 0);
                const float l = src - (float)a0;
                float wgt = 0.f;
                if (a0 == yy) wgt += 1.f - l;
                if (a1 == yy) wgt += l;
                if (wgt != 0.f && ny < 2) { wy[ny] = wgt; iy[ny] = i; ++ny; }
            }
            for (int xx = x_lo; xx <= x_hi; ++xx) {
                float wx[2]; int ix[2]; int nx = 0;
                if (even) {
                    const int j = xx / down, r = xx - j * down;
                    if (r == down / 2 - 1 || r == down / 2) { wx[0] = 0.5f; ix[0] = j; nx = 1; }
                } else
                for (int j = (xx - 1) / down - 1; j <= xx / down + 1; ++j) {
                    if (j < 0 || j >= Wd) continue;
                    const float src = fmaxf((j + 0.5f) * (float)down - 0.5f, 0.f);
                    const int a0 = (int)src, a1 = a0 + (a0 < wp - 1 ? 1 : 0);
                    const float l = src - (float)a0;
                    float wgt = 0.f;
                    if (a0 == xx) wgt += 1.f - l;
                    if (a1 == xx) wgt += l;
                    if (wgt != 0.f && nx < 2) { wx[nx] = wgt; ix[nx] = j; ++nx; }
                }
                for (int p = 0; p < ny; ++p)
                    for (int q = 0; q < nx; ++q) add_at(iy[p], ix[q], wy[p] * wx[q]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cg * 8 + e;
        if (c >= nc) break;
        float* d = dst + b * dbs + ((long long)(c0 + c) * h + y) * w + x;
        *d = accumulate ? *d + acc[e] : acc[e];
    }
}

}  // namespace

// ---- fp16 gradient magnitude management (see include/esr_hip.h).  One 16-byte pixel vector per thread over the whole (H+2)x(W+2)
// plane of every (image, group): the zero border costs 3 % more threads and no index arithmetic.
namespace {

__global__ void grad_absmax_kernel(DView v, long long plane, long long total, uint32_t* __restrict__ slot) {
    // grid-stride over the vectors, wave shuffle + LDS reduction, ONE atomic per workgroup (one per wave measured 0.5 ms per call: tens of
    // thousands of atomics on one address)
    __shared__ uint32_t part[4];
    uint32_t m = 0;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long p = idx % plane, t = idx / plane;
        const int cg = (int)(t % v.ncg), b = (int)(t / v.ncg);
        const uint4 h = v.hi[b * v.bs + cg * v.cs + p];
        const uint32_t w[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t a0 = w[e] & 0x7FFFu, a1 = (w[e] >> 16) & 0x7FFFu;      // |fp16| bit patterns order like the magnitudes
            m = m > a0 ? m : a0;
            m = m > a1 ? m : a1;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(m, off);
        m = m > o ? m : o;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) m = m > part[i] ? m : part[i];
        if (m) atomicMax(slot, m);
    }
}

__global__ void grad_scale_kernel(DView src, DView dst, long long plane, long long total, const uint32_t* __restrict__ slot, int exp,
                                  const float* __restrict__ scale_in, const float* __restrict__ scale_den, float* __restrict__ scale_out) {
    float f;
    if (slot) {
        const uint32_t bits = *slot;
        int k = 0;
        if (bits) {
            const int E = (int)(bits >> 10);                        // biased exponent of max|hi| (0: subnormal)
            // normal: max in [2^(E-15), 2^(E-14));  subnormal with top mantissa bit p: max in [2^(p-24), 2^(p-23))
            k = E ? exp + 14 - E : exp + 23 - (31 - __clz((int)bits));
        }
        f = ldexpf(1.f, k);
        if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = *scale_in * f;
    } else {
        f = *scale_in / *scale_den;
    }
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long p = idx % plane, t = idx / plane;
    const int cg = (int)(t % dst.ncg), b = (int)(t / dst.ncg);
    const long long si = b * src.bs + cg * src.cs + p, di = b * dst.bs + cg * dst.cs + p;
    const uint4* const sp[2] = {src.hi, src.lo};
    uint4* const dp[2] = {(uint4*)dst.hi, (uint4*)dst.lo};
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        if (!sp[pl] || !dp[pl]) continue;
        const uint4 h = sp[pl][si];
        const uint32_t w[4] = {h.x, h.y, h.z, h.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2h(h2f(w[e] & 0xFFFF) * f) | (f2h(h2f(w[e] >> 16) * f) << 16);
        dp[pl][di] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace

extern "C" int esr_grad_absmax(const esr_act_view* v, int B, uint32_t* slot, esr_stream_t stream) {
    if (!v || !v->hi || !slot || B <= 0 || v->fmt != ESR_FMT_F16) return ESR_E_ARG;
    const long long plane = (long long)(v->H + 2) * (v->W + 2), total = plane * v->ncg * B;
    ESR_CLEAR_ERR();
    const long long nblk = (total + 255) / 256;
    hipLaunchKernelGGL(grad_absmax_kernel, dim3((unsigned)(nblk < 1024 ? nblk : 1024)), dim3(256), 0, (hipStream_t)stream, to_dview(*v), plane, total, slot);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_grad_scale(const esr_act_view* src, const esr_act_view* dst, int B, const uint32_t* slot, int exp, const float* scale_in,
                              const float* scale_den, float* scale_out, esr_stream_t stream) {
    if (!src || !dst || !src->hi || !dst->hi || B <= 0 || src->fmt != ESR_FMT_F16 || dst->fmt != ESR_FMT_F16) return ESR_E_ARG;
    if (src->H != dst->H || src->W != dst->W || src->ncg < dst->ncg) return ESR_E_ARG;
    if (slot ? (scale_out && !scale_in) : (!scale_in || !scale_den)) return ESR_E_ARG;
    const long long plane = (long long)(dst->H + 2) * (dst->W + 2), total = plane * dst->ncg * B;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(grad_scale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, to_dview(*src), to_dview(*dst), plane,
                       total, slot, exp, scale_in, scale_den, scale_out);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_act_combine(const esr_act_view* A, float alpha, const esr_act_view* Bv, float beta, int s, const esr_act_view* mask,
                               float mask_slope, const esr_act_view* out, int B, esr_stream_t stream) {
    if (!out || !out->hi || B <= 0 || s < 1) return ESR_E_ARG;
    esr_act_view none{};
    const esr_act_view& a = A ? *A : none;
    const esr_act_view& bv = Bv ? *Bv : none;
    const esr_act_view& m = mask ? *mask : none;
    if (a.hi && (a.H != out->H || a.W != out->W || a.ncg < out->ncg)) return ESR_E_ARG;
    if (bv.hi && (bv.H != out->H * s || bv.W != out->W * s || bv.ncg < out->ncg)) return ESR_E_ARG;
    if (m.hi && (m.H != out->H || m.W != out->W || m.ncg < out->ncg)) return ESR_E_ARG;
    const long long total = (long long)B * out->ncg * out->H * out->W;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(act_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, to_dview(a), alpha, to_dview(bv),
                       beta, s, to_dview(m), mask_slope, to_dview(*out), out->H, out->W, total);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_pixel_unshuffle(const esr_act_view* src, int r, const esr_act_view* dst, int B, esr_stream_t stream) {
    if (!src || !src->hi || !dst || !dst->hi || B <= 0 || r < 2) return ESR_E_ARG;
    if (dst->ncg != src->ncg * r * r || src->H != r * dst->H || src->W != r * dst->W || src->fmt != dst->fmt) return ESR_E_ARG;
    const long long total = (long long)B * dst->ncg * dst->H * dst->W;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(pixel_unshuffle_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, to_dview(*src), r, to_dview(*dst),
                       dst->H, dst->W, total);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_unpack_grad_nchw(const esr_act_view* G, float* dst, int64_t dst_batch_stride, int B, int C, int h, int w, int c0, int nc, int pad,
                                    int down, int accumulate, esr_stream_t stream) {
    if (!G || !G->hi || !dst || B <= 0 || nc <= 0 || c0 < 0 || c0 + nc > C || pad < 0 || down < 1) return ESR_E_ARG;
    if ((h + 2 * pad) % down || (w + 2 * pad) % down) return ESR_E_ARG;
    if (G->H != (h + 2 * pad) / down || G->W != (w + 2 * pad) / down || G->ncg * 8 < nc) return ESR_E_ARG;
    const long long total = (long long)B * ((nc + 7) / 8) * h * w;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(unpack_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, to_dview(*G), dst,
                       (long long)(dst_batch_stride ? dst_batch_stride : (int64_t)C * h * w), C, h, w, c0, nc, pad, down, accumulate, total);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight (and bias) gradient of conv3x3:   dW[co][ci][dy][dx] = alpha * sum_{b,y,x} dY[b,co,y,x] * X[b,ci,y+dy-1,x+dx-1]
// (autograd of nn.Conv2d in the reference, codes/models/modules/block.py:141-142).
//
// A GEMM with M = 32 output channels, N = 32 input channels and K = PIXELS on the bf16 MFMA pipe (v_mfma_f32_32x32x16_bf16, 16
// pixels per instruction) with the forward's split operands: dYlo*Xhi + dYhi*Xlo + dYhi*Xhi, fp32 accumulate.  Both operands
// store 8 CHANNELS contiguously per pixel, so "8 consecutive pixels of one channel per lane" is a transposition — done by gfx950's
// transposing LDS read (ds_read_b64_tr_b16) straight out of the pixel-major tiles that LDS-DMA copies verbatim from HBM.
//   workgroup  = (32-input-channel tile, 32-output-channel tile) x a slice of the 8x32-pixel image tiles; 4 waves; all 9 taps
//   per tile   : LDS-DMA of dY[8 x 32] and the haloed X[10 x 34] (4 channel groups each, hi + lo) into one of two LDS stages,
//                issued one tile ahead of the MFMAs; waves split the tile's rows (split K), 2 rows x 2 K-steps x 9 taps x 3 terms
//   end        : the 4 waves' accumulators are summed through LDS, written to the caller's workspace, and wgrad_reduce_kernel folds
//                the slices into dW / db (float atomics from ~10^3 workgroups onto ~10^4 addresses measured 5x slower than the MFMAs)
namespace {

constexpr int WG_TH = 8, WG_TW = 32;       // tile: 256 pixels

struct WgradArgs {
    DView dy, x, xlat;
    int lat, ups, cout, cin_main, cin_total, B, H, W;
    int Wx_p;                  // padded row pitch of the x source (W/ups + 2)
    int tiles_x, tiles_y, nslices, ncit_main, ncit, mt, ngroups;
    float alpha;
    float* dw;
    float* db;
    float* ws;                 // partial sums: [group][slice][9*1024], then the bias partials [cout tile][slice][32]
    int tapmode;               // 1: esr_wgrad_desc.tap_masks name the space-to-depth pattern (S2D_TAPS below); 0: all taps
    int latk;                  // 1: the latent tile (lat <= 3 channels) runs the one-MFMA-tile form (wgrad_body<..., LATK>)
    int shape;                 // pixel tile of this layer: 0 = 8 rows x 32 columns, 1 = 16 x 16, 2 = 32 x 8 (narrow maps; S2D kernels only)
};
#ifdef ESR_TRACE
// debug build only (make trace): per-workgroup phase stamps of the weight-gradient kernels, 64 slots per workgroup — [0] HW_ID, [1] stamps used,
// [2..] s_memtime at (tile start, copies issued, copies landed, barrier passed, MFMAs done) of the first 11 tiles, [62] / [63] wall clock
// (100 MHz) at start / end; set with esr_debug_trace_wgrad(), read by tools/experiments/trace_wgrad.py
__device__ unsigned long long* g_wtrace = nullptr;
#endif

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ void glds16w(const uint4* src, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_dst) : "memory");
}

// the same copy, source = wave-uniform base (SGPR pair) + per-lane 32-bit byte offset: no 64-bit per-lane address arithmetic
__device__ __forceinline__ void glds16ws(const uint4* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// LDS planes are sized so that consecutive channel groups start 64 bytes apart modulo 256: the four 64-byte runs a half-wave
// touches in one ds_read_b64_tr_b16 (2 groups x 4 pixels, for two 16-lane groups) then fall into four different 16-bank windows
constexpr int XP = (WG_TH + 2) * (WG_TW + 2);        // 340 haloed x pixels per plane; 340 % 16 == 4
constexpr int YP = WG_TH * WG_TW, YPP = 260;         // 256 dy pixels per plane, padded: 260 % 16 == 4
constexpr int XSLOTS = (XP + 63) / 64, YSLOTS = YP / 64;            // 64-pixel DMA slots per plane: 6, 4
constexpr int WG_X_BYTES = 4 * XP * 16, WG_Y_BYTES = 4 * YPP * 16;  // one plane set (4 groups)
static_assert(XP % 16 == 4 && YPP % 16 == 4, "bank spreading of the transposing reads");

// One MFMA operand fragment whose K axis runs over PIXELS, out of the pixel-major [pixel][8 channels] LDS image.  Inside a 16-lane
// group, lane i points at 4 consecutive channels (8 bytes) of pixel (i >> 2) in channel quad (i & 3) of the group's 16 channels; the
// hardware hands lane l the 4 pixels of channel l (profiles/microbench/tr_b16_probe.hip).  Two reads make the 8 K-values of a lane.
__device__ __forceinline__ uint4 frag_tr(const unsigned char* p) {
    typedef __attribute__((address_space(3))) s16x4* lptr;
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(size_t)(p));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(size_t)(p + 64));
    const uint2 u0 = __builtin_bit_cast(uint2, r0), u1 = __builtin_bit_cast(uint2, r1);
    return make_uint4(u0.x, u0.y, u1.x, u1.y);
}

// FMT: element format of both operands (ESR_FMT_BF16 / ESR_FMT_F16) — only the MFMA instruction and the constant 1.0 differ
template <int FMT>
__device__ __forceinline__ f32x16_t mfma_e(uint4 a, uint4 b, f32x16_t c) {
    if constexpr (FMT == ESR_FMT_F16) {
        typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
}

// TM (compile time): the taps whose weight gradient is wanted (structurally zero blocks of the weights are skipped: the unrolled tap loop
// drops their reads and MFMAs; their accumulators stay zero)
// SH (compile time): the 256-pixel tile's shape, 8 x 32 (0), 16 x 16 (1) or 32 x 8 (2) — a 16x16 / 8x8 / 4x4 feature map (the critic's deep
// layers; the small ones stacked into one tall image) fills 100 / 80 / 40 % of its tiles instead of 50 / 20 / 10 %.  The dY plane stays the
// tile's pixels in row-major order (K step j = pixels 16 j .. 16 j + 15), the haloed X tile has pitch TW + 2: a K step is half a row, a row,
// or two rows of it.
// LATK (compile time; lat <= 3, 8 x 32 tiles): the workgroup owns the LATENT input tile.  Its <= 3 channels x 9 taps are <= 27 columns: they are
// laid out as the N axis of ONE MFMA tile — lane n gathers channel n % 3 at tap n / 3 for its 8 pixels (ds_read_u16) — so a K step is 1 (3 in
// split) MFMAs instead of 9 (27), and only wave 0 copies input (the one group there is).
template <int NPL, int NST, int FMT, int TM = 0x1FF, int SH = 0, bool LATK = false>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int group, const int slice, unsigned char* const smem) {
    static_assert(!LATK || (SH == 0 && TM == 0x1FF), "the latent-tile form exists for the plain 8 x 32 tiles");
    constexpr int TW = 32 >> SH, TH = 8 << SH, LGW = 5 - SH;
    constexpr int XPS = (TH + 2) * (TW + 2);                     // haloed pixels of this shape (<= XP, the plane stride)
    static_assert(XPS <= XP && TW * TH == YP, "tile shapes share the LDS plane sizes");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cit = group / a.mt, cot = group % a.mt;            // input-channel tile, output-channel tile
    const bool lat_tile = cit >= a.ncit_main;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    constexpr int STAGE = NPL * (WG_X_BYTES + WG_Y_BYTES);       // [X hi | X lo | dY hi | dY lo], 4 group planes each
    constexpr int NX = NPL * XSLOTS, NY = NPL * YSLOTS;          // DMA instructions per wave per tile: 12 + 8 (6 + 4 without lo)
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    const DView& xv = lat_tile ? a.xlat : a.x;

    f32x16_t acc[9], accb;
#pragma unroll
    for (int i = 0; i < 16; ++i) accb[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const bool do_bias = (cit == 0) && a.db;                     // uniform: the first input tile's workgroups also reduce dY itself
    constexpr uint32_t ONE2 = FMT == ESR_FMT_F16 ? 0x3C003C00u : 0x3F803F80u;
    const uint4 ones = make_uint4(ONE2, ONE2, ONE2, ONE2);   // 1.0 x 8 in the operand format

    // this lane's source address inside a fragment's 16-lane group (see frag_tr)
    const int li = lane & 15, grp16 = lane >> 4;
    const int rb2 = (grp16 & 1) * 2 + ((li & 3) >> 1);           // channel group (of the 32-channel tile's 4) this lane points into
    const int kb = (grp16 >> 1) * 8 + (li >> 2);                 // pixel inside the 16-pixel K step
    const int frag_off = kb * 16 + (li & 1) * 8;                 // + low / high 4 channels of the 16-byte vector
    // (32 x 8 tiles: the second half of a K step is the next row of the haloed tile, TW + 2 - 8 = 2 pixels further than contiguous)
    const int xs_off = rb2 * XP * 16 + frag_off + (SH == 2 ? (kb >> 3) * 32 : 0);
    const int ys_off = NPL * WG_X_BYTES + rb2 * YPP * 16 + frag_off;

    // All DMA of one tile.  Wave w copies channel group w of both operands (hi and lo planes): 6 + 4 slots of 64 pixel vectors per
    // plane, i.e. exactly NX + NY instructions per wave, so that the consumer can wait with a constant vmcnt while the next tile's
    // copies stay in flight.  The per-lane source offsets are computed once per slot and shared by the hi and lo planes.
    static_assert(NX == NPL * XSLOTS && NY == NPL * YSLOTS, "one channel group per wave");
    const int xcg = lat_tile ? wave : cit * 4 + wave;            // this wave's x group (may not exist: zero weights, read group 0's border)
    const bool xhave = xcg < xv.ncg;
    const int ycg = cot * 4 + wave;
    const bool yhave = ycg < a.dy.ncg;
#define ESR_WG_ISSUE(TILE, ST)                                                                                                   \
    do {                                                                                                                         \
        const int tx_ = (TILE) % a.tiles_x;                                                                                      \
        const int r1_ = (TILE) / a.tiles_x;                                                                                      \
        const int ty_ = r1_ % a.tiles_y;                                                                                         \
        const int b_ = r1_ / a.tiles_y;                                                                                          \
        const int x0_ = tx_ * TW, y0_ = ty_ * TH;                                                                                \
        const uint4* const xh_ = xv.hi + b_ * xv.bs + (xhave ? xcg : 0) * xv.cs;                                                 \
        const uint4* const xl_ = NPL == 2 ? xv.lo + b_ * xv.bs + (xhave ? xcg : 0) * xv.cs : nullptr;                            \
        const unsigned xd_ = (ST) + wave * XP * 16;                                                                              \
        _Pragma("unroll") for (int sl = 0; sl < XSLOTS; ++sl) {                                                                  \
            if (LATK && NST == 1 && wave > 0) break;                /* the latent tile is one group: nobody reads planes 1-3 */    \
            const int p = sl * 64 + lane;                                                                                        \
            const int rr = p / (TW + 2), cc = p - rr * (TW + 2);                                                                 \
            const int Yp = y0_ + rr, Xp = x0_ + cc;                 /* padded output-resolution coords of the haloed tile */     \
            int sy = Yp, sx = Xp;                                                                                                \
            if (a.ups == 2) { sy = (Yp + 1) >> 1; sx = (Xp + 1) >> 1; }                                                          \
            else if (a.ups > 2) { sy = (Yp - 1 + a.ups) / a.ups; sx = (Xp - 1 + a.ups) / a.ups; }                                \
            const int off = (xhave && Yp < a.H + 2 && Xp < a.W + 2) ? sy * a.Wx_p + sx : 0;   /* 0: the zero border vector */    \
            if (p < XPS) {                                                                                                       \
                glds16w(xh_ + off, xd_ + sl * 1024);                                                                             \
                if (NPL == 2) glds16w(xl_ + off, xd_ + WG_X_BYTES + sl * 1024);                                                  \
            }                                                                                                                    \
        }                                                                                                                        \
        const uint4* const yh_ = a.dy.hi + b_ * a.dy.bs + (yhave ? ycg : 0) * a.dy.cs;                                           \
        const uint4* const yl_ = NPL == 2 ? a.dy.lo + b_ * a.dy.bs + (yhave ? ycg : 0) * a.dy.cs : nullptr;                      \
        const unsigned yd_ = (ST) + NPL * WG_X_BYTES + wave * YPP * 16;                                                          \
        _Pragma("unroll") for (int sl = 0; sl < YSLOTS; ++sl) {                                                                  \
            const int p = sl * 64 + lane;                                                                                        \
            const int Y = y0_ + (p >> LGW), X = x0_ + (p & (TW - 1));                                                            \
            const int off = (yhave && Y < a.H && X < a.W) ? (Y + 1) * (a.W + 2) + (X + 1) : 0;                                   \
            glds16w(yh_ + off, yd_ + sl * 1024);                                                                                 \
            if (NPL == 2) glds16w(yl_ + off, yd_ + WG_Y_BYTES + sl * 1024);                                                      \
        }                                                                                                                        \
    } while (0)

    // The same copies for the common case (no upsampled source): a slot's offsets inside the tile are tile-independent — computed ONCE per
    // workgroup (relative offset + row / column inside the haloed tile) — so that a tile costs per slot two compares against the tile's distance
    // to the image edge, one add and one select, and the copy takes a uniform base + a 32-bit lane offset.  (The general macro above spends
    // ~12 VALU per slot and a 64-bit add per plane: 4.8 VALU per MFMA in this kernel's counters, profiles/r04_wgrad_pmc.json.)
    const bool fast_issue = NPL == 1 && a.ups == 1;      // (one-plane operands: the hi+lo kernels sit at the 256-register limit and would spill)
    // only the LAST tile column / row of an image can be partial: per slot two precomputed bits — "inside the image when the tile is in the last
    // column" / "... last row" — packed into one register (slot j: bits 2j, 2j + 1; X slots first, then the dY slots)
    int xrel[XSLOTS], yrel[YSLOTS];
    unsigned vbits = 0;
    {
        const int x_last = (a.tiles_x - 1) * TW, y_last = (a.tiles_y - 1) * TH;
#pragma unroll
        for (int sl = 0; sl < XSLOTS; ++sl) {
            const int p = sl * 64 + lane;
            const int rr = p / (TW + 2), cc = p - rr * (TW + 2);
            xrel[sl] = rr * a.Wx_p + cc;
            if (x_last + cc < a.W + 2) vbits |= 1u << (2 * sl);
            if (y_last + rr < a.H + 2) vbits |= 2u << (2 * sl);
        }
#pragma unroll
        for (int sl = 0; sl < YSLOTS; ++sl) {
            const int p = sl * 64 + lane;
            const int rr = p >> LGW, cc = p & (TW - 1);
            yrel[sl] = rr * (a.W + 2) + cc;
            if (x_last + cc < a.W) vbits |= 1u << (2 * (XSLOTS + sl));
            if (y_last + rr < a.H) vbits |= 2u << (2 * (XSLOTS + sl));
        }
    }
    static_assert(2 * (XSLOTS + YSLOTS) <= 32, "validity bits of all slots in one register");
    // (tx, ty, b) of the tile the fast path issues next: decomposed by division ONCE, then stepped by the decomposition of the workgroup's stride
    // (nslices tiles) with two carries — the per-tile divisions were ~70 of a wave's ~500 instructions per tile
    int it_tx, it_ty, it_b, d_tx, d_ty, d_b;
    {
        const int r1 = slice / a.tiles_x, per = a.tiles_x * a.tiles_y, rem = a.nslices % per;
        it_tx = slice - r1 * a.tiles_x; it_b = r1 / a.tiles_y; it_ty = r1 - it_b * a.tiles_y;
        d_b = a.nslices / per; d_ty = rem / a.tiles_x; d_tx = rem - d_ty * a.tiles_x;
    }
    auto issue_step = [&]() {
        it_tx += d_tx;
        if (it_tx >= a.tiles_x) { it_tx -= a.tiles_x; ++it_ty; }
        it_ty += d_ty;
        if (it_ty >= a.tiles_y) { it_ty -= a.tiles_y; ++it_b; }
        it_b += d_b;
    };
    auto issue_fast = [&](const unsigned st_) {
        const int tx_ = it_tx, ty_ = it_ty, b_ = it_b;
        const int x0_ = tx_ * TW, y0_ = ty_ * TH;
        // (uniform) a tile that is not in the last column / row is whole: its bits are forced on; an operand group that does not exist reads
        // the zero border vector for every slot
        unsigned um = (tx_ == a.tiles_x - 1 ? 0u : 0x55555555u) | (ty_ == a.tiles_y - 1 ? 0u : 0xAAAAAAAAu);
        unsigned keep = 0xFFFFFFFFu;
        if (!xhave) keep &= ~((1u << (2 * XSLOTS)) - 1u);
        if (!yhave) keep &= (1u << (2 * XSLOTS)) - 1u;
        const unsigned m = (vbits | um) & keep;
        const uint4* const xh_ = xv.hi + b_ * xv.bs + (xhave ? xcg : 0) * xv.cs;
        const uint4* const xl_ = NPL == 2 ? xv.lo + b_ * xv.bs + (xhave ? xcg : 0) * xv.cs : nullptr;
        const unsigned xd_ = st_ + wave * XP * 16;
        const int xb_ = y0_ * a.Wx_p + x0_;
#pragma unroll
        for (int sl = 0; sl < XSLOTS; ++sl) {
            if (LATK && NST == 1 && wave > 0) break;
            const unsigned vo = ((m >> (2 * sl)) & 3u) == 3u ? (unsigned)(xrel[sl] + xb_) * 16u : 0u;
            if (sl * 64 + lane < XPS) {
                glds16ws(xh_, vo, xd_ + sl * 1024);
                if (NPL == 2) glds16ws(xl_, vo, xd_ + WG_X_BYTES + sl * 1024);
            }
        }
        const uint4* const yh_ = a.dy.hi + b_ * a.dy.bs + (yhave ? ycg : 0) * a.dy.cs;
        const uint4* const yl_ = NPL == 2 ? a.dy.lo + b_ * a.dy.bs + (yhave ? ycg : 0) * a.dy.cs : nullptr;
        const unsigned yd_ = st_ + NPL * WG_X_BYTES + wave * YPP * 16;
        const int yb_ = (y0_ + 1) * (a.W + 2) + x0_ + 1;
#pragma unroll
        for (int sl = 0; sl < YSLOTS; ++sl) {
            const unsigned vo = ((m >> (2 * (XSLOTS + sl))) & 3u) == 3u ? (unsigned)(yrel[sl] + yb_) * 16u : 0u;
            glds16ws(yh_, vo, yd_ + sl * 1024);
            if (NPL == 2) glds16ws(yl_, vo, yd_ + WG_Y_BYTES + sl * 1024);
        }
    };
// (the fast path issues the tiles of this workgroup IN ORDER: every call is followed by issue_step())
#define ESR_WG_ISSUE2(TILE, ST) do { if (fast_issue) { issue_fast((ST)); issue_step(); } else ESR_WG_ISSUE((TILE), (ST)); } while (0)

    // NST == 2: two LDS stages, the next tile's copies in flight under the MFMAs, one workgroup per CU.
    // NST == 1: one stage, two workgroups per CU cover each other's DMA waits (same trade as the conv kernel; selected by the host).
    int tile = slice;
    int cur = 0;
#ifdef ESR_TRACE
    unsigned long long* const tr = g_wtrace ? g_wtrace + (size_t)blockIdx.x * 64 : nullptr;
    int tslot = 2;
    if (tr && tid == 0) { tr[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4); tr[62] = wall_clock64(); }
#define ESR_WTR() do { if (tr && tid == 0 && tslot < 57) tr[tslot++] = __builtin_readcyclecounter(); } while (0)
#else
#define ESR_WTR() do { } while (0)
#endif
    if (NST == 2 && tile < ntiles) ESR_WG_ISSUE2(tile, lds0);
    for (; tile < ntiles; tile += a.nslices) {
        const int nxt = tile + a.nslices;
        ESR_WTR();
        if (NST == 1) {
            ESR_WG_ISSUE2(tile, lds0);
            ESR_WTR();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ESR_WTR();
        } else if (nxt < ntiles) {
            ESR_WG_ISSUE2(nxt, lds0 + (cur ^ 1) * STAGE);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NX + NY) : "memory");     // everything but the copies just issued
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        ESR_WTR();
        const unsigned char* const sx = smem + cur * STAGE + xs_off;
        const unsigned char* const sy = smem + cur * STAGE + ys_off;
        // ---- MFMAs: wave handles rows wave, wave+4; a row of 32 pixels = two K steps of 16
#pragma unroll
        for (int rq = 0; rq < WG_TH / 4; ++rq) {
            const int rr = wave + rq * 4;
#pragma unroll
            for (int ks = 0; ks < WG_TW / 16; ++ks) {
                uint4 fa[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) fa[pl] = frag_tr(sy + pl * WG_Y_BYTES + (rr * WG_TW + ks * 16) * 16);      // K step j = 2 rr + ks
                if constexpr (LATK) {
                    if (do_bias) {                                // (a main input of <= 3 channels run in this form: its workgroups own the bias too)
                        accb = mfma_e<FMT>(fa[0], ones, accb);
                        if (NPL == 2) accb = mfma_e<FMT>(fa[NPL - 1], ones, accb);
                    }
                    // B fragment by gather: this lane's column n = lane & 31 -> (tap n / 3, channel n % 3); its 8 K values are pixels
                    // kblk * 8 .. + 7 of the K step at that tap's shift
                    const int n = lane & 31, tcol = n < 27 ? n / 3 : 8, ccol = n < 27 ? n % 3 : 0;
                    const unsigned char* const gx = smem + cur * STAGE + ((rr + tcol / 3) * (TW + 2) + ks * 16 + (lane >> 5) * 8 + tcol % 3) * 16 + ccol * 2;
                    uint4 fbk[NPL];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        const unsigned char* const g0 = gx + pl * WG_X_BYTES;
                        uint32_t w[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            w[q] = (uint32_t)(*(const unsigned short*)(g0 + (2 * q) * 16)) | ((uint32_t)(*(const unsigned short*)(g0 + (2 * q + 1) * 16)) << 16);
                        fbk[pl] = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                    if (NPL == 2) {
                        acc[0] = mfma_e<FMT>(fa[1], fbk[0], acc[0]);
                        acc[0] = mfma_e<FMT>(fa[0], fbk[NPL - 1], acc[0]);
                    }
                    acc[0] = mfma_e<FMT>(fa[0], fbk[0], acc[0]);
                    continue;
                }
                if (do_bias) {                                    // dY x ones: every column of the tile holds sum_k dY[row][k]
                    accb = mfma_e<FMT>(fa[0], ones, accb);
                    if (NPL == 2) accb = mfma_e<FMT>(fa[NPL - 1], ones, accb);
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (!((TM >> t) & 1)) continue;             // folds at compile time (t is an unrolled constant)
                    uint4 fb[NPL];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        fb[pl] = frag_tr(sx + pl * WG_X_BYTES + (SH == 0 ? ((rr + t / 3) * (TW + 2) + ks * 16 + t % 3)
                                                                       : SH == 1 ? ((2 * rr + ks + t / 3) * (TW + 2) + t % 3)
                                                                                 : ((4 * rr + 2 * ks + t / 3) * (TW + 2) + t % 3)) * 16);
                    if (NPL == 2) {
                        acc[t] = mfma_e<FMT>(fa[1], fb[0], acc[t]);
                        acc[t] = mfma_e<FMT>(fa[0], fb[NPL - 1], acc[t]);
                    }
                    acc[t] = mfma_e<FMT>(fa[0], fb[0], acc[t]);
                }
            }
        }
        ESR_WTR();
        __syncthreads();
        if (NST == 2) cur ^= 1;
    }
#ifdef ESR_TRACE
    if (tr && tid == 0) { tr[1] = tslot; tr[63] = wall_clock64(); }
#endif
    // ---- reduce the 4 waves through LDS (two passes of at most 5 taps: 4 x 5 x 4 KiB = 80 KiB).  A workgroup that owns its
    // (input tile, output tile) alone (nslices == 1) adds straight into dW / db; otherwise its partial sums go to the workspace.
    float* const red = (float*)smem;                             // [wave][tap in pass][16][64]
    const bool direct = a.nslices == 1;
    float* const wsp = direct ? nullptr : a.ws + ((size_t)group * a.nslices + slice) * (9 * 1024);
    if constexpr (LATK) {
        // one accumulator tile per wave: column n = (tap n / 3, channel n % 3); summed over the waves it is scattered to where the nine-tap form
        // puts the same numbers (dW, or this slice's partial tiles — wgrad_reduce_body reads only the channels < lat of a latent tile)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[0][i];
        __syncthreads();
        for (int e = tid; e < 1024; e += 256) {
            float v = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) v += red[w4 * 1024 + e];
            const int ln = e & 63, i = e >> 6, n = ln & 31, half = ln >> 5;
            if (n >= 27) continue;
            const int t = n / 3, c = n % 3;
            if (c >= a.lat) continue;
            const int co = cot * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
            if (direct) { if (co < a.cout) a.dw[((long long)co * a.cin_total + c) * 9 + t] += a.alpha * v; }
            else wsp[t * 1024 + i * 64 + half * 32 + c] = v;
        }
        if (do_bias) {                                            // uniform; column 0 of the dY x ones tile, as below
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = accb[i];
            __syncthreads();
            if (tid < 32) {
                const int row = tid, i = (row & 3) + 4 * (row >> 3), ln = ((row >> 2) & 1) * 32;
                float v = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) v += red[(w4 * 16 + i) * 64 + ln];
                if (direct) { if (cot * 32 + row < a.cout) a.db[cot * 32 + row] += a.alpha * v; }
                else a.ws[(size_t)a.ngroups * a.nslices * (9 * 1024) + ((size_t)cot * a.nslices + slice) * 32 + row] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int t0 = pass * 5, nt = pass == 0 ? 5 : 4;
#pragma unroll
        for (int t = 0; t < 5; ++t)
            if (t < nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) red[((wave * 5 + t) * 16 + i) * 64 + lane] = acc[t0 + t][i];
        if (pass == 1 && do_bias)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[((wave * 5 + 4) * 16 + i) * 64 + lane] = accb[i];
        __syncthreads();
        for (int e = tid; e < nt * 1024; e += 256) {
            float v = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) v += red[w4 * 5 * 1024 + e];
            if (direct) {
                const int ln = e & 63, i = (e >> 6) & 15, t = t0 + (e >> 10);
                const int co = cot * 32 + (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5);     // D row
                const int c = ln & 31;                                                  // D col = channel inside the tile
                int ci = -1;
                if (lat_tile) { if (c < a.lat) ci = c; }
                else if (cit * 32 + c < a.cin_main) ci = a.lat + cit * 32 + c;
                if (co < a.cout && ci >= 0) a.dw[((long long)co * a.cin_total + ci) * 9 + t] += a.alpha * v;
            } else {
                wsp[t0 * 1024 + e] = v;
            }
        }
        if (pass == 1 && do_bias && tid < 32) {
            // column 0 of the dY x ones tile (lanes 0 and 32): row (i&3) + 8*(i>>2) + 4*(lane>>5)
            const int row = tid, i = (row & 3) + 4 * (row >> 3), ln = ((row >> 2) & 1) * 32;
            float v = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) v += red[((w4 * 5 + 4) * 16 + i) * 64 + ln];
            if (direct) { if (cot * 32 + row < a.cout) a.db[cot * 32 + row] += a.alpha * v; }
            else a.ws[(size_t)a.ngroups * a.nslices * (9 * 1024) + ((size_t)cot * a.nslices + slice) * 32 + row] = v;
        }
        __syncthreads();
    }
}

#undef ESR_WG_ISSUE
#undef ESR_WG_ISSUE2
#undef ESR_WTR

// s2d (WgradArgs.tapmode == 1): the layer is a stride-2 conv run as a 3x3 conv over the space-to-depth input (esr_hip/critic.py): main input
// tile i (one 32-channel quad of one parity) has non-zero weights only at the taps S2D_TAPS[i & 3] — five copies of the body, picked by a
// uniform switch, each with its tap set as a compile-time constant
constexpr int S2D_TAPS[4] = {432, 216, 54, 27};
template <int NPL, int NST, int FMT, int SH>
__device__ __forceinline__ void wgrad_dispatch_taps(const WgradArgs& a, const int group, const int slice, unsigned char* const smem) {
    const int cit = group / a.mt;
    if (a.tapmode == 1 && cit < a.ncit_main) {
        switch (cit & 3) {
            case 0: wgrad_body<NPL, NST, FMT, S2D_TAPS[0], SH>(a, group, slice, smem); return;
            case 1: wgrad_body<NPL, NST, FMT, S2D_TAPS[1], SH>(a, group, slice, smem); return;
            case 2: wgrad_body<NPL, NST, FMT, S2D_TAPS[2], SH>(a, group, slice, smem); return;
            default: wgrad_body<NPL, NST, FMT, S2D_TAPS[3], SH>(a, group, slice, smem); return;
        }
    }
    wgrad_body<NPL, NST, FMT, 0x1FF, SH>(a, group, slice, smem);
}
template <int NPL, int NST, int FMT, bool S2D>
__device__ __forceinline__ void wgrad_dispatch(const WgradArgs& a, const int group, const int slice, unsigned char* const smem) {
    if constexpr (S2D) {
        if (a.latk && group / a.mt >= a.ncit_main) return wgrad_body<NPL, NST, FMT, 0x1FF, 0, true>(a, group, slice, smem);
        if (a.shape == 1) return wgrad_dispatch_taps<NPL, NST, FMT, 1>(a, group, slice, smem);
        if (a.shape == 2) return wgrad_dispatch_taps<NPL, NST, FMT, 2>(a, group, slice, smem);
        return wgrad_dispatch_taps<NPL, NST, FMT, 0>(a, group, slice, smem);
    }
    if (a.latk && group / a.mt >= a.ncit_main) return wgrad_body<NPL, NST, FMT, 0x1FF, 0, true>(a, group, slice, smem);
    wgrad_body<NPL, NST, FMT>(a, group, slice, smem);
}

template <int NPL, int NST, int FMT, bool S2D = false>
__global__ __launch_bounds__(256, (NST == 1 || NPL == 1) ? 2 : 1) void conv3x3_wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    wgrad_dispatch<NPL, NST, FMT, S2D>(a, blockIdx.x / a.nslices, blockIdx.x % a.nslices, smem);
}

// Many layers in one launch (the whole backward pass of a generator): workgroup b serves table[map[b].x] as (group map[b].y,
// slice map[b].z).  With hundreds of layers there are enough (layer, input tile, output tile) triples to fill the chip without
// splitting the pixel sum, so each workgroup streams ALL tiles of its layer and owns its 32x32x9 block of dW.
// SIDE: the same kernel for a launch that runs on a SECOND stream under another stream's kernels (the backward's data-gradient chain: ~350 small
// launches of one workgroup per CU that leave most of every CU idle).  Its waves are made to hold more than half of a SIMD's register file (the
// clobber pins the accumulator registers up to a196: 273 registers per wave, allocated as 280), so the hardware can never place a second one on a SIMD: ONE such
// workgroup per CU, whatever LDS is free — the other stream's workgroups (<= 160 registers, <= 80 KB of LDS next to this one's 80) always find
// room on every CU instead of queueing behind two weight-gradient workgroups that live for a millisecond.
template <int NPL, int NST, int FMT, bool S2D = false, bool SIDE = false>
__global__ __launch_bounds__(256, SIDE ? 1 : ((NST == 1 || NPL == 1) ? 2 : 1)) void conv3x3_wgrad_batch_kernel(const WgradArgs* __restrict__ table, const int4* __restrict__ map) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int4 m = map[blockIdx.x];
    if (m.x < 0) return;                                         // padding of the XCD-aware order (uniform)
    const int e = __builtin_amdgcn_readfirstlane(m.x), group = __builtin_amdgcn_readfirstlane(m.y), slice = __builtin_amdgcn_readfirstlane(m.z);
    const WgradArgs a = table[e];
    if constexpr (SIDE) asm volatile("" ::: "a196");          // 76 VGPRs + 197 AGPRs = 273 > 256 (a180 is the least that caps: 257; a210 measures the same)
    wgrad_dispatch<NPL, NST, FMT, S2D>(a, group, slice, smem);
}

// dW += alpha * sum over slices of the partial tiles; one thread per (group, accumulator element), plus cout threads for the bias
__device__ __forceinline__ void wgrad_reduce_body(const WgradArgs& a, const long long idx) {
    const int per = 9 * 1024;
    const long long nmain = (long long)a.ngroups * per;
    if (idx < nmain) {
        const int group = (int)(idx / per), e = (int)(idx % per);
        const float* p = a.ws + (size_t)group * a.nslices * per + e;
        float v = 0.f;
        for (int s = 0; s < a.nslices; ++s) v += p[(size_t)s * per];
        const int cit = group / a.mt, cot = group % a.mt;
        const bool lat_tile = cit >= a.ncit_main;
        const int ln = e & 63, i = (e >> 6) & 15, t = e >> 10;
        const int co = cot * 32 + (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5);     // D row
        const int c = ln & 31;                                                  // D col = channel inside the tile
        int ci = -1;
        if (lat_tile) { if (c < a.lat) ci = c; }
        else if (cit * 32 + c < a.cin_main) ci = a.lat + cit * 32 + c;
        if (co < a.cout && ci >= 0) a.dw[((long long)co * a.cin_total + ci) * 9 + t] += a.alpha * v;
    } else if (a.db && idx < nmain + a.mt * 32) {
        const int e = (int)(idx - nmain);                                       // output channel
        const float* p = a.ws + (size_t)nmain * a.nslices + (size_t)(e >> 5) * a.nslices * 32 + (e & 31);
        float v = 0.f;
        for (int s = 0; s < a.nslices; ++s) v += p[(size_t)s * 32];
        if (e < a.cout) a.db[e] += a.alpha * v;
    }
}
__global__ void wgrad_reduce_kernel(const WgradArgs a) { wgrad_reduce_body(a, (long long)blockIdx.x * blockDim.x + threadIdx.x); }
// blockIdx.y = table entry (entries that wrote straight into dW have nslices == 1 and nothing to fold)
__global__ void wgrad_reduce_batch_kernel(const WgradArgs* __restrict__ table) {
    const WgradArgs a = table[blockIdx.y];
    if (a.nslices > 1) wgrad_reduce_body(a, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// grid decomposition shared by the workspace query and the launch
struct WgradPlan { int tiles_x, tiles_y, ncit_main, ncit, ngroups, nslices, mt, shape; };
// A MAIN input of <= 3 channels and no latent segment (the critic's first conv on RGB, codes/models/modules/architecture.py:452): its one input
// tile is run in the latent tile's form — the 27 (tap, channel) columns as ONE MFMA tile (wgrad_body<..., LATK>) — instead of nine 32-wide tiles
// that are 3/32 full: the layer is presented to the kernel as "no main tiles + a latent tile of cin_main channels".
static bool main_as_latk(const esr_wgrad_desc* d) { return !d->xlat.hi && d->cin_main <= 3 && d->upsample <= 1; }
static bool desc_is_s2d(const esr_wgrad_desc* d) {
    return d->tap_masks[0] == 432 && d->tap_masks[1] == 216 && d->tap_masks[2] == 54 && d->tap_masks[3] == 27 && d->cin_main % 128 == 0 &&
           d->dy.fmt != ESR_FMT_F16;
}
// shapes: the launch runs the kernel flavour that has the 16x16 / 32x8 pixel tiles (the space-to-depth one): take the shape with the fewest tiles
static WgradPlan wgrad_plan(const esr_wgrad_desc* d, int target_wgs = 512, bool shapes = false) {
    WgradPlan p;
    p.mt = (d->cout + 31) / 32;
    p.shape = 0;
    p.tiles_x = (d->W + WG_TW - 1) / WG_TW;
    p.tiles_y = (d->H + WG_TH - 1) / WG_TH;
    if (shapes && (d->upsample <= 1) && !main_as_latk(d))
        for (int sh = 1; sh <= 2; ++sh) {
            const int tw = WG_TW >> sh, th = WG_TH << sh;
            const int tx = (d->W + tw - 1) / tw, ty = (d->H + th - 1) / th;
            if ((long long)tx * ty < (long long)p.tiles_x * p.tiles_y) { p.tiles_x = tx; p.tiles_y = ty; p.shape = sh; }
        }
    p.ncit_main = (d->cin_main + 31) / 32;
    const int lat = d->xlat.hi ? d->lat : 0;
    p.ncit = p.ncit_main + (lat ? 1 : 0);
    if (main_as_latk(d)) { p.ncit_main = 0; p.ncit = 1; }
    p.ngroups = p.ncit * p.mt;
    const int ntiles = p.tiles_x * p.tiles_y * d->B;
    int ns = (target_wgs + p.ngroups - 1) / p.ngroups;   // single launch: ~2 workgroups per CU in total (one resident at a time)
    if (ns > ntiles) ns = ntiles;
    if (ns < 1) ns = 1;
    p.nslices = ns;
    return p;
}

static int wgrad_validate(const esr_wgrad_desc* d) {
    if (!d || !d->dy.hi || !d->x.hi || !d->dw || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->cout <= 0 || d->cin_main <= 0) return ESR_E_ARG;
    const int ups = d->upsample <= 0 ? 1 : d->upsample;
    if (d->x.H * ups != d->H || d->x.W * ups != d->W || d->dy.H != d->H || d->dy.W != d->W) return ESR_E_ARG;
    if (d->xlat.hi && (ups != 1 || d->xlat.H != d->H || d->xlat.W != d->W || d->lat <= 0 || d->lat > 8)) return ESR_E_ARG;
    const bool split = d->dy.lo != nullptr;
    if ((d->x.lo != nullptr) != split) return ESR_E_ARG;
    if (d->xlat.hi && ((d->xlat.lo != nullptr) != split)) return ESR_E_ARG;
    if (d->x.fmt != d->dy.fmt || (d->xlat.hi && d->xlat.fmt != d->dy.fmt)) return ESR_E_ARG;      // one element format per contraction
    if (d->dy.fmt == ESR_FMT_F16 && split) return ESR_E_UNSUPPORTED;   // fp16 operands: the hi planes only (one MFMA per product)
    return ESR_OK;
}

static WgradArgs wgrad_args(const esr_wgrad_desc* d, const WgradPlan& p, float* ws) {
    WgradArgs a{};
    a.shape = p.shape;
    a.latk = (d->xlat.hi && d->lat > 0 && d->lat <= 3) ? 1 : 0;
    a.dy = to_dview(d->dy);
    a.x = to_dview(d->x);
    a.xlat = to_dview(d->xlat);
    a.lat = d->xlat.hi ? d->lat : 0;
    a.ups = d->upsample <= 0 ? 1 : d->upsample;
    a.cout = d->cout;
    a.cin_main = d->cin_main;
    a.cin_total = d->cin_main + a.lat;
    a.B = d->B; a.H = d->H; a.W = d->W;
    a.Wx_p = d->x.W + 2;
    a.tiles_x = p.tiles_x;
    a.tiles_y = p.tiles_y;
    a.ncit_main = p.ncit_main;
    a.ncit = p.ncit;
    a.mt = p.mt;
    a.ngroups = p.ngroups;
    a.nslices = p.nslices;
    a.alpha = d->alpha;
    a.dw = d->dw;
    a.db = d->db;
    a.ws = ws;
    // (a hint: any other mask pattern accumulates all nine taps — zeros where the weights are structurally zero)
    a.tapmode = (d->tap_masks[0] == 432 && d->tap_masks[1] == 216 && d->tap_masks[2] == 54 && d->tap_masks[3] == 27 && d->cin_main % 128 == 0) ? 1 : 0;
    if (main_as_latk(d)) {
        a.xlat = a.x;
        a.lat = d->cin_main;
        a.cin_main = 0;
        a.latk = 1;
    }

    return a;
}

static inline int64_t wgrad_partial_floats(const WgradPlan& p) {
    return p.nslices == 1 ? 0 : (int64_t)p.ngroups * p.nslices * (9 * 1024) + (int64_t)p.mt * p.nslices * 32;
}

// LDS stages of the tile loop.  One-plane operands (bf16 / f16): two stages AND two workgroups per CU (2 x 2 x 38 KB + nothing else = the
// 160 KB; four tiles of copies in flight per CU): generator launch 6.57 -> 6.14 ms, critic launches 675 -> 575 us at the configs[2] shapes.
// hi+lo operands: one stage, two workgroups (two stages would leave one workgroup per CU: measured slower).
static int wgrad_stages(bool one_plane) { return one_plane ? 2 : 1; }
static size_t wgrad_lds(int npl, int nst) {
    const size_t stages = (size_t)nst * npl * (WG_X_BYTES + WG_Y_BYTES), red = (size_t)4 * 5 * 1024 * 4;
    return stages > red ? stages : red;
}

// batch layout inside the caller's workspace: [WgradArgs table][int4 workgroup map][fp32 partial sums]
// Work unit = one (layer, input tile, output tile) block over one image tile.  Every layer's pixel sum is cut into as many slices as
// it takes to keep a workgroup's share near total / 768 (three waves of workgroups over the chip): with hundreds of equal layers
// that is one slice each (no partial sums at all), a few big high-resolution layers get several, a small net gets many.
struct BatchPlan { int64_t unit, nwg, table_bytes, map_bytes, partial_floats; };
static WgradPlan batch_entry_plan(const esr_wgrad_desc* d, int64_t unit, bool shapes) {
    WgradPlan p = wgrad_plan(d, 1, shapes);
    const int64_t ntiles = (int64_t)p.tiles_x * p.tiles_y * d->B;
    int64_t ns = (ntiles + unit - 1) / unit;
    if (ns > ntiles) ns = ntiles;
    if (ns < 1) ns = 1;
    p.nslices = (int)ns;
    return p;
}
static bool batch_is_s2d(const esr_wgrad_desc* descs, int n) {
    for (int i = 0; i < n; ++i)
        if (desc_is_s2d(&descs[i])) return true;
    return false;
}
// The work items of one layer, slice-major so that co-running workgroups of one layer read different images
static void entry_work(int i, const WgradPlan& p, std::vector<int4>& pairs) {
    for (int sl = 0; sl < p.nslices; ++sl)
        for (int g = 0; g < p.ngroups; ++g) pairs.push_back(make_int4(i, g, sl, 0));
}
static BatchPlan batch_plan(const esr_wgrad_desc* descs, int n, int64_t unit_override = 0) {
    BatchPlan b{};
    int64_t work = 0;
    const bool shapes = batch_is_s2d(descs, n);
    for (int i = 0; i < n; ++i) {
        const WgradPlan p = wgrad_plan(&descs[i], 1, shapes);
        work += (int64_t)p.ngroups * p.tiles_x * p.tiles_y * descs[i].B;
    }
    b.unit = work / 768 > 0 ? work / 768 : 1;
    if (unit_override > 0) b.unit = unit_override;      // the slicing of a LARGER set this one is a part of (esr_conv3x3_wgrad_batch_unit)
    for (int i = 0; i < n; ++i) {
        const WgradPlan p = batch_entry_plan(&descs[i], b.unit, shapes);
        b.nwg += (int64_t)p.ngroups * p.nslices;
        b.partial_floats += wgrad_partial_floats(p);
    }
    b.nwg = (b.nwg + 1023) / 1024 * 1024;                       // map entries: the work items padded to whole rounds of the XCD-aware order
    b.table_bytes = (((int64_t)n * sizeof(WgradArgs)) + 255) / 256 * 256;
    b.map_bytes = ((b.nwg * (int64_t)sizeof(int4)) + 255) / 256 * 256;
    return b;
}

}  // namespace

#ifdef ESR_TRACE
extern "C" void esr_debug_trace_wgrad(void* buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wtrace), &buf, sizeof(buf)); }
#endif
extern "C" int64_t esr_conv3x3_wgrad_workspace_floats(const esr_wgrad_desc* d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->cout <= 0 || d->cin_main <= 0) return ESR_E_ARG;
    const int64_t n = wgrad_partial_floats(wgrad_plan(d, 512, desc_is_s2d(d)));
    return n > 0 ? n : 1;
}

extern "C" int esr_conv3x3_wgrad(const esr_wgrad_desc* d, esr_stream_t stream) {
    const int rc = wgrad_validate(d);
    if (rc != ESR_OK) return rc;
    if (!d->workspace || d->workspace_floats < esr_conv3x3_wgrad_workspace_floats(d)) return ESR_E_ARG;
    const WgradPlan p = wgrad_plan(d, 512, desc_is_s2d(d));
    const WgradArgs a = wgrad_args(d, p, d->workspace);
    const bool split = d->dy.lo != nullptr;
    const bool f16 = d->dy.fmt == ESR_FMT_F16;
    const int nst = wgrad_stages(!split);                    // fp16 operands: single plane only (wgrad_validate)
    void (*k)(const WgradArgs) = f16 ? (nst == 2 ? conv3x3_wgrad_kernel<1, 2, 1> : conv3x3_wgrad_kernel<1, 1, 1>)
                               : split ? (nst == 2 ? conv3x3_wgrad_kernel<2, 2, 0> : conv3x3_wgrad_kernel<2, 1, 0>)
                                       : (nst == 2 ? conv3x3_wgrad_kernel<1, 2, 0> : conv3x3_wgrad_kernel<1, 1, 0>);
    if (a.tapmode == 1 && !f16)
        k = split ? (nst == 2 ? conv3x3_wgrad_kernel<2, 2, 0, true> : conv3x3_wgrad_kernel<2, 1, 0, true>)
                  : (nst == 2 ? conv3x3_wgrad_kernel<1, 2, 0, true> : conv3x3_wgrad_kernel<1, 1, 0, true>);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   // k varies per call: no caching
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(k, dim3(p.ngroups * p.nslices), dim3(256), wgrad_lds(split ? 2 : 1, nst), (hipStream_t)stream, a);
    ESR_CHECK_LAUNCH();
    if (p.nslices > 1) {
        const long long nred = (long long)p.ngroups * 9 * 1024 + p.mt * 32;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nred + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
        ESR_CHECK_LAUNCH();
    }
    return ESR_OK;
}

static int64_t batch_workspace_bytes(const esr_wgrad_desc* descs, int n, int64_t unit) {
    if (!descs || n <= 0 || unit < 0) return ESR_E_ARG;
    for (int i = 0; i < n; ++i)
        if (descs[i].B <= 0 || descs[i].H <= 0 || descs[i].W <= 0 || descs[i].cout <= 0 || descs[i].cin_main <= 0) return ESR_E_ARG;
    const BatchPlan b = batch_plan(descs, n, unit);
    return b.table_bytes + b.map_bytes + (b.partial_floats + 1) * 4;
}
extern "C" int64_t esr_conv3x3_wgrad_batch_workspace_bytes(const esr_wgrad_desc* descs, int n) { return batch_workspace_bytes(descs, n, 0); }
extern "C" int64_t esr_conv3x3_wgrad_batch_part_workspace_bytes(const esr_wgrad_desc* descs, int n, int64_t unit) { return batch_workspace_bytes(descs, n, unit); }
extern "C" int64_t esr_conv3x3_wgrad_batch_unit(const esr_wgrad_desc* descs, int n) {
    if (!descs || n <= 0) return ESR_E_ARG;
    for (int i = 0; i < n; ++i)
        if (descs[i].B <= 0 || descs[i].H <= 0 || descs[i].W <= 0 || descs[i].cout <= 0 || descs[i].cin_main <= 0) return ESR_E_ARG;
    return batch_plan(descs, n).unit;
}

static int batch_upload(const esr_wgrad_desc* descs, int n, void* workspace, int64_t workspace_bytes, esr_wgrad_batch_plan* plan, int64_t unit, esr_stream_t stream);
extern "C" int esr_conv3x3_wgrad_batch_upload(const esr_wgrad_desc* descs, int n, void* workspace, int64_t workspace_bytes, esr_wgrad_batch_plan* plan,
                                              esr_stream_t stream) {
    return batch_upload(descs, n, workspace, workspace_bytes, plan, 0, stream);
}
extern "C" int esr_conv3x3_wgrad_batch_part_upload(const esr_wgrad_desc* descs, int n, void* workspace, int64_t workspace_bytes, esr_wgrad_batch_plan* plan,
                                                   int64_t unit, esr_stream_t stream) {
    return unit > 0 ? batch_upload(descs, n, workspace, workspace_bytes, plan, unit, stream) : ESR_E_ARG;
}
static int batch_upload(const esr_wgrad_desc* descs, int n, void* workspace, int64_t workspace_bytes, esr_wgrad_batch_plan* plan, int64_t unit, esr_stream_t stream) {
    if (!descs || n <= 0 || !workspace || !plan) return ESR_E_ARG;
    const bool split = descs[0].dy.lo != nullptr;
    for (int i = 0; i < n; ++i) {
        const int rc = wgrad_validate(&descs[i]);
        if (rc != ESR_OK) return rc;
        if ((descs[i].dy.lo != nullptr) != split || descs[i].dy.fmt != descs[0].dy.fmt) return ESR_E_ARG;      // one operand format per batch
    }
    if (workspace_bytes < batch_workspace_bytes(descs, n, unit)) return ESR_E_ARG;
    const BatchPlan b = batch_plan(descs, n, unit);
    std::vector<WgradArgs> table(n);
    std::vector<int4> work;                                      // the work list in its natural order (layer, slice, group)
    work.reserve((size_t)b.nwg);
    const bool shapes = batch_is_s2d(descs, n);
    float* partials = (float*)((char*)workspace + b.table_bytes + b.map_bytes);
    int64_t pf = 0;
    int max_red = 0;
    for (int i = 0; i < n; ++i) {
        const WgradPlan p = batch_entry_plan(&descs[i], b.unit, shapes);
        table[i] = wgrad_args(&descs[i], p, partials + pf);
        pf += wgrad_partial_floats(p);
        entry_work(i, p, work);
        if (p.nslices > 1 && p.ngroups * 9 * 1024 + p.mt * 32 > max_red) max_red = p.ngroups * 9 * 1024 + p.mt * 32;
    }
    // Order of the work list over the XCDs.  The hardware deals workgroup b to XCD b % 8, so the groups of one layer (same dY tiles for a cot, same
    // X tiles for a cit: neighbours in the list) run on eight different L2s: 25 GB of L2 misses per configs[2] launch (bf16) for 1.9 GB of distinct
    // operands, hit rate 32 %.  Handing each XCD runs of `run` consecutive items cuts the misses (9 vs 12.5 M FETCH_SIZE units at a whole eighth per
    // XCD) — and does NOT buy time: 6.26 (dealt) / 6.10 (runs of 4) / 6.20 (16) / 7.51 (64) / 7.09 ms (an eighth each; the critic's ten unequal
    // layers 0.60 / 0.60 / 0.65 / 0.71 / 1.83 ms): long runs unbalance the XCDs, and the launch is bound by neither the fabric nor the MFMA pipe
    // (37 % busy) but by each workgroup's copy -> wait -> multiply chain.  Runs of 4: the traffic saving that costs nothing.
    std::vector<int4> map((size_t)b.nwg, make_int4(-1, 0, 0, 0));
    const int64_t npair = ((int64_t)work.size() + 1023) / 1024 * 1024;       // grid of the launch
    if (npair > b.nwg) return ESR_E_UNSUPPORTED;
    constexpr int64_t run = 4;
    for (int64_t blk = 0; blk < npair; ++blk) {
        const int64_t xcd = blk % 8, k = blk / 8;
        const int64_t item = ((k / run) * 8 + xcd) * run + k % run;
        if (item < (int64_t)work.size()) map[(size_t)blk] = work[(size_t)item];
    }
    hipStream_t s = (hipStream_t)stream;
    // pageable host memory: the runtime stages it before returning (the vectors go out of scope); a host-blocking copy, not graph-capturable —
    // which is why it is its own entry point: callers upload once per descriptor set and replay esr_conv3x3_wgrad_batch_run
    if (hipMemcpyAsync(workspace, table.data(), (size_t)n * sizeof(WgradArgs), hipMemcpyHostToDevice, s) != hipSuccess) return ESR_E_LAUNCH;
    if (hipMemcpyAsync((char*)workspace + b.table_bytes, map.data(), (size_t)b.nwg * sizeof(int4), hipMemcpyHostToDevice, s) != hipSuccess)
        return ESR_E_LAUNCH;
    plan->nwg = npair;
    plan->table_bytes = b.table_bytes;
    plan->n = n;
    plan->max_red = max_red;
    plan->split = split ? 1 : 0;
    plan->f16 = descs[0].dy.fmt == ESR_FMT_F16 ? 1 : 0;
    plan->s2d = 0;
    for (const WgradArgs& t : table) plan->s2d |= t.tapmode == 1 ? 1 : 0;
    plan->reserved = 0;
    return ESR_OK;
}

static int wgrad_batch_launch(const void* workspace, const esr_wgrad_batch_plan* plan, esr_stream_t stream, bool side);
extern "C" int esr_conv3x3_wgrad_batch_run(const void* workspace, const esr_wgrad_batch_plan* plan, esr_stream_t stream) {
    return wgrad_batch_launch(workspace, plan, stream, false);
}
extern "C" int esr_conv3x3_wgrad_batch_run_side(const void* workspace, const esr_wgrad_batch_plan* plan, esr_stream_t stream) {
    return wgrad_batch_launch(workspace, plan, stream, true);
}
extern "C" int esr_conv3x3_wgrad_side_occupancy(int f16) {
    void (*k)(const WgradArgs*, const int4*) = f16 ? conv3x3_wgrad_batch_kernel<1, 2, 1, false, true> : conv3x3_wgrad_batch_kernel<1, 2, 0, false, true>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { (void)hipGetLastError(); return ESR_E_LAUNCH; }
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k, 256, wgrad_lds(1, 2)) != hipSuccess) { (void)hipGetLastError(); return ESR_E_LAUNCH; }
    return nb;
}
static int wgrad_batch_launch(const void* workspace, const esr_wgrad_batch_plan* plan, esr_stream_t stream, bool side) {
    if (!workspace || !plan || plan->n <= 0 || plan->nwg <= 0) return ESR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool f16 = plan->f16 != 0, split = plan->split != 0;
    const int nst = wgrad_stages(!split);
    void (*k)(const WgradArgs*, const int4*) = f16 ? (nst == 2 ? conv3x3_wgrad_batch_kernel<1, 2, 1> : conv3x3_wgrad_batch_kernel<1, 1, 1>)
                                             : split ? (nst == 2 ? conv3x3_wgrad_batch_kernel<2, 2, 0> : conv3x3_wgrad_batch_kernel<2, 1, 0>)
                                                     : (nst == 2 ? conv3x3_wgrad_batch_kernel<1, 2, 0> : conv3x3_wgrad_batch_kernel<1, 1, 0>);
    if (plan->s2d && !f16)        // some layers are space-to-depth embedded stride-2 convs: the variant that skips their zero blocks
        k = split ? (nst == 2 ? conv3x3_wgrad_batch_kernel<2, 2, 0, true> : conv3x3_wgrad_batch_kernel<2, 1, 0, true>)
                  : (nst == 2 ? conv3x3_wgrad_batch_kernel<1, 2, 0, true> : conv3x3_wgrad_batch_kernel<1, 1, 0, true>);
    if (side && !split && !plan->s2d)          // one-plane operands only: the hi+lo and space-to-depth forms run as they are
        k = f16 ? conv3x3_wgrad_batch_kernel<1, 2, 1, false, true> : conv3x3_wgrad_batch_kernel<1, 2, 0, false, true>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   // k varies per call: no caching
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(k, dim3((unsigned)plan->nwg), dim3(256), wgrad_lds(split ? 2 : 1, nst), s, (const WgradArgs*)workspace,
                       (const int4*)((const char*)workspace + plan->table_bytes));
    ESR_CHECK_LAUNCH();
    if (plan->max_red > 0) {
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)((plan->max_red + 255) / 256), (unsigned)plan->n), dim3(256), 0, s,
                           (const WgradArgs*)workspace);
        ESR_CHECK_LAUNCH();
    }
    return ESR_OK;
}

namespace {
__global__ void wgrad_rebase_kernel(WgradArgs* table, int n, long long delta_bytes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    table[i].dw = (float*)((char*)table[i].dw + delta_bytes);
    if (table[i].db) table[i].db = (float*)((char*)table[i].db + delta_bytes);
}
}  // namespace

extern "C" int esr_conv3x3_wgrad_batch_rebase(void* workspace, const esr_wgrad_batch_plan* plan, int64_t delta_bytes, esr_stream_t stream) {
    if (!workspace || !plan || plan->n <= 0) return ESR_E_ARG;
    if (delta_bytes == 0) return ESR_OK;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(wgrad_rebase_kernel, dim3((plan->n + 63) / 64), dim3(64), 0, (hipStream_t)stream, (WgradArgs*)workspace, plan->n, (long long)delta_bytes);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_conv3x3_wgrad_batch(const esr_wgrad_desc* descs, int n, void* workspace, int64_t workspace_bytes, esr_stream_t stream) {
    esr_wgrad_batch_plan plan;
    const int rc = esr_conv3x3_wgrad_batch_upload(descs, n, workspace, workspace_bytes, &plan, stream);
    return rc != ESR_OK ? rc : esr_conv3x3_wgrad_batch_run(workspace, &plan, stream);
}
