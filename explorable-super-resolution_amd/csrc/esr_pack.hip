// Layout conversion at the module boundary: fp32 NCHW <-> the channel-group activation layout of the conv kernels
// ([B][CG][H+2][W+2][8] bf16 hi/lo planes with a zero border), with the CEM eval-mode replicate padding
// (codes/CEM/CEMnet.py:286-295) and the latent bilinear /sf of RRDBNet.forward (codes/models/modules/architecture.py:284)
// fused into the read.  HBM-bound streaming kernels: one 16-byte vector store per thread, coalesced along W.
#include "esr_common.h"

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void pack_nchw_kernel(const float* __restrict__ src, long long sbs, int C, int h, int w, int c0, int nc, int pad, int down, uint4* hi,
                                 uint4* lo, long long bs, long long cs, int ncg, int Hd, int Wd, int fmt, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int Wp = Wd + 2, Hp = Hd + 2;
    const int X = (int)(idx % Wp);
    long long t = idx / Wp;
    const int Y = (int)(t % Hp);
    t /= Hp;
    const int cg = (int)(t % ncg);
    const int b = (int)(t / ncg);
    uint32_t vh[8], vl[8];
    const bool border = (X == 0) || (Y == 0) || (X == Wp - 1) || (Y == Hp - 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = cg * 8 + e;
        float v = 0.f;
        if (!border && ch < nc) {
            const float* p = src + b * sbs + ((long long)(c0 + ch) * h) * w;
            const int y = Y - 1, x = X - 1;
            if (down == 1) {
                v = p[(long long)clampi(y - pad, 0, h - 1) * w + clampi(x - pad, 0, w - 1)];
            } else {
                // F.interpolate(bilinear, align_corners=False, scale_factor=1/down) on the replicate-padded frame
                const int hp = h + 2 * pad, wp = w + 2 * pad;
                float sy = fmaxf((y + 0.5f) * (float)down - 0.5f, 0.f), sx = fmaxf((x + 0.5f) * (float)down - 0.5f, 0.f);
                const int y0 = (int)sy, x0 = (int)sx;
                const int y1 = y0 + (y0 < hp - 1 ? 1 : 0), x1 = x0 + (x0 < wp - 1 ? 1 : 0);
                const float ly = sy - (float)y0, lx = sx - (float)x0;
                const long long r0 = (long long)clampi(y0 - pad, 0, h - 1) * w, r1 = (long long)clampi(y1 - pad, 0, h - 1) * w;
                const int q0 = clampi(x0 - pad, 0, w - 1), q1 = clampi(x1 - pad, 0, w - 1);
                v = (1.f - ly) * ((1.f - lx) * p[r0 + q0] + lx * p[r0 + q1]) + ly * ((1.f - lx) * p[r1 + q0] + lx * p[r1 + q1]);
            }
        }
        if (fmt == ESR_FMT_F16) { vh[e] = f2h(v); vl[e] = f2h(v - h2f(vh[e])); }
        else split_bf16(v, vh[e], vl[e]);
    }
    const long long o = b * bs + cg * cs + (long long)Y * Wp + X;
    hi[o] = make_uint4(vh[0] | (vh[1] << 16), vh[2] | (vh[3] << 16), vh[4] | (vh[5] << 16), vh[6] | (vh[7] << 16));
    if (lo) lo[o] = make_uint4(vl[0] | (vl[1] << 16), vl[2] | (vl[3] << 16), vl[4] | (vl[5] << 16), vl[6] | (vl[7] << 16));
}

__global__ void unpack_nchw_kernel(const uint4* __restrict__ hi, const uint4* __restrict__ lo, long long bs, long long cs, int H, int W,
                                   int nc, float* __restrict__ dst, int fmt, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (b, cg, y, x)
    if (idx >= total) return;
    const int ncg = (nc + 7) / 8;
    const int x = (int)(idx % W);
    long long t = idx / W;
    const int y = (int)(t % H);
    t /= H;
    const int cg = (int)(t % ncg);
    const int b = (int)(t / ncg);
    const long long o = b * bs + cg * cs + (long long)(y + 1) * (W + 2) + (x + 1);
    const uint4 h = hi[o];
    const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
    uint32_t lw[4] = {0, 0, 0, 0};
    if (lo) { const uint4 l = lo[o]; lw[0] = l.x; lw[1] = l.y; lw[2] = l.z; lw[3] = l.w; }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = cg * 8 + e;
        if (ch >= nc) break;
        const uint32_t hb = (e & 1) ? (hw[e >> 1] >> 16) : (hw[e >> 1] & 0xFFFF);
        const uint32_t lb = (e & 1) ? (lw[e >> 1] >> 16) : (lw[e >> 1] & 0xFFFF);
        dst[((long long)(b * nc + ch) * H + y) * W + x] = fmt == ESR_FMT_F16 ? h2f(hb) + h2f(lb) : bf2f(hb) + bf2f(lb);
    }
}

__global__ void zero_kernel(uint4* p, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = i; k < n; k += stride) p[k] = make_uint4(0, 0, 0, 0);
}

}  // namespace

extern "C" int esr_pack_nchw(const float* src, int64_t src_batch_stride, int B, int C, int h, int w, int c0, int nc, int pad, int down, const esr_act_view* dst,
                             esr_stream_t stream) {
    if (!src || !dst || !dst->hi || B <= 0 || nc <= 0 || c0 < 0 || c0 + nc > C || pad < 0 || down < 1) return ESR_E_ARG;
    if ((h + 2 * pad) % down || (w + 2 * pad) % down) return ESR_E_ARG;
    const int Hd = (h + 2 * pad) / down, Wd = (w + 2 * pad) / down;
    if (dst->H != Hd || dst->W != Wd || dst->ncg * 8 < nc) return ESR_E_ARG;
    const long long total = (long long)B * dst->ncg * (Hd + 2) * (Wd + 2);
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(pack_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (long long)(src_batch_stride ? src_batch_stride : (int64_t)C * h * w), C, h, w, c0, nc, pad,
                       down, (uint4*)dst->hi, (uint4*)dst->lo, (long long)dst->batch_stride, (long long)dst->cg_stride, dst->ncg, Hd, Wd, dst->fmt, total);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_unpack_nchw(const esr_act_view* src, int B, int nc, float* dst, esr_stream_t stream) {
    if (!src || !src->hi || !dst || B <= 0 || nc <= 0 || src->ncg * 8 < nc) return ESR_E_ARG;
    const long long total = (long long)B * ((nc + 7) / 8) * src->H * src->W;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(unpack_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)src->hi,
                       (const uint4*)src->lo, (long long)src->batch_stride, (long long)src->cg_stride, src->H, src->W, nc, dst, src->fmt, total);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_zero(void* p, int64_t n16, esr_stream_t stream) {
    if (!p || n16 < 0) return ESR_E_ARG;
    if (n16 == 0) return ESR_OK;
    long long blocks = (n16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint4*)p, (long long)n16);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_version(void) { return 101; }
