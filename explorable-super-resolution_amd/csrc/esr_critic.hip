// The glue of the critic (reference: Discriminator_VGG_128, codes/models/modules/architecture.py:446-508 — every conv is followed by
// nn.BatchNorm2d in TRAINING mode and LeakyReLU(0.2), block.py:129-146) and of its WGAN-GP double backward (loss.py:260-279) on the conv
// kernels' activation layout: per-channel statistics, normalise + activate, their gradient and the gradient of that gradient, as fused
// single-pass kernels.  HBM-bound: every kernel reads each operand once (16-byte vectors, 8 channels per thread) and writes its result once.
//
//   forward        y -> stats (mean, rstd per channel and GROUP of images)      z = lrelu(a*y + b),  a = gamma*rstd, b = beta - a*mean
//   backward       dyb = dz * lrelu'(a*y + b);  S1 = sum dyb, S2 = sum dyb*xh  (xh = (y - mean)*rstd)
//                  dy = gamma*rstd * (dyb - S1/N - xh*S2/N);   dgamma = S2, dbeta = S1
//   double backward (u = cotangent of dy):  T1 = sum u, T2 = sum u*xh, T3 = sum u*dyb,  Q = T3/N - T1*S1/N^2 - T2*S2/N^2
//                  g_dz = lrelu' * gamma*rstd * (u - T1/N - xh*T2/N)
//                  g_y  = -gamma*rstd^2 * ( xh*Q + (S2/N)*(u - T1/N - xh*T2/N) + (T2/N)*(dyb - S1/N - xh*S2/N) )
//                  g_gamma = rstd * N * Q
// A "group" is a run of B/groups consecutive images with its own batch statistics: several calls of the critic (real, fake, interpolated
// batch) are executed as one launch while each keeps the statistics the reference's separate calls would give it.
// s2d: the activation view is stored space-to-depth (factor 2): logical pixel (y, x) of channel group cg lives at pixel (y/2, x/2) of group
// 16*(cg/4) + 4*s + cg%4, s = 2*(y&1) + (x&1) — the layout in which the following 4x4 stride-2 conv is a 3x3 stride-1 conv (esr_hip/critic.py);
// four consecutive groups (one 32-channel MFMA tile, two K chunks) share a parity, so that the zero blocks of that conv's weights are whole
// (tap, chunk) / (tap, tile) blocks.  Kernels that produce a conv INPUT also write its one-pixel zero border (nobody else does).
#include "esr_common.h"

namespace {

struct CView { uint4* hi; uint4* lo; long long bs, cs; int H, W, fmt; };   // H, W: stored interior size
static inline CView to_cview(const esr_act_view* v) {
    CView c{};
    if (v && v->hi) { c.hi = (uint4*)v->hi; c.lo = (uint4*)v->lo; c.bs = v->batch_stride; c.cs = v->cg_stride; c.H = v->H; c.W = v->W; c.fmt = v->fmt; }
    return c;
}

// offset of logical pixel (y, x) of group cg in image b;  S2D: see the header comment
template <bool S2D>
__device__ __forceinline__ long long voff(const CView& v, int b, int cg, int y, int x) {
    if (S2D) return b * v.bs + (long long)(16 * (cg >> 2) + 4 * (2 * (y & 1) + (x & 1)) + (cg & 3)) * v.cs + (long long)((y >> 1) + 1) * (v.W + 2) + ((x >> 1) + 1);
    return b * v.bs + (long long)cg * v.cs + (long long)(y + 1) * (v.W + 2) + (x + 1);
}
__device__ __forceinline__ void ld8(const CView& v, long long o, float* f) {
    const uint4 h = v.hi[o];
    const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
    if (v.lo) {
        const uint4 l = v.lo[o];
        const uint32_t lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t hb = (e & 1) ? (hw[e >> 1] >> 16) : (hw[e >> 1] & 0xFFFF), lb = (e & 1) ? (lw[e >> 1] >> 16) : (lw[e >> 1] & 0xFFFF);
            f[e] = v.fmt == ESR_FMT_F16 ? h2f(hb) + h2f(lb) : bf2f(hb) + bf2f(lb);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t hb = (e & 1) ? (hw[e >> 1] >> 16) : (hw[e >> 1] & 0xFFFF);
            f[e] = v.fmt == ESR_FMT_F16 ? h2f(hb) : bf2f(hb);
        }
    }
}
__device__ __forceinline__ void st8(const CView& v, long long o, const float* f) {
    uint32_t vh[8], vl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (v.fmt == ESR_FMT_F16) { vh[e] = f2h(f[e]); vl[e] = f2h(f[e] - h2f(vh[e])); }
        else split_bf16(f[e], vh[e], vl[e]);
    }
    v.hi[o] = make_uint4(vh[0] | (vh[1] << 16), vh[2] | (vh[3] << 16), vh[4] | (vh[5] << 16), vh[6] | (vh[7] << 16));
    if (v.lo) v.lo[o] = make_uint4(vl[0] | (vl[1] << 16), vl[2] | (vl[3] << 16), vl[4] | (vl[5] << 16), vl[6] | (vl[7] << 16));
}

// zero the border vectors adjacent to stored pixel (py, px) of the plane that offset `o` points into (py, px: 0-based interior coordinates of
// the STORED plane of size Hs x Ws): the threads of the frame's neighbours cover the whole one-pixel border exactly once or twice
__device__ __forceinline__ void zero_border(const CView& v, long long o, int py, int px, int Hs, int Ws) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    const int P = v.W + 2;
    auto put = [&](long long q) { v.hi[q] = z; if (v.lo) v.lo[q] = z; };
    if (px == 0) put(o - 1);
    if (px == Ws - 1) put(o + 1);
    if (py == 0) { put(o - P); if (px == 0) put(o - P - 1); if (px == Ws - 1) put(o - P + 1); }
    if (py == Hs - 1) { put(o + P); if (px == 0) put(o + P - 1); if (px == Ws - 1) put(o + P + 1); }
}

struct BnArgs {
    CView y, dz, u, out0, out1;       // y: conv output; dz: gradient w.r.t. the activated output; u: cotangent of dy; out0 / out1: results
    int B, groups, C, ncg, H, W;      // H, W: logical (un-S2D) size = y's
    const float *scale, *shift, *mean, *rstd, *gamma;      // [groups][C] (gamma: [C]); scale == NULL: identity affine (no normalisation)
    const double *s2, *s3;            // sums of the first / second backward, [groups][C][2] / [groups][C][3]
    double* sums;                     // reduction target
    float slope;
    int const_stats;                  // the affine is a constant of y (eval-mode BatchNorm / no norm): no mean terms in the gradients
    long long npg;                    // elements per channel and group: (B/groups) * H * W
    int nsplit;
};

// MODE 0: sum y, sum y^2                      (forward statistics)
// MODE 1: sum dyb, sum dyb*xh                 (backward)
// MODE 2: sum u, sum u*xh, sum u*dyb          (double backward)
template <int MODE, bool S2D>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const BnArgs a) {
    constexpr int K = MODE == 2 ? 3 : 2;
    const int cg = blockIdx.x % a.ncg, g = (blockIdx.x / a.ncg) % a.groups, sp = blockIdx.x / (a.ncg * a.groups);
    const int Bg = a.B / a.groups;
    float acc[K][8];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cg * 8 + e;
        const bool ok = c < a.C;
        sc[e] = (ok && a.scale) ? a.scale[g * a.C + c] : 1.f;
        sh[e] = (ok && a.shift) ? a.shift[g * a.C + c] : 0.f;
        mu[e] = (ok && a.mean) ? a.mean[g * a.C + c] : 0.f;
        rs[e] = (ok && a.rstd) ? a.rstd[g * a.C + c] : 1.f;
    }
    const long long hw = (long long)a.H * a.W;
    for (long long p = (long long)sp * 256 + threadIdx.x; p < a.npg; p += (long long)a.nsplit * 256) {
        const int b = g * Bg + (int)(p / hw);
        const int r = (int)(p % hw), yy = r / a.W, xx = r % a.W;
        float fy[8];
        ld8(a.y, voff<false>(a.y, b, cg, yy, xx), fy);
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { acc[0][e] += fy[e]; acc[1][e] += fy[e] * fy[e]; }
        } else {
            float fd[8];
            ld8(a.dz, voff<S2D>(a.dz, b, cg, yy, xx), fd);
            float fu[8];
            if (MODE == 2) ld8(a.u, voff<false>(a.u, b, cg, yy, xx), fu);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pre = sc[e] * fy[e] + sh[e];
                const float dyb = fd[e] * (pre > 0.f ? 1.f : a.slope);
                const float xh = (fy[e] - mu[e]) * rs[e];
                if (MODE == 1) { acc[0][e] += dyb; acc[1][e] += dyb * xh; }
                else { acc[0][e] += fu[e]; acc[1][e] += fu[e] * xh; acc[2][e] += fu[e] * dyb; }
            }
        }
    }
    __shared__ float red[K * 8][256];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[k * 8 + e][threadIdx.x] = acc[k][e];
    __syncthreads();
    if (threadIdx.x < K * 8) {
        double s = 0.0;
        for (int i = 0; i < 256; ++i) s += (double)red[threadIdx.x][i];
        const int k = threadIdx.x / 8, e = threadIdx.x % 8, c = cg * 8 + e;
        if (c < a.C) atomicAdd(a.sums + ((long long)g * a.C + c) * K + k, s);
    }
}

// MODE 0: z = lrelu(scale*y + shift)                                   -> out0 (S2D layout when S2D)
// MODE 1: dy (see header)                                              -> out0;  dz read with S2D indexing when S2D
// MODE 2: g_dz -> out0 (S2D layout when S2D), g_y -> out1
template <int MODE, bool S2D>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.B * a.ncg * a.H * a.W;
    if (idx >= total) return;
    const int xx = (int)(idx % a.W);
    long long t = idx / a.W;
    const int yy = (int)(t % a.H);
    t /= a.H;
    const int cg = (int)(t % a.ncg), b = (int)(t / a.ncg);
    const int g = b / (a.B / a.groups);
    float fy[8], o0[8], o1[8];
    ld8(a.y, voff<false>(a.y, b, cg, yy, xx), fy);
    float fd[8], fu[8];
    if (MODE >= 1) ld8(a.dz, voff<S2D>(a.dz, b, cg, yy, xx), fd);
    if (MODE == 2) ld8(a.u, voff<false>(a.u, b, cg, yy, xx), fu);
    const double inv_n = 1.0 / (double)a.npg;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cg * 8 + e;
        const bool ok = c < a.C;
        const long long gc = (long long)g * a.C + (ok ? c : 0);
        const float sc = (ok && a.scale) ? a.scale[gc] : 1.f, sh = (ok && a.shift) ? a.shift[gc] : 0.f;
        const float pre = sc * fy[e] + sh;
        if (MODE == 0) { o0[e] = ok ? (pre > 0.f ? pre : a.slope * pre) : 0.f; continue; }
        const float m = pre > 0.f ? 1.f : a.slope;
        const float dyb = fd[e] * m;
        if (a.const_stats) {                          // y -> z is a fixed affine map + activation
            if (MODE == 1) o0[e] = ok ? sc * dyb : 0.f;
            else { o0[e] = ok ? m * sc * fu[e] : 0.f; o1[e] = 0.f; }
            continue;
        }
        const float mu = a.mean[gc], rs = a.rstd[gc], gm = a.gamma ? a.gamma[ok ? c : 0] : 1.f;
        const float xh = (fy[e] - mu) * rs;
        const float s1 = (float)(a.s2[gc * 2] * inv_n), s2 = (float)(a.s2[gc * 2 + 1] * inv_n);       // E[dyb], E[dyb*xh]
        const float dterm = dyb - s1 - xh * s2;
        if (MODE == 1) { o0[e] = ok ? gm * rs * dterm : 0.f; continue; }
        const float t1 = (float)(a.s3[gc * 3] * inv_n), t2 = (float)(a.s3[gc * 3 + 1] * inv_n), t3 = (float)(a.s3[gc * 3 + 2] * inv_n);
        const float uterm = fu[e] - t1 - xh * t2;
        const float q = t3 - t1 * s1 - t2 * s2;
        o0[e] = ok ? m * gm * rs * uterm : 0.f;
        o1[e] = ok ? -gm * rs * rs * (xh * q + s2 * uterm + t2 * dterm) : 0.f;
    }
    // every result is the input of a conv (or weight-gradient) launch: write its zero border too
    const long long os = voff<S2D>(a.out0, b, cg, yy, xx), op = voff<false>(MODE == 2 ? a.out1 : a.out0, b, cg, yy, xx);
    if (MODE == 0 || MODE == 2) {
        st8(a.out0, os, o0);
        if (S2D) zero_border(a.out0, os, yy >> 1, xx >> 1, a.H / 2, a.W / 2);
        else zero_border(a.out0, os, yy, xx, a.H, a.W);
    }
    if (MODE == 1) { st8(a.out0, op, o0); zero_border(a.out0, op, yy, xx, a.H, a.W); }
    if (MODE == 2) { st8(a.out1, op, o1); zero_border(a.out1, op, yy, xx, a.H, a.W); }
}

// One thread per channel; the groups are processed in order (the running statistics see the calls in the order the reference makes them).
__global__ void bn_finalize_kernel(const double* sums, int groups, int C, double n, float eps, float momentum, const float* gamma, const float* beta,
                                   float* mean, float* rstd, float* scale, float* shift, float* running_mean, float* running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
    for (int g = 0; g < groups; ++g) {
        const double s = sums[((long long)g * C + c) * 2], ss = sums[((long long)g * C + c) * 2 + 1];
        const double mu = s / n;
        double var = ss / n - mu * mu;
        if (var < 0.0) var = 0.0;
        const float r = (float)(1.0 / sqrt(var + (double)eps));
        const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        mean[g * C + c] = (float)mu;
        rstd[g * C + c] = r;
        scale[g * C + c] = gm * r;
        shift[g * C + c] = bt - gm * r * (float)mu;
        rm = (1.f - momentum) * rm + momentum * (float)mu;
        rv = (1.f - momentum) * rv + momentum * (float)(n > 1.0 ? var * n / (n - 1.0) : var);
    }
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
}

// dgamma = sum_g S2, dbeta = sum_g S1 (backward);  g_gamma = sum_g rstd * N * Q (double backward)
__global__ void bn_param_grads_kernel(const double* s2, const double* s3, const float* rstd, int groups, int C, double n, float* dgamma, float* dbeta,
                                      float* g_gamma) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double dg = 0.0, db = 0.0, gg = 0.0;
    for (int g = 0; g < groups; ++g) {
        const long long gc = (long long)g * C + c;
        const double S1 = s2[gc * 2], S2 = s2[gc * 2 + 1];
        db += S1;
        dg += S2;
        if (s3) {
            const double T1 = s3[gc * 3], T2 = s3[gc * 3 + 1], T3 = s3[gc * 3 + 2];
            gg += (double)rstd[gc] * (T3 - T1 * S1 / n - T2 * S2 / n);
        }
    }
    if (dgamma) dgamma[c] = (float)dg;
    if (dbeta) dbeta[c] = (float)db;
    if (g_gamma) g_gamma[c] = (float)gg;
}

int fill_common(BnArgs& a, const esr_bn_desc* d) {
    if (!d || !d->y.hi || d->B <= 0 || d->groups <= 0 || d->B % d->groups || d->C <= 0 || d->C > d->y.ncg * 8) return ESR_E_ARG;
    a.y = to_cview(&d->y);
    a.dz = to_cview(&d->dz);
    a.u = to_cview(&d->u);
    a.out0 = to_cview(&d->out0);
    a.out1 = to_cview(&d->out1);
    a.B = d->B; a.groups = d->groups; a.C = d->C; a.ncg = (d->C + 7) / 8; a.H = d->y.H; a.W = d->y.W;
    a.scale = d->scale; a.shift = d->shift; a.mean = d->mean; a.rstd = d->rstd; a.gamma = d->gamma;
    a.s2 = d->sums2; a.s3 = d->sums3;
    a.slope = d->slope;
    a.const_stats = d->const_stats;
    a.npg = (long long)(d->B / d->groups) * a.H * a.W;
    long long ns = a.npg / 4096;
    a.nsplit = (int)(ns < 1 ? 1 : (ns > 64 ? 64 : ns));
    if (d->s2d && (a.H % 2 || a.W % 2)) return ESR_E_UNSUPPORTED;
    return ESR_OK;
}
bool s2d_view_ok(const CView& v, const BnArgs& a, bool s2d) { return v.hi && (s2d ? (v.H == a.H / 2 && v.W == a.W / 2) : (v.H == a.H && v.W == a.W)); }

}  // namespace

#define ESR_LAUNCH2(KERNEL, MODE, S2D, GRID, ARGS)                                                                             \
    do {                                                                                                                        \
        ESR_CLEAR_ERR();                                                                                                        \
        if (S2D) hipLaunchKernelGGL((KERNEL<MODE, true>), dim3((unsigned)(GRID)), dim3(256), 0, (hipStream_t)stream, ARGS);     \
        else hipLaunchKernelGGL((KERNEL<MODE, false>), dim3((unsigned)(GRID)), dim3(256), 0, (hipStream_t)stream, ARGS);        \
        ESR_CHECK_LAUNCH();                                                                                                     \
    } while (0)

extern "C" int esr_bn_reduce(const esr_bn_desc* d, int mode, double* sums, esr_stream_t stream) {
    BnArgs a{};
    const int rc = fill_common(a, d);
    if (rc != ESR_OK) return rc;
    if (!sums || mode < 0 || mode > 2) return ESR_E_ARG;
    if (mode >= 1 && (!s2d_view_ok(a.dz, a, d->s2d != 0) || (!a.const_stats && (!a.mean || !a.rstd)))) return ESR_E_ARG;
    if (mode == 2 && !s2d_view_ok(a.u, a, false)) return ESR_E_ARG;
    a.sums = sums;
    const long long grid = (long long)a.ncg * a.groups * a.nsplit;
    if (mode == 0) ESR_LAUNCH2(bn_reduce_kernel, 0, false, grid, a);
    else if (mode == 1) ESR_LAUNCH2(bn_reduce_kernel, 1, d->s2d != 0, grid, a);
    else ESR_LAUNCH2(bn_reduce_kernel, 2, d->s2d != 0, grid, a);
    return ESR_OK;
}

extern "C" int esr_bn_apply(const esr_bn_desc* d, int mode, esr_stream_t stream) {
    BnArgs a{};
    const int rc = fill_common(a, d);
    if (rc != ESR_OK) return rc;
    if (mode < 0 || mode > 2) return ESR_E_ARG;
    const bool s2d = d->s2d != 0;
    if (mode == 0 && !s2d_view_ok(a.out0, a, s2d)) return ESR_E_ARG;
    if (mode >= 1 && !s2d_view_ok(a.dz, a, s2d)) return ESR_E_ARG;
    if (mode == 1 && !s2d_view_ok(a.out0, a, false)) return ESR_E_ARG;
    if (mode == 2 && (!s2d_view_ok(a.u, a, false) || !s2d_view_ok(a.out0, a, s2d) || !s2d_view_ok(a.out1, a, false))) return ESR_E_ARG;
    if (mode >= 1 && !a.const_stats && (!a.mean || !a.rstd || !a.s2 || (mode == 2 && !a.s3))) return ESR_E_ARG;
    const long long total = (long long)a.B * a.ncg * a.H * a.W;
    const long long grid = (total + 255) / 256;
    if (mode == 0) ESR_LAUNCH2(bn_apply_kernel, 0, s2d, grid, a);
    else if (mode == 1) ESR_LAUNCH2(bn_apply_kernel, 1, s2d, grid, a);
    else ESR_LAUNCH2(bn_apply_kernel, 2, s2d, grid, a);
    return ESR_OK;
}

extern "C" int esr_bn_finalize(const double* sums, int groups, int C, int64_t n_per_group, float eps, float momentum, const float* gamma, const float* beta,
                               float* mean, float* rstd, float* scale, float* shift, float* running_mean, float* running_var, esr_stream_t stream) {
    if (!sums || groups <= 0 || C <= 0 || n_per_group <= 0 || !mean || !rstd || !scale || !shift) return ESR_E_ARG;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, sums, groups, C, (double)n_per_group, eps, momentum, gamma, beta,
                       mean, rstd, scale, shift, running_mean, running_var);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_bn_param_grads(const double* sums2, const double* sums3, const float* rstd, int groups, int C, int64_t n_per_group, float* dgamma, float* dbeta,
                                  float* g_gamma, esr_stream_t stream) {
    if (!sums2 || groups <= 0 || C <= 0 || n_per_group <= 0 || (g_gamma && (!sums3 || !rstd))) return ESR_E_ARG;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, sums2, sums3, rstd, groups, C, (double)n_per_group, dgamma,
                       dbeta, g_gamma);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}
