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
    int chunks;                       // bn_apply: blocks per (image, channel group) plane
    // bn_apply<0, .., FIN>: the affine is derived here from the mode-0 sums (what bn_finalize_kernel would have left in scale / shift)
    const double* fin_sums;           // [groups][C][2]
    float eps, momentum;
    const float* beta;
    float *o_mean, *o_rstd, *o_scale, *o_shift, *run_mean, *run_var;
};

// mean, rstd, scale, shift of channel c in group g from the mode-0 sums — bn_finalize_kernel's arithmetic, shared with the fused form
struct BnStat { float mu, r, sc, sh; double var; };
__device__ __forceinline__ BnStat bn_stat(const double* sums, long long gc, double n, float eps, float gm, float bt) {
    const double s = sums[gc * 2], ss = sums[gc * 2 + 1];
    const double mu = s / n;
    double var = ss / n - mu * mu;
    if (var < 0.0) var = 0.0;
    BnStat o;
    o.r = (float)(1.0 / sqrt(var + (double)eps));
    o.mu = (float)mu;
    o.sc = gm * o.r;
    o.sh = bt - gm * o.r * (float)mu;
    o.var = var;
    return o;
}

// raw 16-byte vectors of one pixel (hi [+ lo]) and their decoding: loads are issued for several pixels before the first is decoded
struct Raw8 { uint4 h, l; };
__device__ __forceinline__ Raw8 ldraw(const CView& v, long long o) {
    Raw8 r;
    r.h = v.hi[o];
    r.l = v.lo ? v.lo[o] : make_uint4(0, 0, 0, 0);
    return r;
}
__device__ __forceinline__ void dec8(const CView& v, const Raw8& r, float* f) {
    const uint32_t hw[4] = {r.h.x, r.h.y, r.h.z, r.h.w}, lw[4] = {r.l.x, r.l.y, r.l.z, r.l.w};
    const bool lo = v.lo != nullptr;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t hb = (e & 1) ? (hw[e >> 1] >> 16) : (hw[e >> 1] & 0xFFFF), lb = (e & 1) ? (lw[e >> 1] >> 16) : (lw[e >> 1] & 0xFFFF);
        f[e] = v.fmt == ESR_FMT_F16 ? h2f(hb) + (lo ? h2f(lb) : 0.f) : bf2f(hb) + (lo ? bf2f(lb) : 0.f);
    }
}

// MODE 0: sum y, sum y^2                      (forward statistics)
// MODE 1: sum dyb, sum dyb*xh                 (backward)
// MODE 2: sum u, sum u*xh, sum u*dyb          (double backward)
// Block = (channel group, statistics group, one of nsplit interleaved shares of the group's pixels); every thread walks its pixels two at a
// time (both pixels' loads in flight together), keeps fp32 partial sums, and the block folds them in double: 16 threads per value, each adding 16
// of the 256 per-thread partials, then four shuffles — the fold used to be ONE thread per value walking all 256 (a ~20 k-cycle tail behind a
// main loop of 16-32 pixels per thread).
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
    const long long step = (long long)a.nsplit * 256;
    auto offs = [&](long long p, long long& oy, long long& od, long long& ou) {
        const int b = g * Bg + (int)(p / hw);
        const int r = (int)(p % hw), yy = r / a.W, xx = r % a.W;
        oy = voff<false>(a.y, b, cg, yy, xx);
        od = MODE >= 1 ? voff<S2D>(a.dz, b, cg, yy, xx) : 0;
        ou = MODE == 2 ? voff<false>(a.u, b, cg, yy, xx) : 0;
    };
    auto add = [&](const Raw8& ry, const Raw8& rd, const Raw8& ru) {
        float fy[8], fd[8], fu[8];
        dec8(a.y, ry, fy);
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { acc[0][e] += fy[e]; acc[1][e] += fy[e] * fy[e]; }
            return;
        }
        dec8(a.dz, rd, fd);
        if (MODE == 2) dec8(a.u, ru, fu);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float pre = sc[e] * fy[e] + sh[e];
            const float dyb = fd[e] * (pre > 0.f ? 1.f : a.slope);
            const float xh = (fy[e] - mu[e]) * rs[e];
            if (MODE == 1) { acc[0][e] += dyb; acc[1][e] += dyb * xh; }
            else { acc[0][e] += fu[e]; acc[1][e] += fu[e] * xh; acc[2][e] += fu[e] * dyb; }
        }
    };
    long long p = (long long)sp * 256 + threadIdx.x;
    for (; p + step < a.npg; p += 2 * step) {
        long long oy0, od0, ou0, oy1, od1, ou1;
        offs(p, oy0, od0, ou0);
        offs(p + step, oy1, od1, ou1);
        Raw8 ry0 = ldraw(a.y, oy0), ry1 = ldraw(a.y, oy1), rd0{}, rd1{}, ru0{}, ru1{};
        if (MODE >= 1) { rd0 = ldraw(a.dz, od0); rd1 = ldraw(a.dz, od1); }
        if (MODE == 2) { ru0 = ldraw(a.u, ou0); ru1 = ldraw(a.u, ou1); }
        add(ry0, rd0, ru0);
        add(ry1, rd1, ru1);
    }
    if (p < a.npg) {
        long long oy0, od0, ou0;
        offs(p, oy0, od0, ou0);
        Raw8 ry0 = ldraw(a.y, oy0), rd0{}, ru0{};
        if (MODE >= 1) rd0 = ldraw(a.dz, od0);
        if (MODE == 2) ru0 = ldraw(a.u, ou0);
        add(ry0, rd0, ru0);
    }
    constexpr int RS = 256 + 16;                     // row pitch: the four rows a wave folds at once fall into disjoint bank windows
    __shared__ float red[K * 8][RS];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[k * 8 + e][threadIdx.x] = acc[k][e];
    __syncthreads();
#pragma unroll
    for (int v0 = 0; v0 < K * 8; v0 += 16) {
        const int vi = v0 + (threadIdx.x >> 4), j = threadIdx.x & 15;
        double s = 0.0;
        if (vi < K * 8)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += (double)red[vi][i * 16 + j];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (j == 0 && vi < K * 8) {
            const int k = vi / 8, e = vi % 8, c = cg * 8 + e;
            if (c < a.C) atomicAdd(a.sums + ((long long)g * a.C + c) * K + k, s);
        }
    }
}

// MODE 0: z = lrelu(scale*y + shift)                                   -> out0 (S2D layout when S2D)
// MODE 1: dy (see header)                                              -> out0;  dz read with S2D indexing when S2D
// MODE 2: g_dz -> out0 (S2D layout when S2D), g_y -> out1
// Block = (image, channel group, chunk of U x 256 pixels): the 8 channels' per-channel constants (affine, statistics, the backward sums and their
// double -> float conversions: ~60 loads and a dozen fp64 operations) are formed ONCE per thread and serve its U pixels, whose loads are all
// issued before the first result is computed.  (One pixel per thread re-derived them for every 16-byte vector: the kernels ran at 3 TB/s.)
// FIN (MODE 0): no bn_finalize launch in front of this one — eight threads of every block derive their group's affine for the block's 8
// channels from the sums (fp64, bn_stat) and hand it to the others through LDS; the first block of each channel group also stores mean /
// rstd / scale / shift of ALL groups for the backward passes and moves the running statistics, group after group, as bn_finalize_kernel does.
template <int MODE, bool S2D, int U, bool FIN = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnArgs a) {
    static_assert(!FIN || MODE == 0, "the fused finalize belongs to the forward");
    int t = blockIdx.x;
    const int chunk = t % a.chunks;
    t /= a.chunks;
    const int cg = t % a.ncg, b = t / a.ncg;
    const int g = b / (a.B / a.groups);
    const int hw = a.H * a.W;
    float sc[8], sh[8], mu[8], rs[8], k1[8], s1[8], s2[8], t1[8], t2[8], q[8];
    bool okc[8];
    const double inv_n = 1.0 / (double)a.npg;
    // the pixels' loads go out first: the per-channel constants (and, FIN, the statistics' fp64 arithmetic and its barrier) are formed under them
    Raw8 ry[U], rd[U], ru[U];
    int ys[U], xs[U];
    bool live[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int p = (chunk * U + i) * 256 + threadIdx.x;
        live[i] = p < hw;
        const int pc = live[i] ? p : 0;
        ys[i] = pc / a.W;
        xs[i] = pc - ys[i] * a.W;
        ry[i] = ldraw(a.y, voff<false>(a.y, b, cg, ys[i], xs[i]));
        if (MODE >= 1) rd[i] = ldraw(a.dz, voff<S2D>(a.dz, b, cg, ys[i], xs[i]));
        if (MODE == 2) ru[i] = ldraw(a.u, voff<false>(a.u, b, cg, ys[i], xs[i]));
    }
    __shared__ float fin[FIN ? 16 : 1];
    if constexpr (FIN) {
        const int c = cg * 8 + (int)threadIdx.x;
        if (threadIdx.x < 8 && c < a.C) {
            const float gm = a.gamma ? a.gamma[c] : 1.f, bt = a.beta ? a.beta[c] : 0.f;
            const double n = (double)a.npg;
            const BnStat me = bn_stat(a.fin_sums, (long long)g * a.C + c, n, a.eps, gm, bt);
            fin[threadIdx.x] = me.sc;
            fin[8 + threadIdx.x] = me.sh;
            if (b == 0 && chunk == 0) {
                float rm = a.run_mean ? a.run_mean[c] : 0.f, rv = a.run_var ? a.run_var[c] : 0.f;
                for (int gg = 0; gg < a.groups; ++gg) {
                    const BnStat o = bn_stat(a.fin_sums, (long long)gg * a.C + c, n, a.eps, gm, bt);
                    a.o_mean[gg * a.C + c] = o.mu;
                    a.o_rstd[gg * a.C + c] = o.r;
                    a.o_scale[gg * a.C + c] = o.sc;
                    a.o_shift[gg * a.C + c] = o.sh;
                    rm = (1.f - a.momentum) * rm + a.momentum * o.mu;
                    rv = (1.f - a.momentum) * rv + a.momentum * (float)(n > 1.0 ? o.var * n / (n - 1.0) : o.var);
                }
                if (a.run_mean) a.run_mean[c] = rm;
                if (a.run_var) a.run_var[c] = rv;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cg * 8 + e;
        const bool ok = c < a.C;
        okc[e] = ok;
        const long long gc = (long long)g * a.C + (ok ? c : 0);
        if constexpr (FIN) {
            sc[e] = ok ? fin[e] : 1.f;
            sh[e] = ok ? fin[8 + e] : 0.f;
        } else {
            sc[e] = (ok && a.scale) ? a.scale[gc] : 1.f;
            sh[e] = (ok && a.shift) ? a.shift[gc] : 0.f;
        }
        mu[e] = rs[e] = k1[e] = s1[e] = s2[e] = t1[e] = t2[e] = q[e] = 0.f;
        if (MODE >= 1 && !a.const_stats) {
            mu[e] = a.mean[gc];
            rs[e] = a.rstd[gc];
            k1[e] = (a.gamma ? a.gamma[ok ? c : 0] : 1.f) * rs[e];
            s1[e] = (float)(a.s2[gc * 2] * inv_n);                      // E[dyb], E[dyb*xh]
            s2[e] = (float)(a.s2[gc * 2 + 1] * inv_n);
            if (MODE == 2) {
                t1[e] = (float)(a.s3[gc * 3] * inv_n);
                t2[e] = (float)(a.s3[gc * 3 + 1] * inv_n);
                q[e] = (float)(a.s3[gc * 3 + 2] * inv_n) - t1[e] * s1[e] - t2[e] * s2[e];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
        if (!live[i]) continue;
        const int yy = ys[i], xx = xs[i];
        float fy[8], fd[8], fu[8], o0[8], o1[8];
        dec8(a.y, ry[i], fy);
        if (MODE >= 1) dec8(a.dz, rd[i], fd);
        if (MODE == 2) dec8(a.u, ru[i], fu);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = okc[e];
            const float pre = sc[e] * fy[e] + sh[e];
            if (MODE == 0) { o0[e] = ok ? (pre > 0.f ? pre : a.slope * pre) : 0.f; continue; }
            const float m = pre > 0.f ? 1.f : a.slope;
            const float dyb = fd[e] * m;
            if (a.const_stats) {                          // y -> z is a fixed affine map + activation
                if (MODE == 1) o0[e] = ok ? sc[e] * dyb : 0.f;
                else { o0[e] = ok ? m * sc[e] * fu[e] : 0.f; o1[e] = 0.f; }
                continue;
            }
            const float xh = (fy[e] - mu[e]) * rs[e];
            const float dterm = dyb - s1[e] - xh * s2[e];
            if (MODE == 1) { o0[e] = ok ? k1[e] * dterm : 0.f; continue; }
            const float uterm = fu[e] - t1[e] - xh * t2[e];
            o0[e] = ok ? m * k1[e] * uterm : 0.f;
            o1[e] = ok ? -k1[e] * rs[e] * (xh * q[e] + s2[e] * uterm + t2[e] * dterm) : 0.f;
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
}

// One thread per channel; the groups are processed in order (the running statistics see the calls in the order the reference makes them).
__global__ void bn_finalize_kernel(const double* sums, int groups, int C, double n, float eps, float momentum, const float* gamma, const float* beta,
                                   float* mean, float* rstd, float* scale, float* shift, float* running_mean, float* running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
    for (int g = 0; g < groups; ++g) {
        const BnStat o = bn_stat(sums, (long long)g * C + c, n, eps, gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f);
        mean[g * C + c] = o.mu;
        rstd[g * C + c] = o.r;
        scale[g * C + c] = o.sc;
        shift[g * C + c] = o.sh;
        rm = (1.f - momentum) * rm + momentum * o.mu;
        rv = (1.f - momentum) * rv + momentum * (float)(n > 1.0 ? o.var * n / (n - 1.0) : o.var);
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

#define ESR_LAUNCH3(KERNEL, MODE, S2D, U, GRID, ARGS)                                                                          \
    do {                                                                                                                        \
        ESR_CLEAR_ERR();                                                                                                        \
        if (S2D) hipLaunchKernelGGL((KERNEL<MODE, true, U>), dim3((unsigned)(GRID)), dim3(256), 0, (hipStream_t)stream, ARGS);  \
        else hipLaunchKernelGGL((KERNEL<MODE, false, U>), dim3((unsigned)(GRID)), dim3(256), 0, (hipStream_t)stream, ARGS);     \
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
    // pixels per thread: 4 (2 for the three-operand double backward: registers) on planes that fill such blocks, else 1
    const long long hw = (long long)a.H * a.W;
    const int u = hw >= 1024 ? (mode == 2 ? 2 : 4) : 1;
    a.chunks = (int)((hw + 256 * u - 1) / (256 * u));
    const long long grid = (long long)a.B * a.ncg * a.chunks;
    if (grid > 0x7FFFFFFFll) return ESR_E_UNSUPPORTED;
    if (mode == 0) { if (u == 4) ESR_LAUNCH3(bn_apply_kernel, 0, s2d, 4, grid, a); else ESR_LAUNCH3(bn_apply_kernel, 0, s2d, 1, grid, a); }
    else if (mode == 1) { if (u == 4) ESR_LAUNCH3(bn_apply_kernel, 1, s2d, 4, grid, a); else ESR_LAUNCH3(bn_apply_kernel, 1, s2d, 1, grid, a); }
    else { if (u == 2) ESR_LAUNCH3(bn_apply_kernel, 2, s2d, 2, grid, a); else ESR_LAUNCH3(bn_apply_kernel, 2, s2d, 1, grid, a); }
    return ESR_OK;
}

extern "C" int esr_bn_finalize_apply(const esr_bn_desc* d, const esr_cmd_bn_finalize* f, esr_stream_t stream) {
    BnArgs a{};
    const int rc = fill_common(a, d);
    if (rc != ESR_OK) return rc;
    if (!f || !f->sums || !f->mean || !f->rstd || !f->scale || !f->shift || f->groups != d->groups || f->C != d->C || d->const_stats) return ESR_E_ARG;
    if (f->n_per_group != a.npg) return ESR_E_ARG;
    const bool s2d = d->s2d != 0;
    if (!s2d_view_ok(a.out0, a, s2d)) return ESR_E_ARG;
    a.fin_sums = f->sums; a.eps = f->eps; a.momentum = f->momentum; a.gamma = f->gamma; a.beta = f->beta;
    a.o_mean = f->mean; a.o_rstd = f->rstd; a.o_scale = f->scale; a.o_shift = f->shift; a.run_mean = f->running_mean; a.run_var = f->running_var;
    const long long hw = (long long)a.H * a.W;
    const int u = hw >= 1024 ? 4 : 1;
    a.chunks = (int)((hw + 256 * u - 1) / (256 * u));
    const long long grid = (long long)a.B * a.ncg * a.chunks;
    if (grid > 0x7FFFFFFFll) return ESR_E_UNSUPPORTED;
    ESR_CLEAR_ERR();
    if (u == 4) {
        if (s2d) hipLaunchKernelGGL((bn_apply_kernel<0, true, 4, true>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((bn_apply_kernel<0, false, 4, true>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        if (s2d) hipLaunchKernelGGL((bn_apply_kernel<0, true, 1, true>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((bn_apply_kernel<0, false, 1, true>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    }
    ESR_CHECK_LAUNCH();
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
