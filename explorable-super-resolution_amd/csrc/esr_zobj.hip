// Z-objective kernels.  Soft histogram of a gray image, the O(pixels x bins) core of the reference's SoftHistogramLoss
// (codes/Z_optimization.py:170-209: ComputeSoftHistogram, non-KDE form — gray scale, patch size 1):
//     h[k] = (1/n) sum_i exp( -(d(v_i, c_k) + eps)^2 / T ),   c_k = lo + k (hi - lo)/(K - 1),
//     d(v, c) = min(|v - c|, |v - c - hi|, |v - c + hi|)          (the reference wraps the distance with period `max`, :177-179)
// The reference materialises the n x K matrix in float64 (512 x 512 pixels x 256 bins = 0.5 GB per image, its "GUI latency hog"); here a
// workgroup owns one BIN per thread and streams a slab of pixels through LDS: every thread accumulates its bin in a double register — no n x K
// tensor, no atomics; per-slab partial histograms are folded by the caller (K doubles per slab).  The backward is the transposed loop: one
// PIXEL per thread, the K upstream gradients in LDS.
#include "esr_common.h"

namespace {

constexpr int ZH_SLAB = 4096;      // pixels per workgroup in the forward

__device__ __forceinline__ float wrapped_diff(float v, float c, float hi, float& sgn) {
    // the signed difference among (v - c), (v - c - hi), (v - c + hi) with the smallest magnitude
    float d0 = v - c, d1 = d0 - hi, d2 = d0 + hi;
    float d = d0;
    if (fabsf(d1) < fabsf(d)) d = d1;
    if (fabsf(d2) < fabsf(d)) d = d2;
    sgn = d < 0.f ? -1.f : 1.f;          // d|d|/dv (0 -> +1, as torch's abs() gives sign 0 there only at exact ties: measure zero)
    return fabsf(d);
}

__global__ void soft_hist_fwd_kernel(const float* __restrict__ v, long long n, int K, float lo, float hi, float T, float eps, double* __restrict__ partial) {
    extern __shared__ float px[];        // a slab of pixel values
    const long long base = (long long)blockIdx.x * ZH_SLAB;
    const int cnt = (int)((n - base) < ZH_SLAB ? (n - base) : ZH_SLAB);
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) px[i] = v[base + i];
    __syncthreads();
    const float bw = K > 1 ? (hi - lo) / (float)(K - 1) : 0.f;
    const float invT = 1.f / T;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float c = lo + bw * (float)k;
        double acc = 0.0;
        float part = 0.f;
        for (int i = 0; i < cnt; ++i) {
            float s;
            const float d = wrapped_diff(px[i], c, hi, s) + eps;
            const float e = d * d * invT;
            if (e < 80.f) part += __expf(-e);
            if ((i & 255) == 255) { acc += (double)part; part = 0.f; }      // fold the fp32 partial every 256 terms
        }
        partial[(long long)blockIdx.x * K + k] = acc + (double)part;
    }
}

__global__ void soft_hist_bwd_kernel(const float* __restrict__ v, long long n, int K, float lo, float hi, float T, float eps, const float* __restrict__ gh,
                                     float* __restrict__ gv) {
    extern __shared__ float g[];         // upstream gradient per bin (already divided by n by the caller)
    for (int k = threadIdx.x; k < K; k += blockDim.x) g[k] = gh[k];
    __syncthreads();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = v[i];
    const float bw = K > 1 ? (hi - lo) / (float)(K - 1) : 0.f;
    const float invT = 1.f / T;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        float s;
        const float d = wrapped_diff(x, lo + bw * (float)k, hi, s) + eps;
        const float e = d * d * invT;
        if (e < 80.f) acc += g[k] * __expf(-e) * (-2.f * d * invT) * s;
    }
    gv[i] = acc;
}

}  // namespace

extern "C" int64_t esr_soft_hist_slabs(int64_t n) { return n <= 0 ? ESR_E_ARG : (n + ZH_SLAB - 1) / ZH_SLAB; }

extern "C" int esr_soft_hist_fwd(const float* v, int64_t n, int K, float lo, float hi, float T, float eps, double* partial, esr_stream_t stream) {
    if (!v || !partial || n <= 0 || K < 1 || K > 4096 || !(T > 0.f) || !(hi > lo)) return ESR_E_ARG;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(soft_hist_fwd_kernel, dim3((unsigned)((n + ZH_SLAB - 1) / ZH_SLAB)), dim3(256), ZH_SLAB * sizeof(float), (hipStream_t)stream, v,
                       (long long)n, K, lo, hi, T, eps, partial);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_soft_hist_bwd(const float* v, int64_t n, int K, float lo, float hi, float T, float eps, const float* gh, float* gv, esr_stream_t stream) {
    if (!v || !gh || !gv || n <= 0 || K < 1 || K > 4096 || !(T > 0.f) || !(hi > lo)) return ESR_E_ARG;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(soft_hist_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), K * sizeof(float), (hipStream_t)stream, v, (long long)n, K, lo, hi,
                       T, eps, gh, gv);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

// ---- per-image statistics of the SR output and their gradients: the reductions inside the Z-search loop and the L_struct training loss ----
//   kind 0  masked STD     (codes/Z_optimization.py:383-388: torch.std(image * mask, dim=(1,2,3)))           sums: sum v, sum v^2
//   kind 1  TV_Loss        (codes/Z_optimization.py:324-326: mean |v[x] - v[x+1]| + mean |v[y] - v[y+1]|)     sums: sum |dx|, sum |dy|
//   kind 2  structure tensor (codes/models/modules/loss.py:49-62,141-151: ix = v[x+1] - v[x], iy = v[y+1] - v[y] on the (H-1) x (W-1)
//           frame; mean ix^2, mean iy^2, mean ix iy over channels and pixels)                                   sums: sum ix^2, sum iy^2, sum ix iy
// with v = clamp(x, 0, 1) * mask when asked (the reference clamps the output — Output_Batch(within_0_1=True) — and multiplies by the user's
// image mask in separate full-size passes; here both are part of the read).  One pass over the image per call: a workgroup reduces a slab of
// pixels in registers / LDS and adds three doubles per image; the gradient kernels are one thread per pixel with per-image coefficients
// (the chain rule of the few scalar ops that follow the reduction, evaluated by the caller on [B]-sized tensors).
namespace {

constexpr int IS_THREADS = 256, IS_PER_THREAD = 8;

__device__ __forceinline__ float is_val(const float* __restrict__ img, const float* __restrict__ mask, int clamp01, int H, int W, int c, int y, int x) {
    float v = img[((long long)c * H + y) * W + x];
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    if (mask) v *= mask[(long long)y * W + x];
    return v;
}

template <int KIND>
__global__ __launch_bounds__(IS_THREADS) void img_stats_kernel(const float* __restrict__ x, int C, int H, int W, const float* __restrict__ mask, int clamp01,
                                                               double* __restrict__ sums) {
    const int b = blockIdx.y;
    const float* img = x + (long long)b * C * H * W;
    const long long n = (long long)C * H * W;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const long long base = (long long)blockIdx.x * IS_THREADS * IS_PER_THREAD;
#pragma unroll
    for (int k = 0; k < IS_PER_THREAD; ++k) {
        const long long i = base + (long long)k * IS_THREADS + threadIdx.x;
        if (i >= n) break;
        const int xx = (int)(i % W);
        const long long t = i / W;
        const int yy = (int)(t % H), c = (int)(t / H);
        const float v = is_val(img, mask, clamp01, H, W, c, yy, xx);
        if (KIND == 0) { a0 += v; a1 += v * v; }
        else {
            const float vx = xx + 1 < W ? is_val(img, mask, clamp01, H, W, c, yy, xx + 1) : v;
            const float vy = yy + 1 < H ? is_val(img, mask, clamp01, H, W, c, yy + 1, xx) : v;
            if (KIND == 1) { a0 += fabsf(v - vx); a1 += fabsf(v - vy); }
            else if (xx + 1 < W && yy + 1 < H) { const float ix = vx - v, iy = vy - v; a0 += ix * ix; a1 += iy * iy; a2 += ix * iy; }
        }
    }
    __shared__ float red[3][IS_THREADS];
    red[0][threadIdx.x] = a0; red[1][threadIdx.x] = a1; red[2][threadIdx.x] = a2;
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0.0;
        for (int i = 0; i < IS_THREADS; ++i) s += (double)red[threadIdx.x][i];
        if (threadIdx.x < (KIND == 2 ? 3 : 2)) atomicAdd(sums + (long long)b * 3 + threadIdx.x, s);
    }
}

// dx[b,c,y,x] for per-image coefficients coef[b][0..2]:
//   kind 0: (c0 * v + c1)                                  * dv/dx        (c0 = g / ((N-1) std), c1 = -mean * c0)
//   kind 1: c0 * (sgn(v - v[x+1]) - sgn(v[x-1] - v)) + c1 * (sgn(v - v[y+1]) - sgn(v[y-1] - v))       (c0 = g / Nx, c1 = g / Ny)
//   kind 2: with a(ix, iy) = c0 ix + c2 iy  (d/d ix),  b(ix, iy) = c1 iy + c2 ix  (d/d iy)  on the (H-1) x (W-1) frame:
//           -a(y, x) - b(y, x) + a(y, x-1) + b(y-1, x)       (c0 = 2 g0 / N, c1 = 2 g1 / N, c2 = g2 / N)
// dv/dx = mask * [0 <= x <= 1] (torch.clamp's inclusive gradient) as asked.
template <int KIND>
__global__ __launch_bounds__(IS_THREADS) void img_stats_grad_kernel(const float* __restrict__ x, int C, int H, int W, const float* __restrict__ mask, int clamp01,
                                                                    const float* __restrict__ coef, float* __restrict__ dx, int accumulate) {
    const int b = blockIdx.y;
    const float* img = x + (long long)b * C * H * W;
    const long long n = (long long)C * H * W;
    const long long i = (long long)blockIdx.x * IS_THREADS + threadIdx.x;
    if (i >= n) return;
    const int xx = (int)(i % W);
    const long long t = i / W;
    const int yy = (int)(t % H), c = (int)(t / H);
    const float c0 = coef[b * 3], c1 = coef[b * 3 + 1], c2 = coef[b * 3 + 2];
    auto V = [&](int y, int x_) { return is_val(img, mask, clamp01, H, W, c, y, x_); };
    auto sgn = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
    const float v = V(yy, xx);
    float g;
    if (KIND == 0) g = c0 * v + c1;
    else if (KIND == 1) {
        g = 0.f;
        if (xx + 1 < W) g += c0 * sgn(v - V(yy, xx + 1));
        if (xx > 0) g -= c0 * sgn(V(yy, xx - 1) - v);
        if (yy + 1 < H) g += c1 * sgn(v - V(yy + 1, xx));
        if (yy > 0) g -= c1 * sgn(V(yy - 1, xx) - v);
    } else {
        g = 0.f;
        auto A = [&](int y, int x_) { const float p = V(y, x_); const float ix = V(y, x_ + 1) - p, iy = V(y + 1, x_) - p; return c0 * ix + c2 * iy; };
        auto Bf = [&](int y, int x_) { const float p = V(y, x_); const float ix = V(y, x_ + 1) - p, iy = V(y + 1, x_) - p; return c1 * iy + c2 * ix; };
        if (xx + 1 < W && yy + 1 < H) g -= A(yy, xx) + Bf(yy, xx);
        if (xx > 0 && yy + 1 < H) g += A(yy, xx - 1);
        if (yy > 0 && xx + 1 < W) g += Bf(yy - 1, xx);
    }
    const float raw = img[((long long)c * H + yy) * W + xx];
    if (clamp01 && !(raw >= 0.f && raw <= 1.f)) g = 0.f;          // torch.clamp's gradient: 1 inside and AT the bounds ((x >= min) & (x <= max)), 0 outside (NaN: 0)
    if (mask) g *= mask[(long long)yy * W + xx];
    float* o = dx + (long long)b * n + i;
    *o = accumulate ? *o + g : g;
}

}  // namespace

extern "C" int esr_img_stats(const float* x, int B, int C, int H, int W, const float* mask, int clamp01, int kind, double* sums, esr_stream_t stream) {
    if (!x || !sums || B <= 0 || C <= 0 || H <= 0 || W <= 0 || kind < 0 || kind > 2) return ESR_E_ARG;
    const long long n = (long long)C * H * W;
    const dim3 grid((unsigned)((n + IS_THREADS * IS_PER_THREAD - 1) / (IS_THREADS * IS_PER_THREAD)), (unsigned)B);
    ESR_CLEAR_ERR();
    if (kind == 0) hipLaunchKernelGGL(img_stats_kernel<0>, grid, dim3(IS_THREADS), 0, (hipStream_t)stream, x, C, H, W, mask, clamp01, sums);
    else if (kind == 1) hipLaunchKernelGGL(img_stats_kernel<1>, grid, dim3(IS_THREADS), 0, (hipStream_t)stream, x, C, H, W, mask, clamp01, sums);
    else hipLaunchKernelGGL(img_stats_kernel<2>, grid, dim3(IS_THREADS), 0, (hipStream_t)stream, x, C, H, W, mask, clamp01, sums);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_img_stats_grad(const float* x, int B, int C, int H, int W, const float* mask, int clamp01, int kind, const float* coef, float* dx, int accumulate,
                                  esr_stream_t stream) {
    if (!x || !coef || !dx || B <= 0 || C <= 0 || H <= 0 || W <= 0 || kind < 0 || kind > 2) return ESR_E_ARG;
    const long long n = (long long)C * H * W;
    const dim3 grid((unsigned)((n + IS_THREADS - 1) / IS_THREADS), (unsigned)B);
    ESR_CLEAR_ERR();
    if (kind == 0) hipLaunchKernelGGL(img_stats_grad_kernel<0>, grid, dim3(IS_THREADS), 0, (hipStream_t)stream, x, C, H, W, mask, clamp01, coef, dx, accumulate);
    else if (kind == 1) hipLaunchKernelGGL(img_stats_grad_kernel<1>, grid, dim3(IS_THREADS), 0, (hipStream_t)stream, x, C, H, W, mask, clamp01, coef, dx, accumulate);
    else hipLaunchKernelGGL(img_stats_grad_kernel<2>, grid, dim3(IS_THREADS), 0, (hipStream_t)stream, x, C, H, W, mask, clamp01, coef, dx, accumulate);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}
