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
