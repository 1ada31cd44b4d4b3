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

namespace {

__device__ __forceinline__ void unpack8(const uint4* hi, const uint4* lo, long long o, float (&v)[8]) {
    const uint4 h = hi[o];
    const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = bf2f(hw[e] & 0xFFFF); v[2 * e + 1] = bf2f(hw[e] >> 16); }
    if (lo) {
        const uint4 l = lo[o];
        const uint32_t lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += bf2f(lw[e] & 0xFFFF); v[2 * e + 1] += bf2f(lw[e] >> 16); }
    }
}

__device__ __forceinline__ void pack8(uint4* hi, uint4* lo, long long o, const float (&v)[8]) {
    uint32_t h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split_bf16(v[e], h[e], l[e]);
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
        unpack8(A.hi, A.lo, b * A.bs + cg * A.cs + pix, a8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = alpha * a8[e];
    }
    if (Bv.hi) {
        const int Wb = W * s + 2;
        for (int dy = 0; dy < s; ++dy)
            for (int dx = 0; dx < s; ++dx) {
                float b8[8];
                unpack8(Bv.hi, Bv.lo, b * Bv.bs + cg * Bv.cs + (long long)(y * s + dy + 1) * Wb + (x * s + dx + 1), b8);
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
    pack8((uint4*)out.hi, (uint4*)out.lo, b * out.bs + cg * out.cs + pix, v);
}

// Adjoint of esr_pack_nchw: act-layout gradient (interior (h+2pad)/down x (w+2pad)/down) -> fp32 NCHW gradient of the
// un-padded source, dst[b][c0+c][y][x] (+)= sum over padded positions that the replicate padding maps onto (y,x) of
//   down == 1 : g[pos]
//   down  > 1 : sum of bilinear tap weights * g[lr pos]   (the same taps the forward used)
// `accumulate` adds into dst instead of overwriting (latent: HR-resolution convs + LR-resolution convs both contribute).
__global__ void unpack_grad_kernel(DView G, float* __restrict__ dst, long long dbs, int C, int h, int w, int c0, int nc, int pad, int down,
                                   int accumulate, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, c, y, x) of the un-padded source
    if (idx >= total) return;
    const int x = (int)(idx % w);
    long long t = idx / w;
    const int y = (int)(t % h);
    t /= h;
    const int c = (int)(t % nc);
    const int b = (int)(t / nc);
    const int hp = h + 2 * pad, wp = w + 2 * pad;
    // padded-frame positions mapping onto (y, x)
    const int y_lo = y == 0 ? 0 : y + pad, y_hi = y == h - 1 ? hp - 1 : y + pad;
    const int x_lo = x == 0 ? 0 : x + pad, x_hi = x == w - 1 ? wp - 1 : x + pad;
    const int Hd = hp / down, Wd = wp / down;
    const int cg = c >> 3, e = c & 7;
    const long long base = b * G.bs + cg * G.cs;
    float acc = 0.f;
    auto g_at = [&](int yy, int xx) -> float {   // gradient channel c at act-layout interior pixel (yy, xx)
        const long long o = base + (long long)(yy + 1) * (Wd + 2) + (xx + 1);
        const uint4 hv = G.hi[o];
        const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w};
        float v = bf2f((e & 1) ? (hw[e >> 1] >> 16) : (hw[e >> 1] & 0xFFFF));
        if (G.lo) {
            const uint4 lv = G.lo[o];
            const uint32_t lw[4] = {lv.x, lv.y, lv.z, lv.w};
            v += bf2f((e & 1) ? (lw[e >> 1] >> 16) : (lw[e >> 1] & 0xFFFF));
        }
        return v;
    };
    if (down == 1) {
        for (int yy = y_lo; yy <= y_hi; ++yy)
            for (int xx = x_lo; xx <= x_hi; ++xx) acc += g_at(yy, xx);
    } else {
        // forward: lr(i) = (1-l)*hr(y0) + l*hr(y1), src = (i+0.5)*down-0.5, y0 = floor(src), y1 = min(y0+1, hp-1), l = src - y0
        for (int yy = y_lo; yy <= y_hi; ++yy) {
            for (int xx = x_lo; xx <= x_hi; ++xx) {
                // LR pixels whose taps include padded-HR pixel (yy, xx): at most two per axis
                float wy[2]; int iy[2]; int ny = 0;
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
                float wx[2]; int ix[2]; int nx = 0;
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
                    for (int q = 0; q < nx; ++q) acc += wy[p] * wx[q] * g_at(iy[p], ix[q]);
            }
        }
    }
    float* d = dst + b * dbs + ((long long)(c0 + c) * h + y) * w + x;
    *d = accumulate ? *d + acc : acc;
}

}  // namespace

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

extern "C" int esr_unpack_grad_nchw(const esr_act_view* G, float* dst, int64_t dst_batch_stride, int B, int C, int h, int w, int c0, int nc, int pad,
                                    int down, int accumulate, esr_stream_t stream) {
    if (!G || !G->hi || !dst || B <= 0 || nc <= 0 || c0 < 0 || c0 + nc > C || pad < 0 || down < 1) return ESR_E_ARG;
    if ((h + 2 * pad) % down || (w + 2 * pad) % down) return ESR_E_ARG;
    if (G->H != (h + 2 * pad) / down || G->W != (w + 2 * pad) / down || G->ncg * 8 < nc) return ESR_E_ARG;
    const long long total = (long long)B * nc * h * w;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(unpack_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, to_dview(*G), dst,
                       (long long)(dst_batch_stride ? dst_batch_stride : (int64_t)C * h * w), C, h, w, c0, nc, pad, down, accumulate, total);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}
