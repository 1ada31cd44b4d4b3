// Consistency Enforcing Module filters (fixed depth-wise taps, fp32 NCHW).
// Reference: codes/CEM/CEMnet.py:243-252 (Filter_Layer) as wired in CEM_PyTorch.__init__ (:254-281) and used by
// CEM_PyTorch.forward (:303-311).  The reference runs three dense depth-wise cuDNN convolutions around explicit
// ReplicationPad2d / zero-stuffing / strided-view tensors; here each op is one kernel that
//   * applies the replicate padding by clamping indices (no padded copy),
//   * evaluates the strided downscale only at the kept pixels (1/sf^2 of the reference's MACs),
//   * evaluates the zero-stuffed upscale polyphase-wise (only taps that hit a non-zero sample),
//   * fuses  x - D(g)  into the downscale and  g + U(.) / tanh / crop  into the upscale.
// Index conventions (bit-exact with the reference): LR sample (i,j) sits at HR (sf*i+pre, sf*j+pre), pre = sf - floor(sf/2) - 1
// (codes/CEM/imresize_CEM.py:99-101); filter centre = floor(k/2); the replicate pad of the zero-stuffed image replicates
// whatever its first/last row is (a sample row only when pre == 0, i.e. sf == 2).
// Forms (esr_cem_sep_form): general k^2 kernels (one output per thread; anisotropic taps), separable tile kernels (a workgroup stages a
// window in LDS: small images), a streaming tile kernel (x8 downscale of small images) and — large images, where the tile kernels are bound by their own
// instruction stream — separable WAVE-streaming kernels: a wave walks down a strip of the image with the vertical pass in registers.
#include "esr_common.h"
#include <type_traits>

namespace {

// wave-streaming separable kernels on (the shipping behaviour); the instrumented build can switch them off for A/B runs (esr_debug_cem_wave)
bool g_cem_wave = true;
int g_cem_wave_target = 0;          // > 0 (instrumented build only): strips a launch aims for, instead of 0.9 x the chip's wave slots
int g_cem_wave_rmin = 0;            // > 0 (instrumented build only): downscale strip height, output rows
int g_cem_filt_rmin = 18;           // shortest LR-filter strip (a strip starts K - 1 rows early: 26 / 18 / 13 rows 25 / 21 / 24 us at configs[1])

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Workgroup -> (tile column, tile row, image plane) for the tiled kernels.  Workgroups are dispatched in linear order (x fastest), round-robin
// over the 8 XCDs, each with its own L2: XCD x sweeps a CONTIGUOUS run of the row-major (plane, tile row, tile column) order, so that the
// window overlap of neighbouring tiles (1.45x of g for the x4 downscale) is served by the L2 that fetched it (round 4 counters: 0.299 GB
// fetched for 0.135 GB of g, L2 hit rate 4 % with the linear order).
struct TileId { int bx, by; long long bz; };
__device__ __forceinline__ TileId xcd_tile() {
    const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
    const unsigned lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const unsigned xcd = lin & 7, q = total >> 3, r = total & 7;
    const unsigned t = xcd * q + (xcd < r ? xcd : r) + (lin >> 3);
    const unsigned row = t / nx;
    TileId id;
    id.bx = (int)(t - row * nx);
    id.bz = row / ny;
    id.by = (int)(row - (unsigned)id.bz * ny);
    return id;
}

__global__ void cem_downscale_kernel(const float* __restrict__ y, int h, int w, int sf, int pre, const float* __restrict__ taps, int k,
                                     const float* __restrict__ lr, int lr_pad, float* __restrict__ d, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx % w);
    long long t = idx / w;
    const int i = (int)(t % h);
    const long long bc = t / h;
    const int Hh = h * sf, Wh = w * sf, p = k / 2;
    const float* src = y + bc * Hh * (long long)Wh;
    const int Y0 = sf * i + pre - p, X0 = sf * j + pre - p;
    float acc = 0.f;
    for (int a = 0; a < k; ++a) {
        const float* row = src + (long long)clampi(Y0 + a, 0, Hh - 1) * Wh;
        const float* tr = taps + a * k;
        if (X0 >= 0 && X0 + k <= Wh) {
            for (int c = 0; c < k; ++c) acc = fmaf(tr[c], row[X0 + c], acc);
        } else {
            for (int c = 0; c < k; ++c) acc = fmaf(tr[c], row[clampi(X0 + c, 0, Wh - 1)], acc);
        }
    }
    if (lr) {
        const int h0 = h - 2 * lr_pad, w0 = w - 2 * lr_pad;
        acc = lr[(bc * h0 + clampi(i - lr_pad, 0, h0 - 1)) * w0 + clampi(j - lr_pad, 0, w0 - 1)] - acc;
    }
    d[idx] = acc;
}

__global__ void cem_lrfilter_kernel(const float* __restrict__ x, int h, int w, const float* __restrict__ taps, int k, float* __restrict__ out,
                                    long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx % w);
    long long t = idx / w;
    const int i = (int)(t % h);
    const long long bc = t / h;
    const int p = k / 2;
    const float* src = x + bc * h * (long long)w;
    float acc = 0.f;
    for (int a = 0; a < k; ++a) {
        const float* row = src + (long long)clampi(i + a - p, 0, h - 1) * w;
        const float* tr = taps + a * k;
        if (j - p >= 0 && j - p + k <= w) {
            for (int c = 0; c < k; ++c) acc = fmaf(tr[c], row[j - p + c], acc);
        } else {
            for (int c = 0; c < k; ++c) acc = fmaf(tr[c], row[clampi(j + c - p, 0, w - 1)], acc);
        }
    }
    out[idx] = acc;
}

// ---- LDS-tiled versions of the two filters with many taps per output (k = 17..45: 300-2000 MACs per output pixel).
// The per-thread global-memory versions above re-read every input ~k^2 times through L1/L2 and run at ~1/60 of the VALU rate; here a
// workgroup stages the (replicate-clamped) input window once, every inner-loop operand is a conflict-free ds_read_b32 plus a
// wave-uniform tap that lives in SGPRs, and each thread carries several independent accumulators.

constexpr int LT_TY = 16, LT_TX = 64;     // lrfilter output tile: 16 rows x 64 columns, 4 outputs (x, x+16, x+32, x+48) per thread

__global__ __launch_bounds__(256) void cem_lrfilter_tiled_kernel(const float* __restrict__ x, int h, int w, const float* __restrict__ taps, int k,
                                                               float* __restrict__ out, int pitch) {
    extern __shared__ float tile[];          // (LT_TY + k - 1) rows x pitch; pitch % 32 == 16: two rows of 16 lanes hit disjoint banks
    const int p = k / 2;
    const int x0 = blockIdx.x * LT_TX, y0 = blockIdx.y * LT_TY;
    const long long bc = blockIdx.z;
    const float* src = x + bc * h * (long long)w;
    const int rows = LT_TY + k - 1, cols = LT_TX + k - 1;
    for (int e = threadIdx.x; e < rows * cols; e += 256) {
        const int r = e / cols, c = e - r * cols;
        tile[r * pitch + c] = src[(long long)clampi(y0 + r - p, 0, h - 1) * w + clampi(x0 + c - p, 0, w - 1)];
    }
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int a = 0; a < k; ++a) {
        const float* row = tile + (ty + a) * pitch + tx;
        const float* tr = taps + a * k;
        for (int c = 0; c < k; ++c) {
            const float t = tr[c];
            a0 = fmaf(t, row[c], a0);
            a1 = fmaf(t, row[c + 16], a1);
            a2 = fmaf(t, row[c + 32], a2);
            a3 = fmaf(t, row[c + 48], a3);
        }
    }
    const int Y = y0 + ty;
    if (Y < h) {
        float* o = out + (bc * h + Y) * (long long)w + x0 + tx;
        if (x0 + tx < w) o[0] = a0;
        if (x0 + tx + 16 < w) o[16] = a1;
        if (x0 + tx + 32 < w) o[32] = a2;
        if (x0 + tx + 48 < w) o[48] = a3;
    }
}

constexpr int DT = 16;                       // downscale output tile: 16 x 16 low-resolution pixels, one per thread

// The input window is stored de-interleaved by column phase — plane[col % sf][row][col / sf] — so that for a given tap the 16 lanes of
// an output row (input columns sf apart) read consecutive words; qpitch % 8 == 4 for sf = 4 (sf * qpitch % 32 == 16) spreads the rows.
__global__ __launch_bounds__(256) void cem_downscale_tiled_kernel(const float* __restrict__ y, int h, int w, int sf, int pre, const float* __restrict__ taps,
                                                                int k, const float* __restrict__ lr, int lr_pad, float* __restrict__ d, int qpitch,
                                                                int rows) {
    extern __shared__ float tile[];          // [sf][rows][qpitch]
    const int p = k / 2, Hh = h * sf, Wh = w * sf;
    const int j0 = blockIdx.x * DT, i0 = blockIdx.y * DT;
    const long long bc = blockIdx.z;
    const float* src = y + bc * Hh * (long long)Wh;
    const int Yb = sf * i0 + pre - p, Xb = sf * j0 + pre - p;        // window origin in the high-resolution frame
    const int cols = (DT - 1) * sf + k;
    const int lg = (sf & (sf - 1)) == 0 ? __builtin_ctz(sf) : -1;    // power-of-two factors: shifts instead of divisions
    for (int r = threadIdx.x >> 6; r < rows; r += 4) {               // a wave per row: coalesced reads, no index divisions
        const float* grow = src + (long long)clampi(Yb + r, 0, Hh - 1) * Wh;
        for (int c = threadIdx.x & 63; c < cols; c += 64) {
            const int ph = lg >= 0 ? (c & (sf - 1)) : c % sf, q = lg >= 0 ? (c >> lg) : c / sf;
            tile[(ph * rows + r) * qpitch + q] = grow[clampi(Xb + c, 0, Wh - 1)];
        }
    }
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    float acc0 = 0.f, acc1 = 0.f;
    for (int a = 0; a < k; ++a) {
        const float* tr = taps + a * k;
        const int r = ty * sf + a;
        for (int ph = 0; ph < sf; ++ph) {     // taps of one column phase: consecutive words of that phase's plane, no index divisions
            const float* pl = tile + (ph * rows + r) * qpitch + tx;
            int c = ph, q = 0;
            for (; c + sf < k; c += 2 * sf, q += 2) {
                acc0 = fmaf(tr[c], pl[q], acc0);
                acc1 = fmaf(tr[c + sf], pl[q + 1], acc1);
            }
            if (c < k) acc0 = fmaf(tr[c], pl[q], acc0);
        }
    }
    const int i = i0 + ty, j = j0 + tx;
    if (i < h && j < w) {
        float acc = acc0 + acc1;
        if (lr) {
            const int h0 = h - 2 * lr_pad, w0 = w - 2 * lr_pad;
            acc = lr[(bc * h0 + clampi(i - lr_pad, 0, h0 - 1)) * w0 + clampi(j - lr_pad, 0, w0 - 1)] - acc;
        }
        d[(bc * h + i) * (long long)w + j] = acc;
    }
}

template <bool TWO>
__global__ void cem_upscale_kernel(const float* __restrict__ f, const float* __restrict__ f2, int h, int w, int sf, int pre,
                                   const float* __restrict__ taps, int k, const float* __restrict__ g, int crop, int mode, float range,
                                   float* __restrict__ out, float* __restrict__ out2, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int Hh = h * sf, Wh = w * sf, Ho = Hh - 2 * crop, Wo = Wh - 2 * crop, p = k / 2;
    const int xo = (int)(idx % Wo);
    long long t = idx / Wo;
    const int yo = (int)(t % Ho);
    const long long bc = t / Ho;
    const int Y = yo + crop, X = xo + crop;
    const float* s1 = f + bc * h * (long long)w;
    const float* s2 = TWO ? f2 + bc * h * (long long)w : nullptr;
    float u1 = 0.f, u2 = 0.f;
    // first tap index whose stuffed coordinate lands on a sample:  Y + a - p == pre (mod sf)
    int a0 = (pre + p - Y) % sf; if (a0 < 0) a0 += sf;
    int b0 = (pre + p - X) % sf; if (b0 < 0) b0 += sf;
    auto row_accum = [&](int a, int i) {
        const float* tr = taps + a * k;
        const float* r1 = s1 + (long long)i * w;
        const float* r2 = TWO ? s2 + (long long)i * w : nullptr;
        for (int b = b0; b < k; b += sf) {
            const int xx = X + b - p;
            if (xx < 0 || xx >= Wh) continue;
            const int j = (xx - pre) / sf;
            u1 = fmaf(tr[b], r1[j], u1);
            if (TWO) u2 = fmaf(tr[b], r2[j], u2);
        }
        if (pre == 0) {   // replicate pad replicates sample column 0 (only sf == 2): every tap left of the frame hits column 0
            for (int b = 0; X + b - p < 0 && b < k; ++b) {
                u1 = fmaf(tr[b], r1[0], u1);
                if (TWO) u2 = fmaf(tr[b], r2[0], u2);
            }
        }
    };
    for (int a = a0; a < k; a += sf) {
        const int yy = Y + a - p;
        if (yy < 0 || yy >= Hh) continue;
        row_accum(a, (yy - pre) / sf);
    }
    if (pre == 0)
        for (int a = 0; Y + a - p < 0 && a < k; ++a) row_accum(a, 0);

    const long long go = (bc * Hh + Y) * (long long)Wh + X;
    float r;
    if (mode == 0) r = u1;
    else if (mode == 1) r = g[go] + u1;
    else if (mode == 2) r = u1 + tanhf(g[go] - u2) * range;
    else { r = u1; out2[idx] = g[go] - u2; }
    out[idx] = r;
}

constexpr int UT_Y = 16, UT_X = 64;          // upscale output tile: 16 x 64 high-resolution pixels, 4 rows per wave, one pixel per thread

// Same arithmetic and edge rules as cem_upscale_kernel (polyphase: only taps that land on a sample; the replicate pad of the
// zero-stuffed image replicates a sample column only when pre == 0), with the low-resolution window and the taps staged in LDS and
// the per-tap index divisions replaced by increments.
template <bool TWO>
__global__ __launch_bounds__(256) void cem_upscale_tiled_kernel(const float* __restrict__ f, const float* __restrict__ f2, int h, int w, int sf, int pre,
                                                              const float* __restrict__ taps, int k, const float* __restrict__ g, int crop, int mode,
                                                              float range, float* __restrict__ out, float* __restrict__ out2, int wr, int wc) {
    extern __shared__ float sm[];             // taps [k*k] | window 1 [wr][wc] | window 2 [wr][wc]
    float* const st = sm;
    float* const w1 = sm + k * k;
    float* const w2 = w1 + wr * wc;
    const int Hh = h * sf, Wh = w * sf, Ho = Hh - 2 * crop, Wo = Wh - 2 * crop, p = k / 2;
    const long long bc = blockIdx.z;
    const int xo0 = blockIdx.x * UT_X, yo0 = blockIdx.y * UT_Y;
    // low-resolution window origin: the first sample any output of the tile can touch (floor division, may be negative)
    const int fy = yo0 + crop - p - pre, fx = xo0 + crop - p - pre;
    const int ib = (fy >= 0 ? fy / sf : -((-fy + sf - 1) / sf)), jb = (fx >= 0 ? fx / sf : -((-fx + sf - 1) / sf));
    for (int e = threadIdx.x; e < k * k; e += 256) st[e] = taps[e];
    const float* s1 = f + bc * h * (long long)w;
    const float* s2 = TWO ? f2 + bc * h * (long long)w : nullptr;
    for (int e = threadIdx.x; e < wr * wc; e += 256) {
        const int r = e / wc, c = e - r * wc;
        const int i = ib + r, j = jb + c;
        const bool in = i >= 0 && i < h && j >= 0 && j < w;
        w1[e] = in ? s1[(long long)i * w + j] : 0.f;
        if (TWO) w2[e] = in ? s2[(long long)i * w + j] : 0.f;
    }
    __syncthreads();
    const int xo = xo0 + (threadIdx.x & 63), yo = yo0 + (threadIdx.x >> 6);
#pragma unroll 1
    for (int rr = 0; rr < UT_Y; rr += 4) {
        const int yq = yo + rr;
        if (xo >= Wo || yq >= Ho) continue;
        const int Y = yq + crop, X = xo + crop;
        float u1 = 0.f, u2 = 0.f;
        int a0 = (pre + p - Y) % sf; if (a0 < 0) a0 += sf;
        int b0 = (pre + p - X) % sf; if (b0 < 0) b0 += sf;
        const int jfirst = (X + b0 - p - pre) / sf - jb;          // exact: X + b0 - p == pre (mod sf); may be negative only outside the frame
        auto row_accum = [&](int a, int il) {                     // il: window row
            const float* tr = st + a * k;
            const float* r1 = w1 + il * wc;
            const float* r2 = w2 + il * wc;
            int jl = jfirst;
            for (int b = b0; b < k; b += sf, ++jl) {
                const int xx = X + b - p;
                if (xx < 0 || xx >= Wh) continue;
                u1 = fmaf(tr[b], r1[jl], u1);
                if (TWO) u2 = fmaf(tr[b], r2[jl], u2);
            }
            if (pre == 0) {
                for (int b = 0; X + b - p < 0 && b < k; ++b) {
                    u1 = fmaf(tr[b], r1[-jb], u1);                // sample column 0
                    if (TWO) u2 = fmaf(tr[b], r2[-jb], u2);
                }
            }
        };
        int il = (Y + a0 - p - pre) / sf - ib;
        for (int a = a0; a < k; a += sf, ++il) {
            const int yy = Y + a - p;
            if (yy < 0 || yy >= Hh) continue;
            row_accum(a, il);
        }
        if (pre == 0)
            for (int a = 0; Y + a - p < 0 && a < k; ++a) row_accum(a, -ib);
        const long long go = (bc * Hh + Y) * (long long)Wh + X;
        const long long idx = (bc * Ho + yq) * (long long)Wo + xo;
        float r;
        if (mode == 0) r = u1;
        else if (mode == 1) r = g[go] + u1;
        else if (mode == 2) r = u1 + tanhf(g[go] - u2) * range;
        else { r = u1; out2[idx] = g[go] - u2; }
        out[idx] = r;
    }
}


// sum_{c < k} t[c] * at(c) with four accumulators: four scalar tap loads and four LDS reads in flight per step.  (Round 5 also tried the taps
// held across the lanes of a register and read with v_readlane — no memory access per tap: the lrfilter ran 2.3x SLOWER, 38 -> 86 us at
// configs[1]; the scalar cache serves these 17..45 words without a miss and its loads pipeline.)
template <typename At>
__device__ __forceinline__ float tap_sum(const float* __restrict__ t, const int k, At at) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int c = 0;
    for (; c + 4 <= k; c += 4) {
        a0 = fmaf(t[c], at(c), a0);
        a1 = fmaf(t[c + 1], at(c + 1), a1);
        a2 = fmaf(t[c + 2], at(c + 2), a2);
        a3 = fmaf(t[c + 3], at(c + 3), a3);
    }
    for (; c < k; ++c) a0 = fmaf(t[c], at(c), a0);
    return (a0 + a1) + (a2 + a3);
}

// tap_sum for a column of values `stride` words apart (the vertical passes): the same four running sums in the same order — tap c goes to sum
// c & 3, the tail to sum 0 — with the address as a pointer that advances once per four taps (the lambda form above re-derives c * stride per tap
// on the scalar unit: round-5 counters of the downscale kernel, 26.0 M scalar for 34.2 M vector instructions).
__device__ __forceinline__ float tap_sum_strided(const float* __restrict__ t, const int k, const float* p, const int stride) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int s2 = 2 * stride, s3 = 3 * stride, s4 = 4 * stride;
    int c = 0;
    for (; c + 4 <= k; c += 4, p += s4) {
        a0 = fmaf(t[c], p[0], a0);
        a1 = fmaf(t[c + 1], p[stride], a1);
        a2 = fmaf(t[c + 2], p[s2], a2);
        a3 = fmaf(t[c + 3], p[s3], a3);
    }
    for (; c < k; ++c, p += stride) a0 = fmaf(t[c], p[0], a0);
    return (a0 + a1) + (a2 + a3);
}

// tap_sum over a row stored de-interleaved by phase: window column c sits in plane c % SF at column c / SF (planes `pstride` words apart).  Same
// sums in the same order as tap_sum; SF is a compile-time constant, the taps are walked in blocks of lcm(SF, 4) so that every tap's plane, column
// step and running sum are constants of the unrolled block and the address is one pointer per plane that advances once per block.
template <int SF>
__device__ __forceinline__ float tap_sum_phased(const float* __restrict__ t, const int k, const float* base, const int pstride) {
    constexpr int BLK = SF % 4 == 0 ? SF : (SF % 2 == 0 ? 2 * SF : 4 * SF);      // lcm(SF, 4)
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const int kmain = k & ~3;                                 // taps [0, kmain) in fours, the tail to sum 0 (tap_sum's rule)
    int c0 = 0;
    const float* p = base;                                    // column c0 / SF of plane 0
    for (; c0 + BLK <= kmain; c0 += BLK, p += BLK / SF) {
#pragma unroll
        for (int j = 0; j < BLK; ++j) a[j & 3] = fmaf(t[c0 + j], p[(j % SF) * pstride + j / SF], a[j & 3]);
    }
#pragma unroll
    for (int j = 0; j < BLK; ++j) {                           // the last, partial block
        const int c = c0 + j;
        if (c < kmain) a[j & 3] = fmaf(t[c], p[(j % SF) * pstride + j / SF], a[j & 3]);
        else if (c < k) a[0] = fmaf(t[c], p[(j % SF) * pstride + j / SF], a[0]);
    }
    return (a[0] + a[1]) + (a[2] + a[3]);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Separable fast path.  The bicubic ds_kernel and its inv_hTh are rank one (sigma_2 / sigma_1 ~ 1e-16), taps[a][b] = tv[a]*th[b]:
// every filter is a horizontal pass followed by a vertical one (or the reverse) on the tile a workgroup holds in LDS — 2k instead of k^2 MACs
// per output (k = 17..45).  Same windows, same clamping / zero-stuffing index rules as the 2-D kernels above; the host selects this path at
// construction when the kernel passes the rank test and keeps the 2-D path for anisotropic (estimated) kernels.

__global__ __launch_bounds__(256) void cem_lrfilter_sep_kernel(const float* __restrict__ x, int h, int w, const float* __restrict__ tv, const float* __restrict__ th,
                                                             int k, float* __restrict__ out, int pitch) {
    extern __shared__ float tile[];          // window (LT_TY + k - 1) x pitch | horizontal pass (LT_TY + k - 1) x (LT_TX + 1)
    const int p = k / 2;
    const int x0 = blockIdx.x * LT_TX, y0 = blockIdx.y * LT_TY;
    const long long bc = blockIdx.z;
    const float* src = x + bc * h * (long long)w;
    const int rows = LT_TY + k - 1, cols = LT_TX + k - 1;
    float* const hp = tile + rows * pitch;
    // staging in batches of SB independent loads per thread: a load -> LDS-write -> next-load chain costs one memory latency per element
    constexpr int SB = 8;
    for (int e0 = threadIdx.x; e0 < rows * cols; e0 += 256 * SB) {
        float tmp[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int e = e0 + u * 256, r = e / cols, c = e - r * cols;
            tmp[u] = e < rows * cols ? src[(long long)clampi(y0 + r - p, 0, h - 1) * w + clampi(x0 + c - p, 0, w - 1)] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int e = e0 + u * 256, r = e / cols, c = e - r * cols;
            if (e < rows * cols) tile[r * pitch + c] = tmp[u];
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & 63;
    for (int r = threadIdx.x >> 6; r < rows; r += 4) {        // a wave per window row: 64 consecutive outputs, conflict-free reads
        const float* row = tile + r * pitch + lx;
        float a0 = 0.f, a1 = 0.f;
        int c = 0;
#pragma unroll 4
        for (; c + 1 < k; c += 2) { a0 = fmaf(th[c], row[c], a0); a1 = fmaf(th[c + 1], row[c + 1], a1); }
        if (c < k) a0 = fmaf(th[c], row[c], a0);
        hp[r * (LT_TX + 1) + lx] = a0 + a1;
    }
    __syncthreads();
    for (int ty = threadIdx.x >> 6; ty < LT_TY; ty += 4) {
        const float* col = hp + ty * (LT_TX + 1) + lx;
        float a0 = 0.f, a1 = 0.f;
        int a = 0;
#pragma unroll 4
        for (; a + 1 < k; a += 2) { a0 = fmaf(tv[a], col[a * (LT_TX + 1)], a0); a1 = fmaf(tv[a + 1], col[(a + 1) * (LT_TX + 1)], a1); }
        if (a < k) a0 = fmaf(tv[a], col[a * (LT_TX + 1)], a0);
        const int Y = y0 + ty, X = x0 + lx;
        if (Y < h && X < w) out[(bc * h + Y) * (long long)w + X] = a0 + a1;
    }
}

// SFT: the scale factor as a compile-time constant (2, 3, 4, 8) so that the index divisions / modulos fold into shifts and multiplies; 0 = run-time
template <int SFT>
__global__ __launch_bounds__(256) void cem_downscale_sep_kernel(const float* __restrict__ y, int h, int w, int sf_rt, int pre, const float* __restrict__ tv,
                                                              const float* __restrict__ th, int k, const float* __restrict__ lr, int lr_pad,
                                                              float* __restrict__ d, int qpitch, int rows, int hpp) {
    extern __shared__ float tile[];          // [sf][rows][qpitch] de-interleaved window (as cem_downscale_tiled_kernel) | horizontal pass [rows][hpp]
    const int sf = SFT ? SFT : sf_rt;
    const int p = k / 2, Hh = h * sf, Wh = w * sf;
    const TileId id = xcd_tile();
    const int j0 = id.bx * DT, i0 = id.by * DT;
    const long long bc = id.bz;
    const float* src = y + bc * Hh * (long long)Wh;
    const int Yb = sf * i0 + pre - p, Xb = sf * j0 + pre - p;
    const int cols = (DT - 1) * sf + k;
    const int lg = (sf & (sf - 1)) == 0 ? __builtin_ctz(sf) : -1;
    float* const hp = tile + sf * rows * qpitch;
    {
        // the window [rows][cols] as 16-byte vectors of 4 columns starting at a multiple of 4 (rows of a contiguous fp32 image whose width is a
        // multiple of 4 are 16-byte aligned); vectors that stick out of the image row, or an unaligned image, take four clamped scalar loads.
        // SB vectors per thread are in flight together; each lands as four LDS words in the de-interleaved planes.
        constexpr int SB = 8;                                 // (x4, k = 17: the 77-row window is 6.3 vectors per thread — one round trip, not two)
        const int Xa = (Xb >> 2) << 2;                       // floor to a multiple of 4 (arithmetic shift: also for negative Xb)
        const int nvec = (Xb + cols - Xa + 3) >> 2;
        const bool vec_ok = (Wh & 3) == 0 && (((size_t)src) & 15) == 0;
        for (int e0 = threadIdx.x; e0 < rows * nvec; e0 += 256 * SB) {
            float4 tmp[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int e = e0 + u * 256;
                tmp[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < rows * nvec) {
                    const int r = e / nvec, x = Xa + 4 * (e - r * nvec);
                    const float* rowp = src + (long long)clampi(Yb + r, 0, Hh - 1) * Wh;
                    if (vec_ok && x >= 0 && x + 3 < Wh) tmp[u] = *(const float4*)(rowp + x);
                    else tmp[u] = make_float4(rowp[clampi(x, 0, Wh - 1)], rowp[clampi(x + 1, 0, Wh - 1)], rowp[clampi(x + 2, 0, Wh - 1)], rowp[clampi(x + 3, 0, Wh - 1)]);
                }
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int e = e0 + u * 256;
                if (e >= rows * nvec) continue;
                const int r = e / nvec, c0 = Xa + 4 * (e - r * nvec) - Xb;
                const float v[4] = {tmp[u].x, tmp[u].y, tmp[u].z, tmp[u].w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int c = c0 + t;
                    if (c < 0 || c >= cols) continue;
                    const int ph = lg >= 0 ? (c & (sf - 1)) : c % sf, q = lg >= 0 ? (c >> lg) : c / sf;
                    tile[(ph * rows + r) * qpitch + q] = v[t];
                }
            }
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 15;
    // horizontal (strided) pass: 16 kept columns of every window row.  The two 16-lane halves of a 32-lane LDS group take rows r and r + sf:
    // with sf * qpitch = sf * hpp = 16 (mod 32) (chosen by the host) their reads of the window planes and their writes of the pass's output fall on
    // disjoint banks — consecutive rows collided on 12 of 32 banks (round-4 counters: 49 % of the LDS cycles were conflict cycles)
    const int half16 = (threadIdx.x >> 4) & 1, pi0 = threadIdx.x >> 5;         // pair index inside a step of 8 pairs
    const int npairs = (rows + 2 * sf - 1) / (2 * sf) * sf;
    for (int pi = pi0; pi < npairs; pi += 8) {
        const int blk = SFT ? pi / SFT : pi / sf;
        const int r = blk * 2 * sf + (pi - blk * sf) + half16 * sf;
        if (r >= rows) continue;
        // window column c of row r sits in plane c % sf at column c / sf
        const float* base = tile + r * qpitch + tx;
        const int pstride = rows * qpitch;
        if constexpr (SFT != 0) hp[r * hpp + tx] = tap_sum_phased<SFT>(th, k, base, pstride);
        else hp[r * hpp + tx] = tap_sum(th, k, [&](const int c) { const int q = c / sf; return base[(c - q * sf) * pstride + q]; });
    }
    __syncthreads();
    const int ty = threadIdx.x >> 4;
    const int i = i0 + ty, j = j0 + tx;
    if (i < h && j < w) {
        const float* col = hp + ty * sf * hpp + tx;
        float acc = tap_sum_strided(tv, k, col, hpp);
        if (lr) {
            const int h0 = h - 2 * lr_pad, w0 = w - 2 * lr_pad;
            acc = lr[(bc * h0 + clampi(i - lr_pad, 0, h0 - 1)) * w0 + clampi(j - lr_pad, 0, w0 - 1)] - acc;
        }
        d[(bc * h + i) * (long long)w + j] = acc;
    }
}

// Streaming form of the separable downscale (round 3).  The tile kernel above stages a ((DT-1) sf + k)^2 window per 16x16 outputs — 109 KB of
// LDS at sf = 8, k = 45: one workgroup per CU, every one of its seven load batches a full memory round trip (1.9 ms per configs[4] batch,
// 0.43 TB/s).  Here a workgroup owns a strip of 32 output columns x DS_ROWS output rows and walks down the high-resolution rows DS_STEP at a time
// (16: 0.52 ms at configs[4]; 8 rows 0.82, 24 rows 1.45, 32 rows 1.03 — profiles/r05_cem_stream_step.log):
// load DS_STEP window rows (coalesced 16-byte vectors, the next step's loads in flight under this step's arithmetic), horizontal pass (thread =
// (row, output column): k taps out of the de-interleaved row images) into a ring of the last RING h-rows, then every output row whose k
// h-rows are complete is emitted (vertical pass out of the ring).  ~25 KB of LDS: six workgroups per CU cover each other's latencies; the
// window overlap re-reads 1.1-1.2x of g instead of 1.7x.  Same index rules (clamped window = replicate padding, strided pick at `pre`),
// same fused  lr_pad - D(y)  output as the tile kernel; the sums run in the same order (horizontal a0/a1 pairs, then vertical).
constexpr int DS_COLS = 32, DS_ROWS = 64, DS_STEP = 16, DS_RPT = DS_STEP / 8, DS_SB = 5;      // window rows per step; rows per thread in the horizontal pass
template <int SFT>
__global__ __launch_bounds__(256) void cem_downscale_stream_kernel(const float* __restrict__ y, int h, int w, int sf_rt, int pre, const float* __restrict__ tv,
                                                                 const float* __restrict__ th, int k, const float* __restrict__ lr, int lr_pad,
                                                                 float* __restrict__ d, int qpitch, int ring) {
    extern __shared__ float tile[];          // [sf][DS_STEP][qpitch] de-interleaved window rows | ring [ring][DS_COLS + 1] of horizontal-pass rows
    const int sf = SFT ? SFT : sf_rt;
    const int p = k / 2, Hh = h * sf, Wh = w * sf;
    const int j0 = blockIdx.x * DS_COLS, i0 = blockIdx.y * DS_ROWS;
    const int nrow = min(DS_ROWS, h - i0);                   // output rows of this strip
    const long long bc = blockIdx.z;
    const float* src = y + bc * Hh * (long long)Wh;
    const int Yb = sf * i0 + pre - p, Xb = sf * j0 + pre - p;
    const int cols = (DS_COLS - 1) * sf + k;
    const int nwin = (nrow - 1) * sf + k;                    // window rows of the strip
    const int lg = (sf & (sf - 1)) == 0 ? __builtin_ctz(sf) : -1;
    float* const hp = tile + sf * DS_STEP * qpitch;
    // the taps out of LDS (broadcast reads): as global loads inside the tap loops they were a dependent L1 round trip per FMA, every step
    float* const tvs = hp + ring * (DS_COLS + 1);
    float* const ths = tvs + k;
    for (int e = threadIdx.x; e < k; e += 256) { tvs[e] = tv[e]; ths[e] = th[e]; }
    const int Xa = (Xb >> 2) << 2;                           // floor to a multiple of 4 (arithmetic shift: also for negative Xb)
    const int nvec = (Xb + cols - Xa + 3) >> 2;
    const bool vec_ok = (Wh & 3) == 0 && (((size_t)src) & 15) == 0;
    constexpr int SB = DS_SB;                                // 16-byte vectors per thread per step: DS_STEP * nvec <= 256 * SB (checked by the host)
    constexpr int PD = 2;                                    // steps of loads in flight (a step's arithmetic is much shorter than a memory round trip)
    // A thread's SB vectors sit at the same (row of the step, column) every step: the index arithmetic (two integer divisions, the clamps, the
    // de-interleaved LDS addresses of the four elements) is done ONCE — per step a vector costs one row clamp and one multiply.  (Recomputed
    // every step it was ~1000 VALU instructions per thread and step: the kernel was bound by its own address arithmetic.)
    int vr[SB], vx[SB], vdst[SB][4];
    bool vfast[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
        const int e = threadIdx.x + u * 256;
        const int r = e / nvec;
        vr[u] = r < DS_STEP ? r : -1;
        vx[u] = Xa + 4 * (e - r * nvec);
        vfast[u] = vec_ok && vx[u] >= 0 && vx[u] + 3 < Wh;
        const int c0 = vx[u] - Xb;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = c0 + t;
            const int ph = lg >= 0 ? (c & (sf - 1)) : c % sf, q = lg >= 0 ? (c >> lg) : c / sf;
            vdst[u][t] = (r < DS_STEP && c >= 0 && c < cols) ? (ph * DS_STEP + r) * qpitch + q : -1;
        }
    }
    float4 tmp[PD][SB];
    auto issue = [&](auto SLOT, const int step) {
        constexpr int S = decltype(SLOT)::value;
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            tmp[S][u] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int wr = step * DS_STEP + vr[u];
            if (vr[u] >= 0 && wr < nwin) {
                const int x = vx[u];
                const float* rowp = src + (long long)clampi(Yb + wr, 0, Hh - 1) * Wh;
                if (vfast[u]) tmp[S][u] = *(const float4*)(rowp + x);
                else tmp[S][u] = make_float4(rowp[clampi(x, 0, Wh - 1)], rowp[clampi(x + 1, 0, Wh - 1)], rowp[clampi(x + 2, 0, Wh - 1)], rowp[clampi(x + 3, 0, Wh - 1)]);
            }
        }
    };
    const int nsteps = (nwin + DS_STEP - 1) / DS_STEP;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    int next_i = 0;                                          // first output row not emitted yet
    auto body = [&](auto SLOT, const int step) {
        constexpr int S = decltype(SLOT)::value;
        // registers -> de-interleaved row images
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const float v[4] = {tmp[S][u].x, tmp[S][u].y, tmp[S][u].z, tmp[S][u].w};
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (vdst[u][t] >= 0) tile[vdst[u][t]] = v[t];
        }
        __syncthreads();
        if (step + PD < nsteps) issue(SLOT, step + PD);      // this slot's registers are free again: PD steps of loads stay in flight
        {   // horizontal (strided) pass: thread (ty, tx) -> window row step*8 + ty, output column tx
            // (taps in index order, eight LDS reads in flight per batch: as a nested run-time loop of dependent read -> FMA pairs the pass cost
            // an LDS round trip per tap — 2.7 us per step, more than the step's memory traffic)
            // each thread runs DS_RPT window rows (ty, ty + 8, ...) of its output column: a tap is read (and its phase / column computed) once
            // per DS_RPT products
            float acc[DS_RPT];
#pragma unroll
            for (int j = 0; j < DS_RPT; ++j) acc[j] = 0.f;
            const float* const pl = tile + ty * qpitch + tx;
#pragma unroll 4
            for (int c = 0; c < k; ++c) {
                const int ph = lg >= 0 ? (c & (sf - 1)) : c % sf, q = lg >= 0 ? (c >> (lg >= 0 ? lg : 0)) : c / sf;
                const float tc = ths[c];
                const float* const pc = pl + ph * DS_STEP * qpitch + q;
#pragma unroll
                for (int j = 0; j < DS_RPT; ++j) acc[j] = fmaf(tc, pc[j * 8 * qpitch], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < DS_RPT; ++j)
                hp[((step * DS_STEP + ty + 8 * j) & (ring - 1)) * (DS_COLS + 1) + tx] = acc[j];
        }
        __syncthreads();
        // vertical pass: output rows whose last window row (i sf + k - 1) is inside the rows done so far; up to DS_STEP of them per step
        const int done = min((step + 1) * DS_STEP, nwin);
        int last_i = (done - k) >= 0 ? (done - k) / sf : -1;
        if (last_i > nrow - 1) last_i = nrow - 1;
        for (int ib = next_i; ib <= last_i; ib += 8) {
            const int il = ib + ty;
            const int i = i0 + il, j = j0 + tx;
            if (il <= last_i && j < w) {
                float a0 = 0.f, a1 = 0.f;
                const int r0 = il * sf;
#pragma unroll 8
                for (int a = 0; a < k; ++a) {
                    const float v = tvs[a] * hp[((r0 + a) & (ring - 1)) * (DS_COLS + 1) + tx];
                    if (a & 1) a1 += v; else a0 += v;
                }
                float acc = a0 + a1;
                if (lr) {
                    const int h0 = h - 2 * lr_pad, w0 = w - 2 * lr_pad;
                    acc = lr[(bc * h0 + clampi(i - lr_pad, 0, h0 - 1)) * w0 + clampi(j - lr_pad, 0, w0 - 1)] - acc;
                }
                d[(bc * h + i) * (long long)w + j] = acc;
            }
        }
        if (last_i >= next_i) next_i = last_i + 1;
        __syncthreads();                                     // the ring rows read above may be overwritten two steps from now; the tile right away
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    static_assert(PD == 2, "two register slots");
    issue(I0{}, 0);
    if (1 < nsteps) issue(I1{}, 1);
    for (int step = 0; step < nsteps; step += PD) {
        body(I0{}, step);
        if (step + 1 < nsteps) body(I1{}, step + 1);
    }
}

// Polyphase upscale, separable: the vertical pass combines, for every output ROW of the tile, the <= ceil(k/sf) window rows whose taps land on
// samples (plus the pre == 0 replicate rule) into one row of window-column values; the horizontal pass does the same along the row.
// Tile of the separable upscale: 64 x 64 high-resolution pixels, a thread owns one quad of columns in FOUR rows (16 rows apart).  (16 x 64 with one
// quad per thread until round 5: the polyphase index arithmetic of a quad — which taps hit samples, where they start in the window — depends on
// the column only and is now formed once per thread and serves four rows; a workgroup's latency chain load -> pass -> pass -> store moves four
// times the pixels.)
constexpr int US_Y = 64, US_X = 64, US_RPT = US_Y / 16;
// FILT (round 6): the LR filter in front of the upscale — out = U(K(f)) instead of U(f) — folded into this kernel.  K = inv_hTh is a kf-tap
// separable filter on the LR grid with replicate padding (cem_lrfilter_sep_kernel: 27 taps for the x4 bicubic CEM): a tile's window of K(f)
// ((US + k) / sf + 3 = 23 LR pixels a side at x4) needs f on a (23 + kf - 1)^2 window — staged with clamped indices, filtered in LDS with the
// separate kernel's two passes and summation order (bit-identical to it: the 64 x 16 tiles of that kernel and these windows evaluate the same
// FMA sequence per LR pixel), 45 k multiply-adds per tile and plane.  What it removes: a launch of its own (37.5 us of latency at configs[1],
// 17 MB written and read back) and an LR-sized intermediate per projection.
template <bool TWO, int SFT, bool FILT = false>
__global__ __launch_bounds__(256) void cem_upscale_sep_kernel(const float* __restrict__ f, const float* __restrict__ f2, int h, int w, int sf_rt, int pre,
                                                            const float* __restrict__ tv, const float* __restrict__ th, int k, const float* __restrict__ g,
                                                            int crop, int mode, float range, float* __restrict__ out, float* __restrict__ out2, int wr, int wc, int vp,
                                                            const float* __restrict__ tvf, const float* __restrict__ thf, int kf, int ep, int hpp) {
    extern __shared__ float sm[];             // window 1 [wr][wc] | window 2 | vertical pass 1 [US_Y][vp] | vertical pass 2 | tv[k] | th[k] | FILT: raw window [wr + kf - 1][ep] | its horizontal pass [wr + kf - 1][hpp]   (vp = 16 mod 32: two rows of a 32-lane group on disjoint banks)
    float* const w1 = sm;
    float* const w2 = w1 + wr * wc;
    float* const v1 = w2 + (TWO ? wr * wc : 0);
    float* const v2 = v1 + US_Y * vp;
    const int sf = SFT ? SFT : sf_rt;
    const int Hh = h * sf, Wh = w * sf, Ho = Hh - 2 * crop, Wo = Wh - 2 * crop, p = k / 2;
    const TileId id = xcd_tile();
    const long long bc = id.bz;
    const int xo0 = id.bx * US_X, yo0 = id.by * US_Y;
    // the taps in LDS: both passes index them per lane (the polyphase offset depends on the output row / column) — through global memory that
    // was one vector load per tap and output
    float* const tvs = v2 + (TWO ? US_Y * vp : 0);
    float* const ths = tvs + k;
    for (int e = threadIdx.x; e < 2 * k; e += 256) tvs[e] = e < k ? tv[e] : th[e - k];
    // (what the horizontal pass's thread needs of g — 4 consecutive output columns of its rows — is requested NOW: its memory round trip runs
    // under the window staging and both passes instead of after them)
    static_assert(16 * (US_X / 4) == 256, "one quad of output columns per thread and row step");
    const int ry = threadIdx.x / (US_X / 4), xo = xo0 + 4 * (threadIdx.x % (US_X / 4));
    const bool full = xo + 3 < Wo;
    float gv[US_RPT][4];
    bool vec_g[US_RPT];
#pragma unroll
    for (int q = 0; q < US_RPT; ++q) {
        const int yq = yo0 + ry + 16 * q;
        const bool live = xo < Wo && yq < Ho;
        const float* gp = (mode >= 1 && live) ? g + (bc * Hh + yq + crop) * (long long)Wh + xo + crop : nullptr;
        vec_g[q] = live && full && mode >= 1 && ((((size_t)gp) & 15) == 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) gv[q][t] = 0.f;
        if (vec_g[q]) { const float4 t4 = *(const float4*)gp; gv[q][0] = t4.x; gv[q][1] = t4.y; gv[q][2] = t4.z; gv[q][3] = t4.w; }
        else if (mode >= 1 && live) {
#pragma unroll
            for (int t = 0; t < 4; ++t) if (xo + t < Wo) gv[q][t] = gp[t];
        }
    }
    const int fy = yo0 + crop - p - pre, fx = xo0 + crop - p - pre;
    const int ib = (fy >= 0 ? fy / sf : -((-fy + sf - 1) / sf)), jb = (fx >= 0 ? fx / sf : -((-fx + sf - 1) / sf));
    const float* s1 = f + bc * h * (long long)w;
    const float* s2 = TWO ? f2 + bc * h * (long long)w : nullptr;
    if constexpr (FILT) {
        float* const eraw = ths + k;
        float* const hp = eraw + (wr + kf - 1) * ep;
        const int pf = kf / 2, er = wr + kf - 1, ec = wc + kf - 1;
        for (int win = 0; win < (TWO ? 2 : 1); ++win) {
            const float* const src = win ? s2 : s1;
            float* const wdst = win ? w2 : w1;
            // the raw window, replicate-padded by clamping (the LR filter's rule); SB independent loads per thread in flight together
            constexpr int SB = 8;
            for (int e0 = threadIdx.x; e0 < er * ec; e0 += 256 * SB) {
                float tmp[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int e = e0 + u * 256, r = e / ec, c = e - r * ec;
                    tmp[u] = e < er * ec ? src[(long long)clampi(ib - pf + r, 0, h - 1) * w + clampi(jb - pf + c, 0, w - 1)] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int e = e0 + u * 256, r = e / ec, c = e - r * ec;
                    if (e < er * ec) eraw[r * ep + c] = tmp[u];
                }
            }
            __syncthreads();
            for (int e = threadIdx.x; e < er * wc; e += 256) {           // horizontal pass (cem_lrfilter_sep_kernel's sums: even taps, odd taps)
                const int r = e / wc, c = e - r * wc;
                const float* row = eraw + r * ep + c;
                float a0 = 0.f, a1 = 0.f;
                int cc = 0;
#pragma unroll 4
                for (; cc + 1 < kf; cc += 2) { a0 = fmaf(thf[cc], row[cc], a0); a1 = fmaf(thf[cc + 1], row[cc + 1], a1); }
                if (cc < kf) a0 = fmaf(thf[cc], row[cc], a0);
                hp[r * hpp + c] = a0 + a1;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < wr * wc; e += 256) {           // vertical pass; samples outside the LR image do not exist for the upscale: zero
                const int r = e / wc, c = e - r * wc;
                const float* col = hp + r * hpp + c;
                float a0 = 0.f, a1 = 0.f;
                int a = 0;
#pragma unroll 4
                for (; a + 1 < kf; a += 2) { a0 = fmaf(tvf[a], col[a * hpp], a0); a1 = fmaf(tvf[a + 1], col[(a + 1) * hpp], a1); }
                if (a < kf) a0 = fmaf(tvf[a], col[a * hpp], a0);
                const int i = ib + r, j = jb + c;
                wdst[e] = (i >= 0 && i < h && j >= 0 && j < w) ? a0 + a1 : 0.f;
            }
            __syncthreads();
        }
    } else {
        for (int e = threadIdx.x; e < wr * wc; e += 256) {
            const int r = e / wc, c = e - r * wc;
            const int i = ib + r, j = jb + c;
            const bool in = i >= 0 && i < h && j >= 0 && j < w;
            w1[e] = in ? s1[(long long)i * w + j] : 0.f;
            if (TWO) w2[e] = in ? s2[(long long)i * w + j] : 0.f;
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < US_Y * wc; e += 256) {      // vertical pass
        const int ry1 = e / wc, c = e - ry1 * wc;
        const int Y1 = yo0 + ry1 + crop;
        float u1 = 0.f, u2 = 0.f;
        if (Y1 < Hh) {
            int a0 = (pre + p - Y1) % sf; if (a0 < 0) a0 += sf;
            int il = (Y1 + a0 - p - pre) / sf - ib;
            for (int a = a0; a < k; a += sf, ++il) {
                const int yy = Y1 + a - p;
                if (yy < 0 || yy >= Hh) continue;
                const float ta = tvs[a];
                u1 = fmaf(ta, w1[il * wc + c], u1);
                if (TWO) u2 = fmaf(ta, w2[il * wc + c], u2);
            }
            if (pre == 0)
                for (int a = 0; Y1 + a - p < 0 && a < k; ++a) {
                    u1 = fmaf(tvs[a], w1[-ib * wc + c], u1);
                    if (TWO) u2 = fmaf(tvs[a], w2[-ib * wc + c], u2);
                }
        }
        v1[ry1 * vp + c] = u1;
        if (TWO) v2[ry1 * vp + c] = u2;
    }
    __syncthreads();
    // horizontal pass: a thread owns 4 consecutive output columns in US_RPT rows.  Per column: the first tap that lands on a sample, the window
    // column it meets there and how many taps follow inside the image — once; then every row is that many multiply-adds out of its pass-1 row.
    // g comes in and the result goes out as 16-byte vectors when the rows are 16-byte aligned (else element by element).
    if (xo >= Wo) return;
    int b0[4], jl0[4], nb[4], nrep[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int X = xo + t + crop;
        int b = (pre + p - X) % sf; if (b < 0) b += sf;
        int jl = (X + b - p - pre) / sf - jb;
        while (b < k && X + b - p < 0) { b += sf; ++jl; }          // taps left of the image: dropped (zero-stuffed image, no sample there)
        int n = 0;
        for (int bb = b; bb < k && X + bb - p < Wh; bb += sf) ++n;
        b0[t] = b; jl0[t] = jl; nb[t] = n;
        int nr = 0;                                                  // pre == 0: the taps that reach left of the image replicate the first sample
        if (pre == 0) while (nr < k && X + nr - p < 0) ++nr;
        nrep[t] = nr;
    }
#pragma unroll
    for (int q = 0; q < US_RPT; ++q) {
        const int yq = yo0 + ry + 16 * q;
        if (yq >= Ho) continue;
        const long long idx = (bc * Ho + yq) * (long long)Wo + xo;
        const bool vec_o = full && (((size_t)(out + idx)) & 15) == 0 && (mode != 3 || (((size_t)(out2 + idx)) & 15) == 0);
        const float* r1 = v1 + (ry + 16 * q) * vp;
        const float* r2 = v2 + (ry + 16 * q) * vp;
        float res[4], res2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float u1 = 0.f, u2 = 0.f;
            for (int i = 0; i < nb[t]; ++i) {
                const float tb = ths[b0[t] + i * sf];
                u1 = fmaf(tb, r1[jl0[t] + i], u1);
                if (TWO) u2 = fmaf(tb, r2[jl0[t] + i], u2);
            }
            for (int b = 0; b < nrep[t]; ++b) {
                u1 = fmaf(ths[b], r1[-jb], u1);
                if (TWO) u2 = fmaf(ths[b], r2[-jb], u2);
            }
            res2[t] = 0.f;
            if (mode == 0) res[t] = u1;
            else if (mode == 1) res[t] = gv[q][t] + u1;
            else if (mode == 2) res[t] = u1 + tanhf(gv[q][t] - u2) * range;
            else { res[t] = u1; res2[t] = gv[q][t] - u2; }
        }
        if (vec_o) {
            *(float4*)(out + idx) = make_float4(res[0], res[1], res[2], res[3]);
            if (mode == 3) *(float4*)(out2 + idx) = make_float4(res2[0], res2[1], res2[2], res2[3]);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (xo + t < Wo) {
                    out[idx + t] = res[t];
                    if (mode == 3) out2[idx + t] = res2[t];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Wave-streaming separable kernels (round 6).  The tile kernels above spend ~1100 (downscale) and ~100 (upscale) vector instructions per output
// pixel — staging index arithmetic, one LDS read per multiply-add — and sit at 74 % of the chip's VALU issue slots while moving 1.9-2.9 TB/s
// (profiles/r06_cem_pmc.json): bound by their own instruction stream, not by HBM.  Here a WAVE owns a strip of the image and walks down its rows
// with everything that depends on the column formed once: no workgroup barrier, no per-row index arithmetic, the vertical pass in registers.
//
// Downscale: lane l holds 4 adjacent window columns (one aligned 16-byte load per high-resolution row).  A window row sf m + ph feeds output rows
// m - t with tap sf t + ph (t < NA = ceil(k / sf)): NA running sums per column, the oldest one complete at the end of every block of sf rows — it is
// shifted out, passed through this wave's LDS image (de-interleaved by phase: the horizontal pass reads consecutive words), and lane j < ncol
// finishes output column j with k multiply-adds.  Taps are padded with zeros to NA sf entries so that every LDS offset and tap index of the unrolled
// loops is an immediate.  Same index rules as the tile kernels (clamped window = replicate padding, strided pick at `pre`, fused  lr_pad - D(y)).
// Sums run in tap order in ONE accumulator per pass (the tile kernels use four / two partial sums): results agree to fp32 rounding, not to the bit.
constexpr int WV_LANES = 64;
template <int SFT, int NA>
__global__ __launch_bounds__(256) void cem_downscale_wave_kernel(const float* __restrict__ y, int h, int w, int pre, const float* __restrict__ tv,
                                                               const float* __restrict__ th, int k, const float* __restrict__ lr, int lr_pad,
                                                               float* __restrict__ d, int nout, int R, int nct, int nst, long long nitems) {
    constexpr int sf = SFT, NT = NA * SFT;
    constexpr int QP = (256 / SFT + 1) > (WV_LANES + NA) ? (256 / SFT + 1) : (WV_LANES + NA);      // words per phase plane of a wave's row image
    __shared__ float rows_lds[4][SFT * QP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* const L = rows_lds[wv];
    // strips in (plane, column tile, row strip) order, four consecutive ones per workgroup; XCD x (workgroups g with g & 7 == x) sweeps a contiguous run
    const unsigned nwg = gridDim.x, g = blockIdx.x, xcd = g & 7, q = nwg >> 3, rr = nwg & 7;
    const long long item = (long long)(xcd * q + (xcd < rr ? xcd : rr) + (g >> 3)) * 4 + wv;
    if (item >= nitems) return;
    const int st = (int)(item % nst);
    const long long t2 = item / nst;
    const int ct = (int)(t2 % nct);
    const long long bc = t2 / nct;
    const int p = k / 2, Hh = h * sf, Wh = w * sf;
    const int j0 = ct * nout, i0 = st * R;
    const int nrow = min(R, h - i0), ncol = min(nout, w - j0);
    const int Yb = sf * i0 + pre - p, Xb = sf * j0 + pre - p;
    const int Xa = (Xb >> 2) << 2, off = Xb - Xa;             // the lane's vector starts at a multiple of 4: window column c = 4 lane + t - off
    const int cols = (ncol - 1) * sf + k;                     // window columns this strip needs
    const float* const src = y + bc * Hh * (long long)Wh;
    const int x = Xa + 4 * lane;
    const bool need = 4 * lane - off < cols && 4 * lane + 3 - off >= 0;
    const bool fast = (Wh & 3) == 0 && (((size_t)src) & 15) == 0 && x >= 0 && x + 3 < Wh;
    int xc[4], slot[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        xc[t] = clampi(x + t, 0, Wh - 1);
        const int c = 4 * lane + t - off;
        slot[t] = (c >= 0 && c < cols) ? (c % sf) * QP + c / sf : -1;
    }
    // Odd strips walk UP: a strip and its lower neighbour then finish on their shared window rows at the same time, a strip and its upper neighbour
    // start on theirs together — the k - sf rows two strips share are in the XCD's L2 when the second one asks (with all strips walking down the second
    // request came a whole strip later: 1.36 x the image from HBM).  Walking up = the same loop over the mirrored window with the vertical taps
    // reversed; an output row of an odd strip sums its vertical taps in descending order (fp32 rounding) — which is why strips are cut by the image
    // height alone (esr_cem_downscale_sep): every output's arithmetic is the same whatever the batch.
    const bool up = __builtin_amdgcn_readfirstlane(st & 1) != 0;         // (scalar: the tap loads below stay scalar loads)
    float tvv[NT], thv[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const int av = up ? NT - 1 - a : a;
        tvv[a] = av < k ? tv[av < k ? av : 0] : 0.f;
        thv[a] = a < k ? th[a] : 0.f;
    }
    // (the zero taps of the padded horizontal pass multiply image words past the strip's last window column: they have to be finite)
    for (int e = lane; e < SFT * QP; e += 64) L[e] = 0.f;
    const int nwin = (nrow - 1) * sf + k;                     // window rows of the strip
    const int nblk = nrow + NA - 1;                           // blocks of sf window rows (the last ones past nwin: zeros)
    float acc[NA][4];
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = 0.f;
    float4 cur[SFT], nxt[SFT];
    auto load_blk = [&](float4 (&dst)[SFT], const int m) {
#pragma unroll
        for (int ph = 0; ph < sf; ++ph) {
            const int wr = up ? sf * nblk - 1 - (sf * m + ph) : sf * m + ph;
            dst[ph] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (wr < nwin && need) {
                const float* const rowp = src + (long long)clampi(Yb + wr, 0, Hh - 1) * Wh;
                if (fast) dst[ph] = *(const float4*)(rowp + x);
                else dst[ph] = make_float4(rowp[xc[0]], rowp[xc[1]], rowp[xc[2]], rowp[xc[3]]);
            }
        }
    };
    const int h0 = h - 2 * lr_pad, w0 = w - 2 * lr_pad;
    load_blk(cur, 0);
    for (int m = 0; m < nblk; ++m) {
        if (m + 1 < nblk) load_blk(nxt, m + 1);               // the next block's rows are in flight under this block's arithmetic
#pragma unroll
        for (int ph = 0; ph < sf; ++ph) {
            const float v[4] = {cur[ph].x, cur[ph].y, cur[ph].z, cur[ph].w};
#pragma unroll
            for (int t = 0; t < NA; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[t][c] = fmaf(tvv[sf * t + ph], v[c], acc[t][c]);
        }
        const int il = m - (NA - 1);                          // the output row whose last window row was in this block
        if (il >= 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (slot[c] >= 0) L[slot[c]] = acc[NA - 1][c];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (a wave's LDS operations execute in order: written, then read, by the same wave)
            if (lane < ncol) {
                float o = 0.f;
#pragma unroll
                for (int b = 0; b < NT; ++b) o = fmaf(thv[b], L[(b % sf) * QP + lane + b / sf], o);
                const int i = i0 + (up ? nrow - 1 - il : il), j = j0 + lane;
                if (lr) o = lr[(bc * h0 + clampi(i - lr_pad, 0, h0 - 1)) * w0 + clampi(j - lr_pad, 0, w0 - 1)] - o;
                d[(bc * h + i) * (long long)w + j] = o;
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int t = NA - 1; t > 0; --t)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][c] = acc[t - 1][c];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[0][c] = 0.f;
#pragma unroll
        for (int ph = 0; ph < sf; ++ph) cur[ph] = nxt[ph];
    }
}

// Upscale (polyphase, zero-stuffed samples at sf i + pre, pre > 0): lane l owns 4 adjacent OUTPUT columns — one 16-byte load of g and one 16-byte store
// per output row — and, for the vertical pass, low-resolution column jb + l (and jb + 64 + l where the wave's 256 output columns reach that far).
//   vertical pass   output rows are walked in blocks of sf rows aligned to the sample grid: row r of a block uses the taps sf t (r = 0) or
//                   sf - r + sf t (r > 0) on NA consecutive sample rows — a window of NA + 1 sample values per lane in registers, one new value per block
//                   (requested a block ahead); the pass-1 value of every sample column goes to this wave's LDS row;
//   horizontal pass per lane, once: which NV = NA + 1 consecutive pass-1 values its four columns meet and, per column, the NV taps that go with them
//                   (zeros where a column's taps start one sample later or end at k); per row NV LDS reads and 4 NV multiply-adds, g added, stored.
// No workgroup barrier, no index arithmetic per row.  Modes and index rules as cem_upscale_sep_kernel; sums in tap order in one accumulator per pass
// (the tile kernel's order as well).
template <bool TWO, int SFT, int NA>
__global__ __launch_bounds__(256) void cem_upscale_wave_kernel(const float* __restrict__ f, const float* __restrict__ f2, int h, int w, int pre,
                                                             const float* __restrict__ tv, const float* __restrict__ th, int k, const float* __restrict__ g,
                                                             int crop, int mode, float range, float* __restrict__ out, float* __restrict__ out2, int S, int tcols,
                                                             int nct, int nst, long long nitems) {
    constexpr int sf = SFT, NV = NA + 1, NIN = TWO ? 2 : 1;
    constexpr int NT1 = 252 / SFT + NV + 2;                   // pass-1 values a wave's 256 output columns can meet
    constexpr int NP = (NT1 + 63) / 64;                       // vertical passes per row (sample columns per lane)
    __shared__ float t_lds[4][NIN][NP * 64 + 8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned nwg = gridDim.x, gi = blockIdx.x, xcd = gi & 7, q = nwg >> 3, rr = nwg & 7;
    const long long item = (long long)(xcd * q + (xcd < rr ? xcd : rr) + (gi >> 3)) * 4 + wv;
    if (item >= nitems) return;
    const int st = (int)(item % nst);
    const long long t2 = item / nst;
    const int ct = (int)(t2 % nct);
    const long long bc = t2 / nct;
    const int p = k / 2, Hh = h * sf, Wh = w * sf, Ho = Hh - 2 * crop, Wo = Wh - 2 * crop;
    const int xo0 = ct * tcols, yq0 = st * S;
    const int nrows = min(S, Ho - yq0);
    const int xo = xo0 + 4 * lane;
    const bool live = 4 * lane < tcols && xo < Wo;             // this lane stores something
    auto cdiv = [](const int a, const int b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); };      // ceil(a / b), b > 0
    // sample column met by the first tap of output column X that lands on a sample: ceil((X - p - pre) / sf)
    const int jb = cdiv(xo0 + crop - p - pre, sf);
    const int X0 = xo + crop;
    const int base = cdiv(X0 - p - pre, sf) - jb;             // first pass-1 value of this lane's columns, relative to the wave's first
    float tq[4][NV];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int X = X0 + t;
        int b0 = (pre + p - X) % sf; if (b0 < 0) b0 += sf;
        const int dt = cdiv(X - p - pre, sf) - jb - base;     // 0 or 1
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int u = v - dt, b = b0 + sf * u;
            tq[t][v] = (u >= 0 && b < k && live) ? th[b < k ? (b >= 0 ? b : 0) : 0] : 0.f;
        }
    }
    // vertical taps per row of a block (uniform): tvr[r][t]
    float tvr[SFT][NA];
#pragma unroll
    for (int r = 0; r < sf; ++r)
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int a = (r == 0 ? 0 : sf - r) + sf * t;
            tvr[r][t] = a < k ? tv[a] : 0.f;
        }
    const float* const s1 = f + bc * h * (long long)w;
    const float* const s2 = TWO ? f2 + bc * h * (long long)w : nullptr;
    const int Y0 = yq0 + crop;                                // first output row of the strip, un-cropped coordinates
    int ph0 = (Y0 - p - pre) % sf; if (ph0 < 0) ph0 += sf;
    int Yblk = Y0 - ph0;                                      // block start: Yblk = p + pre (mod sf)
    int ibb = (Yblk - p - pre) / sf;                          // (exact; may be negative)
    const int nblk = (Y0 + nrows - Yblk + sf - 1) / sf;
    // sample window: e[in][pass][0 .. NA] = rows ibb .. ibb + NA of this lane's sample column(s); e[..][NA + 1] = the next block's new row
    float e[NIN][NP][NA + 2];
    int jcol[NP];
    bool jok[NP];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) { jcol[pp] = jb + lane + 64 * pp; jok[pp] = jcol[pp] >= 0 && jcol[pp] < w && lane + 64 * pp < NT1; }
    auto sample = [&](const int in, const int pp, const int i) -> float {
        if (!jok[pp] || i < 0 || i >= h) return 0.f;
        return (in ? s2 : s1)[(long long)i * w + jcol[pp]];
    };
#pragma unroll
    for (int in = 0; in < NIN; ++in)
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int v = 0; v < NA + 1; ++v) e[in][pp][v] = sample(in, pp, ibb + v);
    const bool gvec = mode >= 1 && live && xo + 3 < Wo && (Wh & 3) == 0 && (((size_t)(g + bc * Hh * (long long)Wh + X0)) & 15) == 0;
    const bool ovec = live && xo + 3 < Wo && (Wo & 3) == 0 && (((size_t)(out + bc * Ho * (long long)Wo + xo)) & 15) == 0 &&
                      (mode != 3 || (((size_t)(out2 + bc * Ho * (long long)Wo + xo)) & 15) == 0);
    float* const T1 = t_lds[wv][0];
    float* const T2 = t_lds[wv][NIN - 1];
    for (int blk = 0; blk < nblk; ++blk, Yblk += sf, ++ibb) {
        // next block's new sample row, and this block's rows of g: all requested before the arithmetic
#pragma unroll
        for (int in = 0; in < NIN; ++in)
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) e[in][pp][NA + 1] = blk + 1 < nblk ? sample(in, pp, ibb + NA + 1) : 0.f;
        float gv[SFT][4];
#pragma unroll
        for (int r = 0; r < sf; ++r) {
            const int Y = Yblk + r;
#pragma unroll
            for (int t = 0; t < 4; ++t) gv[r][t] = 0.f;
            if (mode >= 1 && live && Y >= Y0 && Y < Y0 + nrows) {
                const float* const gp = g + (bc * Hh + Y) * (long long)Wh + X0;
                if (gvec) { const float4 t4 = *(const float4*)gp; gv[r][0] = t4.x; gv[r][1] = t4.y; gv[r][2] = t4.z; gv[r][3] = t4.w; }
                else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (xo + t < Wo) gv[r][t] = gp[t];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < sf; ++r) {
            const int Y = Yblk + r;
            if (Y < Y0 || Y >= Y0 + nrows) continue;          // (uniform)
            // vertical pass: this row's value at every sample column of the wave
#pragma unroll
            for (int in = 0; in < NIN; ++in)
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    float u = 0.f;
#pragma unroll
                    for (int t = 0; t < NA; ++t) u = fmaf(tvr[r][t], e[in][pp][t + (r == 0 ? 0 : 1)], u);
                    (in ? T2 : T1)[lane + 64 * pp] = u;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (a wave's LDS operations execute in order)
            if (live) {
                float r1[NV], r2[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) { r1[v] = T1[base + v]; r2[v] = TWO ? T2[base + v] : 0.f; }
                float res[4], res2[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float u1 = 0.f, u2 = 0.f;
#pragma unroll
                    for (int v = 0; v < NV; ++v) { u1 = fmaf(tq[t][v], r1[v], u1); if (TWO) u2 = fmaf(tq[t][v], r2[v], u2); }
                    // (as selects, not as cem_upscale_sep_kernel's if-chain over two result arrays: hipcc 7.2 compiled that form of THIS kernel to a mode-3
                    // `out` of stale register contents — found by the comparison with the tile kernel, tools/experiments/cem_wave_ab.py)
                    const float gd = gv[r][t] - u2;                 // (modes 2 and 3; TWO only)
                    res[t] = mode == 1 ? gv[r][t] + u1 : (mode == 2 ? u1 + tanhf(gd) * range : u1);
                    res2[t] = gd;
                }
                const long long idx = (bc * Ho + (Y - crop)) * (long long)Wo + xo;
                if (ovec) {
                    *(float4*)(out + idx) = make_float4(res[0], res[1], res[2], res[3]);
                    if (mode == 3) *(float4*)(out2 + idx) = make_float4(res2[0], res2[1], res2[2], res2[3]);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (xo + t < Wo) {
                            out[idx + t] = res[t];
                            if (mode == 3) out2[idx + t] = res2[t];
                        }
                }
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int in = 0; in < NIN; ++in)
#pragma unroll
            for (int pp = 0; pp < NP; ++pp)
#pragma unroll
                for (int v = 0; v < NA + 1; ++v) e[in][pp][v] = e[in][pp][v + 1];
    }
}

// LR filter (K = inv_hTh: 27 or 35 taps a side, replicate padding), wave-streaming: lane l owns window columns 2 l, 2 l + 1 of the wave's 128 — a pair of
// running sums per tap in registers (packed fp32 multiply-adds): window row r feeds output rows r - t with tap t, the oldest pair is complete after every
// row, goes through the wave's LDS row and lane l finishes output columns 2 l, 2 l + 1 with K multiply-adds each out of 14 (18) 8-byte LDS reads.
// 128 - (K - 1) output columns per wave; strips of R output rows start K - 1 rows early.  Vertical sums in tap order, horizontal sums as even + odd taps (the tile
// kernel: horizontal pass first): the two forms agree to fp32 rounding.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int K>
__global__ __launch_bounds__(256) void cem_lrfilter_wave_kernel(const float* __restrict__ x, int h, int w, const float* __restrict__ tv, const float* __restrict__ th,
                                                              float* __restrict__ out, int nout, int R, int nct, int nst, long long nitems) {
    __shared__ float v_lds[4][128 + 4];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (uniform: strip, tile and plane indices — and the taps — stay scalar: 21 -> 17 us; the same line costs the downscale kernel 6 us in scalar-register spills)
    float* const V = v_lds[wv];
    const unsigned nwg = gridDim.x, g = blockIdx.x, xcd = g & 7, q = nwg >> 3, rr = nwg & 7;
    const long long item = (long long)(xcd * q + (xcd < rr ? xcd : rr) + (g >> 3)) * 4 + wv;
    if (item >= nitems) return;
    const int st = (int)(item % nst);
    const long long t2 = item / nst;
    const int ct = (int)(t2 % nct);
    const long long bc = t2 / nct;
    constexpr int p = K / 2;
    const int j0 = ct * nout, i0 = st * R;
    const int nrow = min(R, h - i0), ncol = min(nout, w - j0);
    const float* const src = x + bc * h * (long long)w;
    const int cw = 2 * lane;                                  // this lane's window columns cw, cw + 1
    const bool need = cw < ncol + K - 1;
    const int xa = clampi(j0 - p + cw, 0, w - 1), xb = clampi(j0 - p + cw + 1, 0, w - 1);
    float tvv[K], thv[K];
#pragma unroll
    for (int a = 0; a < K; ++a) { tvv[a] = tv[a]; thv[a] = th[a]; }
    f32x2 acc[K];
#pragma unroll
    for (int t = 0; t < K; ++t) acc[t] = f32x2{0.f, 0.f};
    const int nwin = nrow + K - 1;
    auto load_row = [&](const int wr) -> f32x2 {
        f32x2 v = f32x2{0.f, 0.f};
        if (wr < nwin && need) {
            const float* const rowp = src + (long long)clampi(i0 - p + wr, 0, h - 1) * w;
            v.x = rowp[xa]; v.y = rowp[xb];
        }
        return v;
    };
    constexpr int PF = 4;                                     // rows of loads in flight
    f32x2 pf[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) pf[u] = load_row(u);
    for (int wr0 = 0; wr0 < nwin; wr0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int wr = wr0 + u;
            const f32x2 v = pf[u];
            pf[u] = load_row(wr + PF);
            if (wr >= nwin) continue;                          // (uniform)
#pragma unroll
            for (int t = 0; t < K; ++t) acc[t] = acc[t] + tvv[t] * v;      // (contracted to packed fused multiply-adds)
            const int il = wr - (K - 1);
            if (il >= 0) {
                if (need) *(f32x2*)(V + cw) = acc[K - 1];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (a wave's LDS operations execute in order)
                if (cw < ncol) {
                    f32x2 r[K / 2 + 1];
#pragma unroll
                    for (int b = 0; b < K / 2 + 1; ++b) r[b] = *(const f32x2*)(V + cw + 2 * b);
                    // even and odd taps in their own sums (two dependent chains of K / 2 per output instead of one of K: this pass is the wave's latency)
                    float o0[2] = {0.f, 0.f}, o1[2] = {0.f, 0.f};
#pragma unroll
                    for (int b = 0; b < K; ++b) {
                        const float va = (b & 1) ? r[b / 2].y : r[b / 2].x;                       // V[cw + b]
                        const float vb = ((b + 1) & 1) ? r[(b + 1) / 2].y : r[(b + 1) / 2].x;     // V[cw + 1 + b]
                        o0[b & 1] = fmaf(thv[b], va, o0[b & 1]);
                        o1[b & 1] = fmaf(thv[b], vb, o1[b & 1]);
                    }
                    float* const op = out + (bc * h + i0 + il) * (long long)w + j0 + cw;
                    op[0] = o0[0] + o0[1];
                    if (cw + 1 < ncol) op[1] = o1[0] + o1[1];
                }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int t = K - 1; t > 0; --t) acc[t] = acc[t - 1];
            acc[0] = f32x2{0.f, 0.f};
        }
    }
}

}  // namespace

// Strips (waves) of a wave-streaming launch: ONE resident round of the chip.  A strip count just above what the chip holds at once costs a second,
// nearly empty round (configs[1] upscale: 4224 strips on 4096 wave slots 50 us, 3648 strips 39 us); fewer, longer strips lose memory parallelism
// (2112: 46 us).  Slots = CUs x 4 workgroups-per-CU-by-occupancy x 4 waves, asked once per kernel; the launch aims at 0.9 of them.
static int wave_strips_target(const void* kernel) {
    if (g_cem_wave_target > 0) return g_cem_wave_target;        // (instrumented build: esr_debug_cem_wave)
    struct Entry { const void* k; int dev; int slots; };
    static thread_local Entry cache[8];
    static thread_local int next = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (const Entry& e : cache)
        if (e.k == kernel && e.dev == dev && e.slots) return e.slots;
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || cus <= 0 || per_cu <= 0) {
        (void)hipGetLastError();
        return 3072;
    }
    const int slots = (int)(0.9 * cus * per_cu * 4);
    cache[next] = Entry{kernel, dev, slots};
    next = (next + 1) % 8;
    return slots;
}

// which form of the separable kernels an image geometry takes (esr_cem_sep_form): decided from (sf, k, pre, h, w) alone, never from the batch
static bool downscale_wave_ok(int sf, int k, int h, int w) {
    const int na = (k + sf - 1) / sf;
    return g_cem_wave && (sf == 2 || sf == 3 || sf == 4 || sf == 8) && na >= 4 && na <= 6 && (long long)h * w >= 4096;
}
static bool upscale_wave_ok(int sf, int k, int pre, int h, int w) {
    // pre > 0: no sample on the first row / column of the zero-stuffed image (sf = 2 keeps the tile kernel and its replicate rule)
    const int na = (k + sf - 1) / sf;
    return g_cem_wave && (sf == 3 || sf == 4 || sf == 8) && pre > 0 && na >= 4 && na <= 6 && (long long)h * w >= 4096;
}
static bool lrfilter_wave_ok(int k, int h, int w) { return g_cem_wave && (k == 27 || k == 35) && (long long)h * w >= 16384; }
static bool downscale_stream_ok(int sf, int k, size_t* lds_out, int* qp_out, int* ring_out) {
    const int cols = (DS_COLS - 1) * sf + k;
    int ring = 16;
    while (ring < k + 2 * DS_STEP + sf) ring <<= 1;       // rows an un-emitted output still needs + the step being written
    int qp = (cols + sf - 1) / sf + 1;
    qp |= 1;
    const size_t lds_s = ((size_t)sf * DS_STEP * qp + (size_t)ring * (DS_COLS + 1) + 2 * (size_t)k) * 4;
    const bool fits = DS_STEP * ((cols + 6) / 4 + 1) <= 256 * DS_SB && lds_s <= 64 * 1024;
    // measured: configs[4] (sf 8, k 45: 109 KB tile window, one workgroup per CU) 1.88 -> 1.05 ms; configs[1] (sf 4, k 17: 24 KB)
    // 111 -> 135 us — the streaming kernel pays ~1 us of barriers and LDS round trips per 8 rows, the tile kernel only loses where its
    // window crowds the CU
    const int rows_t = (DT - 1) * sf + k;
    const size_t lds_tile = ((size_t)sf * rows_t * ((rows_t + sf - 1) / sf + 2) + (size_t)rows_t * (DT + 1)) * 4;
    if (lds_out) *lds_out = lds_s;
    if (qp_out) *qp_out = qp;
    if (ring_out) *ring_out = ring;
    return fits && lds_tile > 64 * 1024;
}

extern "C" int esr_cem_sep_form(int op, int sf, int k, int pre, int h, int w) {
    if (sf < 1 || k < 1 || !(k & 1) || h <= 0 || w <= 0) return ESR_E_ARG;
    if (op == 0) return downscale_wave_ok(sf, k, h, w) ? 2 : (downscale_stream_ok(sf, k, nullptr, nullptr, nullptr) ? 1 : 0);
    if (op == 1) return upscale_wave_ok(sf, k, pre, h, w) ? 2 : 0;
    if (op == 2) return lrfilter_wave_ok(k, h, w) ? 2 : 0;
    return ESR_E_ARG;
}

extern "C" int esr_cem_downscale(const float* y, int B, int C, int h, int w, int sf, int pre, const float* taps, int k, const float* lr,
                                 int lr_pad, float* d, esr_stream_t stream) {
    if (!y || !taps || !d || B <= 0 || C <= 0 || h <= 0 || w <= 0 || sf < 1 || k < 1 || !(k & 1) || pre < 0 || pre >= sf) return ESR_E_ARG;
    if (lr && (h - 2 * lr_pad <= 0 || w - 2 * lr_pad <= 0 || lr_pad < 0)) return ESR_E_ARG;
    const long long total = (long long)B * C * h * w;
    ESR_CLEAR_ERR();
    {
        const int rows = (DT - 1) * sf + k, qcols = (rows + sf - 1) / sf + 1;
        int qpitch = qcols;
        while ((sf * qpitch) % 32 != 16 && qpitch < qcols + 32) ++qpitch;      // rows of 16 lanes on disjoint banks (possible for sf = 2^n <= 16)
        if ((sf * qpitch) % 32 != 16) qpitch = qcols | 1;
        const size_t lds = (size_t)sf * rows * qpitch * 4;
        if (lds <= 150 * 1024 && (long long)B * C <= 65535) {
            ESR_ALLOW_160K_LDS(cem_downscale_tiled_kernel);
            hipLaunchKernelGGL(cem_downscale_tiled_kernel, dim3((w + DT - 1) / DT, (h + DT - 1) / DT, B * C), dim3(256), lds, (hipStream_t)stream, y, h, w,
                               sf, pre, taps, k, lr, lr_pad, d, qpitch, rows);
            ESR_CHECK_LAUNCH();
            return ESR_OK;
        }
    }
    hipLaunchKernelGGL(cem_downscale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, h, w, sf, pre, taps, k,
                       lr, lr_pad, d, total);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_cem_lrfilter(const float* x, int B, int C, int h, int w, const float* taps, int k, float* out, esr_stream_t stream) {
    if (!x || !taps || !out || B <= 0 || C <= 0 || h <= 0 || w <= 0 || k < 1 || !(k & 1)) return ESR_E_ARG;
    const long long total = (long long)B * C * h * w;
    ESR_CLEAR_ERR();
    {
        int pitch = LT_TX + k - 1;
        pitch += (16 - pitch % 32 + 32) % 32;                                   // pitch % 32 == 16
        const size_t lds = (size_t)(LT_TY + k - 1) * pitch * 4;
        if (lds <= 150 * 1024 && (long long)B * C <= 65535) {
            ESR_ALLOW_160K_LDS(cem_lrfilter_tiled_kernel);
            hipLaunchKernelGGL(cem_lrfilter_tiled_kernel, dim3((w + LT_TX - 1) / LT_TX, (h + LT_TY - 1) / LT_TY, B * C), dim3(256), lds, (hipStream_t)stream,
                               x, h, w, taps, k, out, pitch);
            ESR_CHECK_LAUNCH();
            return ESR_OK;
        }
    }
    hipLaunchKernelGGL(cem_lrfilter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, h, w, taps, k, out,
                       total);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_cem_upscale(const float* f, const float* f2, int B, int C, int h, int w, int sf, int pre, const float* taps, int k,
                               const float* g, int crop, int mode, float range, float* out, float* out2, esr_stream_t stream) {
    if (!f || !taps || !out || B <= 0 || C <= 0 || h <= 0 || w <= 0 || sf < 2 || k < 1 || !(k & 1) || pre < 0 || pre >= sf) return ESR_E_ARG;
    if (mode < 0 || mode > 3 || crop < 0 || 2 * crop >= h * sf || 2 * crop >= w * sf) return ESR_E_ARG;
    if (mode >= 1 && !g) return ESR_E_ARG;
    if (mode >= 2 && !f2) return ESR_E_ARG;
    if (mode == 3 && !out2) return ESR_E_ARG;
    const long long total = (long long)B * C * (h * sf - 2 * crop) * (w * sf - 2 * crop);
    const dim3 grid((unsigned)((total + 255) / 256));
    ESR_CLEAR_ERR();
    {
        // low-resolution window of one 16 x 64 output tile: every sample within +-p of it
        const int wr = (UT_Y + 2 * (k / 2)) / sf + 3, wc = (UT_X + 2 * (k / 2)) / sf + 3;
        const size_t lds = ((size_t)k * k + (size_t)(mode >= 2 ? 2 : 1) * wr * wc + (mode >= 2 ? 0 : 0)) * 4 + (mode >= 2 ? 0 : (size_t)wr * wc * 4);
        const int Ho = h * sf - 2 * crop, Wo = w * sf - 2 * crop;
        if (lds <= 60 * 1024 && (long long)B * C <= 65535) {
            const dim3 tg((Wo + UT_X - 1) / UT_X, (Ho + UT_Y - 1) / UT_Y, B * C);
            if (mode >= 2)
                hipLaunchKernelGGL(cem_upscale_tiled_kernel<true>, tg, dim3(256), lds, (hipStream_t)stream, f, f2, h, w, sf, pre, taps, k, g, crop, mode,
                                   range, out, out2, wr, wc);
            else
                hipLaunchKernelGGL(cem_upscale_tiled_kernel<false>, tg, dim3(256), lds, (hipStream_t)stream, f, f2, h, w, sf, pre, taps, k, g, crop, mode,
                                   range, out, out2, wr, wc);
            ESR_CHECK_LAUNCH();
            return ESR_OK;
        }
    }
    if (mode >= 2)
        hipLaunchKernelGGL(cem_upscale_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, f, f2, h, w, sf, pre, taps, k, g, crop, mode, range,
                           out, out2, total);
    else
        hipLaunchKernelGGL(cem_upscale_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, f, f2, h, w, sf, pre, taps, k, g, crop, mode, range,
                           out, out2, total);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}


// ---- separable variants: taps[a][b] = tv[a] * th[b] (device arrays of k floats each).  Same arguments and results as the 2-D entry points
// above up to fp32 rounding of the tap products; ESR_E_UNSUPPORTED when the tile does not fit (the caller then uses the 2-D entry point).
extern "C" int esr_cem_downscale_sep(const float* y, int B, int C, int h, int w, int sf, int pre, const float* tv, const float* th, int k, const float* lr,
                                     int lr_pad, float* d, esr_stream_t stream) {
    if (!y || !tv || !th || !d || B <= 0 || C <= 0 || h <= 0 || w <= 0 || sf < 1 || k < 1 || !(k & 1) || pre < 0 || pre >= sf) return ESR_E_ARG;
    if (lr && (h - 2 * lr_pad <= 0 || w - 2 * lr_pad <= 0 || lr_pad < 0)) return ESR_E_ARG;
    if ((long long)B * C > 65535) return ESR_E_UNSUPPORTED;
    {
        // wave-streaming kernel (round 6): images large enough to give every CU several waves; chosen from the image geometry alone, so that a batch
        // and its chunks run the same arithmetic
        const int na = (k + sf - 1) / sf;
        if (downscale_wave_ok(sf, k, h, w)) {
            int nout = (252 - (k - 1)) / sf + 1;                // a lane's four columns start up to 3 columns left of the window
            if (nout > WV_LANES) nout = WV_LANES;
            const int nct = (w + nout - 1) / nout;
            nout = (w + nct - 1) / nct;                         // equal tiles
            typedef void (*wk_t)(const float*, int, int, int, const float*, const float*, int, const float*, int, float*, int, int, int, int, long long);
#define ESR_DW_PICK(SF_) (na == 4 ? cem_downscale_wave_kernel<SF_, 4> : na == 5 ? cem_downscale_wave_kernel<SF_, 5> : cem_downscale_wave_kernel<SF_, 6>)
            const wk_t wk = sf == 2 ? ESR_DW_PICK(2) : sf == 3 ? ESR_DW_PICK(3) : sf == 4 ? ESR_DW_PICK(4) : ESR_DW_PICK(8);
#undef ESR_DW_PICK
            // strips of 12-14 output rows, cut by the image height ALONE: odd strips walk up (their outputs sum the vertical taps in the other
            // order), so the cut must not depend on the batch
            const long long cols_total = (long long)B * C * nct;
            // (configs[1], h = 148: 10 / 12 / 14 / 16 / 20 rows 45 / 34.5 / 37 / 35 / 42 us; configs[4], h = 280: 275 / 289 / 270 / 285 / 273 us)
            const int rows = g_cem_wave_rmin > 0 ? g_cem_wave_rmin : (h < 256 ? 12 : 14);
            int nst = (h + rows - 1) / rows;
            const int R = (h + nst - 1) / nst;
            nst = (h + R - 1) / R;
            const long long nitems = cols_total * nst;
            const unsigned nwg = (unsigned)((nitems + 3) / 4);
            ESR_CLEAR_ERR();
            hipLaunchKernelGGL(wk, dim3(nwg), dim3(256), 0, (hipStream_t)stream, y, h, w, pre, tv, th, k, lr, lr_pad, d, nout, R, nct, nst, nitems);
            ESR_CHECK_LAUNCH();
            return ESR_OK;
        }
    }
    {
        // streaming kernel: strips of DS_COLS x DS_ROWS outputs, DS_STEP window rows per step
        size_t lds_s = 0;
        int qp = 0, ring = 0;
        if (downscale_stream_ok(sf, k, &lds_s, &qp, &ring)) {
            ESR_CLEAR_ERR();
            void (*ks)(const float*, int, int, int, int, const float*, const float*, int, const float*, int, float*, int, int) =
                sf == 2 ? cem_downscale_stream_kernel<2> : sf == 3 ? cem_downscale_stream_kernel<3> : sf == 4 ? cem_downscale_stream_kernel<4>
                : sf == 8 ? cem_downscale_stream_kernel<8> : cem_downscale_stream_kernel<0>;
            hipLaunchKernelGGL(ks, dim3((w + DS_COLS - 1) / DS_COLS, (h + DS_ROWS - 1) / DS_ROWS, B * C), dim3(256), lds_s, (hipStream_t)stream, y, h, w, sf, pre, tv,
                               th, k, lr, lr_pad, d, qp, ring);
            ESR_CHECK_LAUNCH();
            return ESR_OK;
        }
    }
    const int rows = (DT - 1) * sf + k, qcols = (rows + sf - 1) / sf + 1;
    // pitches of the window planes and of the horizontal pass's output: rows sf apart on disjoint halves of the 32 banks (possible for sf = 2^n
    // and 3; see the kernel's horizontal pass)
    int qpitch = qcols, hpp = DT;
    while ((sf * qpitch) % 32 != 16 && qpitch < qcols + 32) ++qpitch;
    if ((sf * qpitch) % 32 != 16) qpitch = qcols | 1;
    while ((sf * hpp) % 32 != 16 && hpp < DT + 32) ++hpp;
    if ((sf * hpp) % 32 != 16) hpp = DT + 1;
    const size_t lds = ((size_t)sf * rows * qpitch + (size_t)rows * hpp) * 4;
    if (lds > 150 * 1024 || (long long)B * C > 65535) return ESR_E_UNSUPPORTED;
    ESR_CLEAR_ERR();
    void (*kern)(const float*, int, int, int, int, const float*, const float*, int, const float*, int, float*, int, int, int) =
        sf == 2 ? cem_downscale_sep_kernel<2> : sf == 3 ? cem_downscale_sep_kernel<3> : sf == 4 ? cem_downscale_sep_kernel<4> : sf == 8 ? cem_downscale_sep_kernel<8>
                                                                                                                              : cem_downscale_sep_kernel<0>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((w + DT - 1) / DT, (h + DT - 1) / DT, B * C), dim3(256), lds, (hipStream_t)stream, y, h, w, sf, pre, tv, th, k, lr, lr_pad, d,
                       qpitch, rows, hpp);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_cem_lrfilter_sep(const float* x, int B, int C, int h, int w, const float* tv, const float* th, int k, float* out, esr_stream_t stream) {
    if (!x || !tv || !th || !out || B <= 0 || C <= 0 || h <= 0 || w <= 0 || k < 1 || !(k & 1)) return ESR_E_ARG;
    if (lrfilter_wave_ok(k, h, w) && (long long)B * C <= 65535) {
        typedef void (*fk_t)(const float*, int, int, const float*, const float*, float*, int, int, int, int, long long);
        const fk_t fk = k == 27 ? cem_lrfilter_wave_kernel<27> : cem_lrfilter_wave_kernel<35>;
        int nout = 128 - (k - 1);
        const int nct = (w + nout - 1) / nout;
        nout = ((w + nct - 1) / nct + 1) & ~1;                  // equal tiles, whole column pairs
        const long long cols_total = (long long)B * C * nct;
        int nst = (int)(wave_strips_target((const void*)fk) / cols_total);
        if (nst < 1) nst = 1;
        int R = (h + nst - 1) / nst;
        if (R < g_cem_filt_rmin) R = g_cem_filt_rmin;            // (every strip runs k - 1 rows before its first output)
        if (R > h) R = h;
        nst = (h + R - 1) / R;
        R = (h + nst - 1) / nst;
        const long long nitems = cols_total * nst;
        ESR_CLEAR_ERR();
        hipLaunchKernelGGL(fk, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, h, w, tv, th, out, nout, R, nct, nst, nitems);
        ESR_CHECK_LAUNCH();
        return ESR_OK;
    }
    int pitch = LT_TX + k - 1;
    pitch += (16 - pitch % 32 + 32) % 32;
    const size_t lds = ((size_t)(LT_TY + k - 1) * pitch + (size_t)(LT_TY + k - 1) * (LT_TX + 1)) * 4;
    if (lds > 150 * 1024 || (long long)B * C > 65535) return ESR_E_UNSUPPORTED;
    ESR_CLEAR_ERR();
    ESR_ALLOW_160K_LDS(cem_lrfilter_sep_kernel);
    hipLaunchKernelGGL(cem_lrfilter_sep_kernel, dim3((w + LT_TX - 1) / LT_TX, (h + LT_TY - 1) / LT_TY, B * C), dim3(256), lds, (hipStream_t)stream, x, h, w, tv, th,
                       k, out, pitch);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

// tvf / thf / kf: the LR filter folded in front of the upscale (esr_cem_filter_upscale_sep), or NULL / 0
static int upscale_sep_launch(const float* f, const float* f2, int B, int C, int h, int w, int sf, int pre, const float* tvf, const float* thf, int kf,
                              const float* tv, const float* th, int k, const float* g, int crop, int mode, float range, float* out, float* out2,
                              esr_stream_t stream) {
    if (!f || !tv || !th || !out || B <= 0 || C <= 0 || h <= 0 || w <= 0 || sf < 2 || k < 1 || !(k & 1) || pre < 0 || pre >= sf) return ESR_E_ARG;
    if (mode < 0 || mode > 3 || crop < 0 || 2 * crop >= h * sf || 2 * crop >= w * sf) return ESR_E_ARG;
    if ((mode >= 1 && !g) || (mode >= 2 && !f2) || (mode == 3 && !out2)) return ESR_E_ARG;
    const bool filt = kf > 0;
    if (filt && (!tvf || !thf || !(kf & 1))) return ESR_E_ARG;
    {
        // wave-streaming kernel (round 6): chosen from the image geometry alone (a batch and its chunks run the same arithmetic)
        const int na = (k + sf - 1) / sf;
        if (!filt && upscale_wave_ok(sf, k, pre, h, w)) {
            const int Ho = h * sf - 2 * crop, Wo = w * sf - 2 * crop;
            const int nct = (Wo + 255) / 256;
            const int tcols = ((Wo + nct - 1) / nct + 3) & ~3;          // equal column tiles, whole quads
            typedef void (*uk_t)(const float*, const float*, int, int, int, const float*, const float*, int, const float*, int, int, float, float*, float*, int, int, int,
                                 int, long long);
            uk_t uk;
#define ESR_UW_PICK(TWO_)                                                                                                                                   \
    (sf == 3 ? (na == 4 ? cem_upscale_wave_kernel<TWO_, 3, 4> : na == 5 ? cem_upscale_wave_kernel<TWO_, 3, 5> : cem_upscale_wave_kernel<TWO_, 3, 6>)              \
     : sf == 4 ? (na == 4 ? cem_upscale_wave_kernel<TWO_, 4, 4> : na == 5 ? cem_upscale_wave_kernel<TWO_, 4, 5> : cem_upscale_wave_kernel<TWO_, 4, 6>)            \
               : (na == 4 ? cem_upscale_wave_kernel<TWO_, 8, 4> : na == 5 ? cem_upscale_wave_kernel<TWO_, 8, 5> : cem_upscale_wave_kernel<TWO_, 8, 6>))
            uk = mode >= 2 ? ESR_UW_PICK(true) : ESR_UW_PICK(false);
#undef ESR_UW_PICK
            const long long cols_total = (long long)B * C * nct;
            int nst = (int)(wave_strips_target((const void*)uk) / cols_total);
            if (nst < 1) nst = 1;
            int S = (Ho + nst - 1) / nst;
            S = (S + sf - 1) / sf * sf;                                  // whole blocks of sf rows
            if (S < 4 * sf) S = 4 * sf;
            nst = (Ho + S - 1) / S;
            S = ((Ho + nst - 1) / nst + sf - 1) / sf * sf;              // strips of equal height
            nst = (Ho + S - 1) / S;
            const long long nitems = cols_total * nst;
            const unsigned nwg = (unsigned)((nitems + 3) / 4);
            ESR_CLEAR_ERR();
            hipLaunchKernelGGL(uk, dim3(nwg), dim3(256), 0, (hipStream_t)stream, f, f2, h, w, pre, tv, th, k, g, crop, mode, range, out, out2, S, tcols, nct, nst,
                               nitems);
            ESR_CHECK_LAUNCH();
            return ESR_OK;
        }
    }
    const int wr = (US_Y + 2 * (k / 2)) / sf + 3, wc = (US_X + 2 * (k / 2)) / sf + 3;
    const int two = mode >= 2 ? 2 : 1;
    const int vp = wc + (16 - wc % 32 + 32) % 32;                       // pass-1 row pitch = 16 (mod 32)
    const int er = wr + kf - 1, ec = wc + kf - 1;
    const int ep = ec | 1, hpp = wc | 1;                                // odd pitches: a column walk down the rows touches every bank once
    const size_t lds = ((size_t)two * wr * wc + (size_t)two * US_Y * vp + 2 * (size_t)k + (filt ? (size_t)er * ep + (size_t)er * hpp : 0)) * 4;
    if (lds > 64 * 1024 || (long long)B * C > 65535) return ESR_E_UNSUPPORTED;
    const int Ho = h * sf - 2 * crop, Wo = w * sf - 2 * crop;
    const dim3 tg((Wo + US_X - 1) / US_X, (Ho + US_Y - 1) / US_Y, B * C);
    ESR_CLEAR_ERR();
    typedef void (*up_t)(const float*, const float*, int, int, int, int, const float*, const float*, int, const float*, int, int, float, float*, float*, int, int, int,
                         const float*, const float*, int, int, int);
    up_t kern;
#define ESR_UP_PICK(TWO_, FILT_)                                                                                                                   \
    (sf == 2 ? cem_upscale_sep_kernel<TWO_, 2, FILT_> : sf == 3 ? cem_upscale_sep_kernel<TWO_, 3, FILT_> : sf == 4 ? cem_upscale_sep_kernel<TWO_, 4, FILT_> \
     : sf == 8 ? cem_upscale_sep_kernel<TWO_, 8, FILT_> : cem_upscale_sep_kernel<TWO_, 0, FILT_>)
    if (filt) kern = mode >= 2 ? ESR_UP_PICK(true, true) : ESR_UP_PICK(false, true);
    else kern = mode >= 2 ? ESR_UP_PICK(true, false) : ESR_UP_PICK(false, false);
#undef ESR_UP_PICK
    hipLaunchKernelGGL(kern, tg, dim3(256), lds, (hipStream_t)stream, f, f2, h, w, sf, pre, tv, th, k, g, crop, mode, range, out, out2, wr, wc, vp, tvf, thf, kf,
                       ep, hpp);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_cem_upscale_sep(const float* f, const float* f2, int B, int C, int h, int w, int sf, int pre, const float* tv, const float* th, int k,
                                   const float* g, int crop, int mode, float range, float* out, float* out2, esr_stream_t stream) {
    return upscale_sep_launch(f, f2, B, C, h, w, sf, pre, nullptr, nullptr, 0, tv, th, k, g, crop, mode, range, out, out2, stream);
}

extern "C" int esr_cem_filter_upscale_sep(const float* e, const float* e2, int B, int C, int h, int w, int sf, int pre, const float* tvf, const float* thf,
                                          int kf, const float* tv, const float* th, int k, const float* g, int crop, int mode, float range, float* out,
                                          float* out2, esr_stream_t stream) {
    if (kf < 1) return ESR_E_ARG;
    return upscale_sep_launch(e, e2, B, C, h, w, sf, pre, tvf, thf, kf, tv, th, k, g, crop, mode, range, out, out2, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Adjoint (transpose) of the CEM filters.  All three forward ops have the form
//     y[q] = sum_{a,b} taps[a][b] * x_frame[ clamp(m(q_y) + a - p) ][ clamp(m(q_x) + b - p) ],   m(q) = q*sq + oq
// on a "frame" of N_y x N_x pixels (replicate padding = index clamping), of which only positions n*sn + on are real unknowns
// (DownscaleOP: frame = HR, sq = sf, oq = pre;  Conv_LR_with_Inv_hTh_OP: frame = LR, sq = 1;  Upscale_OP: frame = HR, sq = 1,
// unknowns at sn = sf, on = pre).  The adjoint is the gather
//     dx[n] = sum_q dy[q] * T[ry][rx][ iy(q_y) ][ ix(q_x) ],      n' = n*sn + on
// where for an interior frame position the tap index is n' - m(q) + p, and for the first / last frame row (column) every tap
// that the clamp folded onto it counts: T holds prefix / suffix sums of the taps along that axis (9 tables [3][3][k][k],
// built on the host: 0 = prefix (first row), 1 = plain (interior), 2 = suffix (last row)).
__global__ void cem_adjoint_kernel(const float* __restrict__ dy, int hq, int wq, int sq, int oq, int Ny, int Nx, const float* __restrict__ tabs, int k,
                                   int hn, int wn, int sn, int on, float* __restrict__ dx, int accumulate, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int nx = (int)(idx % wn);
    long long t = idx / wn;
    const int ny = (int)(t % hn);
    const long long bc = t / hn;
    const int p = k / 2;
    const int fy = ny * sn + on, fx = nx * sn + on;           // frame position of this unknown
    const int ry = fy == 0 ? 0 : (fy == Ny - 1 ? 2 : 1), rx = fx == 0 ? 0 : (fx == Nx - 1 ? 2 : 1);
    const float* T = tabs + (size_t)(ry * 3 + rx) * k * k;
    const float* src = dy + bc * hq * (long long)wq;
    // q ranges whose window touches this frame position (interior), or everything that can be clamped onto it (edges)
    auto range = [&](int f, int r, int N, int nq, int& q0, int& q1) {
        int lo = f + p - (k - 1) - oq, hi = f + p - oq;     // m(q) in [f+p-(k-1), f+p]
        if (r == 0) lo = -(1 << 28);                         // first row: also every q with m(q) - p < 0 ... m(q) <= p
        if (r == 2) hi = (1 << 28);
        q0 = lo <= 0 ? 0 : (lo + sq - 1) / sq;
        q1 = hi < 0 ? -1 : hi / sq;
        if (q1 > nq - 1) q1 = nq - 1;
        (void)N;
    };
    auto tap_index = [&](int f, int r, int N, int q) -> int {
        const int m = q * sq + oq;
        if (r == 1) return f - m + p;
        if (r == 0) { const int a = p - m; return a > k - 1 ? k - 1 : a; }        // prefix table: all taps a' <= p - m
        const int a = N - 1 - m + p; return a < 0 ? 0 : a;                         // suffix table: all taps a' >= N-1-m+p
    };
    int qy0, qy1, qx0, qx1;
    range(fy, ry, Ny, hq, qy0, qy1);
    range(fx, rx, Nx, wq, qx0, qx1);
    float acc = 0.f;
    for (int qy = qy0; qy <= qy1; ++qy) {
        const int a = tap_index(fy, ry, Ny, qy);
        if (a < 0 || a >= k) continue;
        const float* row = src + (long long)qy * wq;
        const float* tr = T + a * k;
        for (int qx = qx0; qx <= qx1; ++qx) {
            const int b = tap_index(fx, rx, Nx, qx);
            if (b < 0 || b >= k) continue;
            acc = fmaf(tr[b], row[qx], acc);
        }
    }
    dx[idx] = accumulate ? dx[idx] + acc : acc;
}

// The same adjoint for rank-one taps (taps = tv (x) th: the bicubic kernels and their inv_hTh — every filter of the shipped configs): the 2-D tables
// factor into per-axis tables T[ry][rx][a][b] = Tv[ry][a] * Th[rx][b] ([3][k] each: prefix / plain / suffix), so the gather is two 1-D passes,
// horizontal (dy [hq][wq] -> tmp [hq][wn]) then vertical (tmp -> dx [hn][wn]): k_y + k_x products per unknown instead of k_y * k_x (the 27 x 27
// inv_hTh filter: 54 instead of 729), every read of the vertical pass coalesced along x.  One thread per output; AXIS 0: along x, 1: along y.
// out = (base ? base : 0) + alpha * sum  (the projection's backward folds  dfull - D^T(de)  into the last pass).
template <int AXIS>
__global__ void cem_adjoint_1d_kernel(const float* __restrict__ src, int hs, int ws, int sq, int oq, int N, const float* __restrict__ tab, int k,
                                      int n_out, int sn, int on, const float* __restrict__ base, float alpha, float* __restrict__ dst, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    // AXIS 0: dst [bc][hs][n_out], the sum runs along a source row;  AXIS 1: dst [bc][n_out][ws], the sum runs down a source column
    const int wo = AXIS == 0 ? n_out : ws, ho = AXIS == 0 ? hs : n_out;
    const int x = (int)(idx % wo);
    long long t = idx / wo;
    const int y = (int)(t % ho);
    const long long bc = t / ho;
    const int n = AXIS == 0 ? x : y, nq = AXIS == 0 ? ws : hs;
    const int p = k / 2;
    const int f = n * sn + on;                                   // frame position of this unknown
    const int r = f == 0 ? 0 : (f == N - 1 ? 2 : 1);
    const float* T = tab + r * k;
    int lo = f + p - (k - 1) - oq, hi = f + p - oq;              // m(q) = q*sq + oq in [f+p-(k-1), f+p]; first / last frame position: everything clamped onto it
    if (r == 0) lo = -(1 << 28);
    if (r == 2) hi = (1 << 28);
    int q0 = lo <= 0 ? 0 : (lo + sq - 1) / sq;
    int q1 = hi < 0 ? -1 : hi / sq;
    if (q1 > nq - 1) q1 = nq - 1;
    const float* sp = src + bc * hs * (long long)ws + (AXIS == 0 ? (long long)y * ws : (long long)x);
    const long long stride = AXIS == 0 ? 1 : ws;
    float acc = 0.f;
    for (int q = q0; q <= q1; ++q) {
        const int m = q * sq + oq;
        int a;
        if (r == 1) a = f - m + p;
        else if (r == 0) { a = p - m; a = a > k - 1 ? k - 1 : a; }
        else { a = N - 1 - m + p; a = a < 0 ? 0 : a; }
        if (a < 0 || a >= k) continue;
        acc = fmaf(T[a], sp[q * stride], acc);
    }
    dst[idx] = (base ? base[idx] : 0.f) + alpha * acc;
}

// tabs_v / tabs_h: [3][k] prefix / plain / suffix sums of the vertical / horizontal factor; tmp: B*C*hq*wn floats of scratch
extern "C" int esr_cem_adjoint_sep(const float* dy, int B, int C, int hq, int wq, int sq, int oq, int Ny, int Nx, const float* tabs_v, const float* tabs_h,
                                   int k, int hn, int wn, int sn, int on, float* tmp, const float* base, float alpha, float* dx, esr_stream_t stream) {
    if (!dy || !tabs_v || !tabs_h || !tmp || !dx || B <= 0 || C <= 0 || hq <= 0 || wq <= 0 || hn <= 0 || wn <= 0 || sq < 1 || sn < 1 || k < 1 || !(k & 1)) return ESR_E_ARG;
    if ((hn - 1) * sn + on >= Ny || (wn - 1) * sn + on >= Nx || (hq - 1) * sq + oq >= Ny || (wq - 1) * sq + oq >= Nx) return ESR_E_ARG;
    const long long t1 = (long long)B * C * hq * wn, t2 = (long long)B * C * hn * wn;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(cem_adjoint_1d_kernel<0>, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, hq, wq, sq, oq, Nx, tabs_h, k, wn, sn, on,
                       (const float*)nullptr, 1.f, tmp, t1);
    hipLaunchKernelGGL(cem_adjoint_1d_kernel<1>, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)tmp, hq, wn, sq, oq, Ny, tabs_v, k,
                       hn, sn, on, base, alpha, dx, t2);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_cem_adjoint(const float* dy, int B, int C, int hq, int wq, int sq, int oq, int Ny, int Nx, const float* tabs, int k,
                               int hn, int wn, int sn, int on, float* dx, int accumulate, esr_stream_t stream) {
    if (!dy || !tabs || !dx || B <= 0 || C <= 0 || hq <= 0 || wq <= 0 || hn <= 0 || wn <= 0 || sq < 1 || sn < 1 || k < 1 || !(k & 1)) return ESR_E_ARG;
    if ((hn - 1) * sn + on >= Ny || (wn - 1) * sn + on >= Nx || (hq - 1) * sq + oq >= Ny || (wq - 1) * sq + oq >= Nx) return ESR_E_ARG;
    const long long total = (long long)B * C * hn * wn;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(cem_adjoint_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, hq, wq, sq, oq, Ny, Nx, tabs, k,
                       hn, wn, sn, on, dx, accumulate, total);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

#ifdef ESR_TRACE
extern "C" void esr_debug_cem_wave(int on) { g_cem_wave = (on & 1) != 0; if (on >> 8) g_cem_wave_target = on >> 8; if ((on >> 1) & 0x7F) { g_cem_wave_rmin = (on >> 1) & 0x7F; g_cem_filt_rmin = (on >> 1) & 0x7F; } }
#endif
