// Weight / bias gradients of a whole residual dense block in one work decomposition (round 3).
//
// The five convs of a ResidualDenseBlock_5C (codes/models/modules/block.py:230-235) read one growing buffer X = [x (64) | c1 | c2 | c3 | c4]
// and their output gradients sit back to back in the block's gradient buffer G' = [dy conv4 (64) | dy conv3 (32) | dy conv2 | dy conv1 | dy conv0]
// (esr_hip/engine.py, packed_rdb_t).  A 32-channel tile `it` of X is an operand of EVERY conv c >= max(0, it - 1) — i.e. of the first
// n_out = 6, 6, 5, 4, 3, 2 (it = 0..5) 32-channel tiles of G'.  conv3x3_wgrad_batch_kernel gives every (input tile, output tile) pair its
// own workgroup, which copies both tiles into LDS for 9 taps of MFMAs: 38 KB per 36 MFMAs per wave in 'split' — more than the CU can
// ingest (27 B/cycle) at the rate the matrix pipe consumes it.  Here a workgroup owns ONE input tile against ALL the output tiles that pair
// with it: per 2 x 32-pixel tile it copies the haloed X tile once and n_out dY tiles, and its four waves split the (tap, output tile)
// pairs — output tiles in two halves x taps {0..4} / {5..8} (+ the bias as a tenth "tap": dY x ones) — so that every wave runs all pixels
// (no cross-wave reduction) with <= 3 x 5 accumulator tiles (240 registers).  LDS ingest per MFMA: 96 B instead of 267 B ('split').
// Operands as in esr_bwd.hip: K = pixels, both operands pixel-major with 8 channels per 16-byte vector, transposed on the fly by
// ds_read_b64_tr_b16; split terms dYlo*Xhi + dYhi*Xlo + dYhi*Xhi.
#include "esr_common.h"
#include <vector>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int TH = 2, TW = 32;                         // pixel tile: 2 rows x 32 columns = 4 K steps of 16 pixels
constexpr int XPIX = (TH + 2) * (TW + 2);              // 136 haloed x pixels per group plane
constexpr int XP = 148, YP = TH * TW, YPP = 68;        // padded plane sizes: == 4 (mod 16) — bank spreading of the transposing reads
static_assert(XP % 16 == 4 && YPP % 16 == 4 && XP >= XPIX, "plane padding");
constexpr int XSLOTS = (XPIX + 63) / 64;               // 3 copy slots (64 pixel vectors) per x group plane; 1 per dY group plane

struct RdbConv { float* dw; float* db; int cin_total; float alpha; };
struct RdbArgs {
    DView x, z, g;                 // the block's activation buffer (24 groups), the latent group (optional), the gradient buffer G' (24 groups)
    int lat, B, H, W;
    RdbConv conv[5];
};

__device__ __forceinline__ void glds16w(const uint4* src, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
__device__ __forceinline__ uint4 frag_tr(const unsigned char* p) {
    typedef __attribute__((address_space(3))) s16x4* lptr;
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(size_t)(p));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(size_t)(p + 64));
    const uint2 u0 = __builtin_bit_cast(uint2, r0), u1 = __builtin_bit_cast(uint2, r1);
    return make_uint4(u0.x, u0.y, u1.x, u1.y);
}
template <int FMT>
__device__ __forceinline__ f32x16_t mfma_e(uint4 a, uint4 b, f32x16_t c) {
    if constexpr (FMT == ESR_FMT_F16) {
        typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
}

// G' tile j -> (conv, 32-row tile inside the conv's output)
__device__ __forceinline__ void tile_conv(int j, int& c, int& cot) {
    if (j < 2) { c = 4; cot = j; } else { c = 5 - j; cot = 0; }
}

// One workgroup: input tile `it` (0..5: main tiles of X; 6: the latent group) of block `a`, pixel tiles [slice, ntiles) step nslices.
// NJ = output tiles per wave half = ceil(n_out / 2).
template <int NPL, int FMT, int NJ>
__device__ __forceinline__ void rdb_body(const RdbArgs& a, const RdbConv* __restrict__ convs, const int it, const int n_out, const int slice, const int nslices,
                                         unsigned char* const smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jh = wave >> 1, th = wave & 1;                     // output-tile half, tap half
    const bool lat_tile = it == 6;
    const bool do_bias = it == 0;                                // the workgroups of input tile 0 hold every dY tile: they also reduce dY itself
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int ngy = 4 * n_out;                                   // dY group planes staged
    const int plane_vecs = 4 * XP + ngy * YPP;                   // one element plane of a stage: [x: 4 groups | dy: 4 n_out groups]
    const int stage_bytes = NPL * plane_vecs * 16;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * a.B;
    const DView& xv = lat_tile ? a.z : a.x;
    const int Wp = a.W + 2;

    f32x16_t acc[NJ][5];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[jj][k][i] = 0.f;
    constexpr uint32_t ONE2 = FMT == ESR_FMT_F16 ? 0x3C003C00u : 0x3F803F80u;
    const uint4 ones = make_uint4(ONE2, ONE2, ONE2, ONE2);

    // this lane's source address inside a fragment's 16-lane group (esr_bwd.hip, frag_tr)
    const int li = lane & 15, grp16 = lane >> 4;
    const int rb2 = (grp16 & 1) * 2 + ((li & 3) >> 1);
    const int kb = (grp16 >> 1) * 8 + (li >> 2);
    const int frag_off = kb * 16 + (li & 1) * 8;
    // this wave's taps: slot k -> tap th*5 + k; the second half has four real taps, its fifth slot is the bias (dY x ones) or idle
    int xoff[5];
    bool tap_ok[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int t = th * 5 + k;
        tap_ok[k] = t < 9;
        const int tt = t < 9 ? t : 8;
        xoff[k] = (rb2 * XP + (tt / 3) * (TW + 2) + tt % 3) * 16 + frag_off;
    }
    const bool bias_slot = th == 1 && do_bias;
    // this wave's output tiles: jj -> G' tile jh*NJ + jj (clamped: a padded slot recomputes the last tile and is not written)
    int yoff[NJ];
    bool j_ok[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = jh * NJ + jj;
        j_ok[jj] = j < n_out;
        yoff[jj] = (4 * XP + (4 * (j < n_out ? j : n_out - 1) + rb2) * YPP) * 16 + frag_off;
    }

    // ---- copies of one pixel tile: 12 x slots + 4 n_out dY slots per element plane, dealt round-robin to the four waves
    const int nslot = 4 * XSLOTS + ngy;
    auto issue = [&](const int tile, const unsigned st) {
        const int tx = tile % tiles_x;
        const int r1 = tile / tiles_x;
        const int ty = r1 % tiles_y;
        const int b = r1 / tiles_y;
        const int x0 = tx * TW, y0 = ty * TH;
        for (int s = wave; s < nslot; s += 4) {                  // wave-uniform
            if (s < 4 * XSLOTS) {
                const int grp = s / XSLOTS, sl = s - grp * XSLOTS;
                const int xcg = lat_tile ? grp : it * 4 + grp;
                const bool have = xcg < xv.ncg;
                const int p = sl * 64 + lane;
                const int rr = p / (TW + 2), cc = p - rr * (TW + 2);
                const int Yp = y0 + rr, Xp = x0 + cc;            // padded coordinates of the haloed tile
                const int off = (have && Yp < a.H + 2 && Xp < a.W + 2) ? Yp * Wp + Xp : 0;          // 0: the zero border vector
                if (p < XPIX) {
                    const unsigned dst = st + (unsigned)(grp * XP + sl * 64) * 16;      // (+ lane * 16 by the hardware)
                    glds16w(xv.hi + b * xv.bs + (have ? xcg : 0) * xv.cs + off, dst);
                    if (NPL == 2) glds16w(xv.lo + b * xv.bs + (have ? xcg : 0) * xv.cs + off, dst + plane_vecs * 16);
                }
            } else {
                const int ycg = s - 4 * XSLOTS;                  // group of G'
                const int Y = y0 + (lane >> 5), X = x0 + (lane & 31);
                const int off = (Y < a.H && X < a.W) ? (Y + 1) * Wp + (X + 1) : 0;
                const unsigned dst = st + (unsigned)(4 * XP + ycg * YPP) * 16;
                glds16w(a.g.hi + b * a.g.bs + ycg * a.g.cs + off, dst);
                if (NPL == 2) glds16w(a.g.lo + b * a.g.bs + ycg * a.g.cs + off, dst + plane_vecs * 16);
            }
        }
    };
    // copies per wave per tile (for the counted wait): the x slots past XPIX's last lanes still issue (exec-masked), so the count is exact
    const int my_slots = (nslot - wave + 3) / 4;
    const int my_copies = NPL * my_slots;

    int tile = slice;
    int cur = 0;
    if (tile < ntiles) issue(tile, lds0);
    for (; tile < ntiles; tile += nslices) {
        const int nxt = tile + nslices;
        if (nxt < ntiles) {
            issue(nxt, lds0 + (cur ^ 1) * stage_bytes);
            // everything but the copies just issued (s_waitcnt takes an immediate: the few possible counts are enumerated)
            switch (my_copies) {
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
                case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const unsigned char* const sp = smem + cur * stage_bytes;
#pragma unroll
        for (int rr = 0; rr < TH; ++rr) {
#pragma unroll
            for (int ks = 0; ks < TW / 16; ++ks) {
                const int ypix = (rr * TW + ks * 16) * 16, xpix = (rr * (TW + 2) + ks * 16) * 16;
                uint4 fa[NJ][NPL], fb[5][NPL];
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) fa[jj][pl] = frag_tr(sp + pl * plane_vecs * 16 + yoff[jj] + ypix);
#pragma unroll
                for (int k = 0; k < 5; ++k)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        fb[k][pl] = frag_tr(sp + pl * plane_vecs * 16 + xoff[k] + xpix);
                        if (k == 4 && bias_slot) fb[k][pl] = pl == 0 ? ones : make_uint4(0, 0, 0, 0);      // uniform select
                    }
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        if (NPL == 2) {
                            acc[jj][k] = mfma_e<FMT>(fa[jj][1], fb[k][0], acc[jj][k]);
                            acc[jj][k] = mfma_e<FMT>(fa[jj][0], fb[k][NPL - 1], acc[jj][k]);
                        }
                        acc[jj][k] = mfma_e<FMT>(fa[jj][0], fb[k][0], acc[jj][k]);
                    }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    // ---- every wave owns its (output tile, tap) blocks for all pixels of its slice: straight into dW / db (atomics when the pixel sum is sliced)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        if (!j_ok[jj]) continue;
        int c, cot;
        tile_conv(jh * NJ + jj, c, cot);
        const RdbConv cv = convs[c];             // (from the table in global memory: a run-time index into the by-value copy would live in scratch)
        const int col = lane & 31;
        int ci = -1;
        if (lat_tile) { if (col < a.lat) ci = col; }
        else ci = a.lat + it * 32 + col;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const bool is_bias = k == 4 && bias_slot;
            if (!tap_ok[k] && !is_bias) continue;
            const int t = th * 5 + k;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = cot * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                const float v = cv.alpha * acc[jj][k][i];
                if (is_bias) {
                    if (col == 0 && cv.db) { if (nslices > 1) atomicAdd(cv.db + co, v); else cv.db[co] += v; }
                } else if (ci >= 0) {
                    float* const p = cv.dw + ((long long)co * cv.cin_total + ci) * 9 + t;
                    if (nslices > 1) atomicAdd(p, v); else *p += v;
                }
            }
        }
    }
}

template <int NPL, int FMT>
__global__ __launch_bounds__(256, 1) void wgrad_rdb_kernel(const RdbArgs* __restrict__ table, const int4* __restrict__ map, int nslices) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int4 m = map[blockIdx.x];
    const int e = __builtin_amdgcn_readfirstlane(m.x), it = __builtin_amdgcn_readfirstlane(m.y), slice = __builtin_amdgcn_readfirstlane(m.z);
    const RdbArgs a = table[e];
    const int n_out = it == 6 ? 6 : (it < 2 ? 6 : 7 - it);
    const RdbConv* const convs = table[e].conv;
    if (n_out >= 5) rdb_body<NPL, FMT, 3>(a, convs, it, n_out, slice, nslices, smem);
    else if (n_out >= 3) rdb_body<NPL, FMT, 2>(a, convs, it, n_out, slice, nslices, smem);
    else rdb_body<NPL, FMT, 1>(a, convs, it, n_out, slice, nslices, smem);
}

__global__ void rdb_rebase_kernel(RdbArgs* table, int n, long long delta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < 5; ++c) {
        table[i].conv[c].dw = (float*)((char*)table[i].conv[c].dw + delta);
        if (table[i].conv[c].db) table[i].conv[c].db = (float*)((char*)table[i].conv[c].db + delta);
    }
}

int64_t table_bytes(int n) { return ((int64_t)n * sizeof(RdbArgs) + 255) / 256 * 256; }

}  // namespace

extern "C" int64_t esr_wgrad_rdb_workspace_bytes(int n) {
    if (n <= 0) return ESR_E_ARG;
    return table_bytes(n) + (int64_t)n * 7 * 2 * sizeof(int4) + 256;
}

extern "C" int esr_wgrad_rdb_upload(const esr_wgrad_rdb_desc* descs, int n, void* workspace, int64_t workspace_bytes, esr_wgrad_rdb_plan* plan, esr_stream_t stream) {
    if (!descs || n <= 0 || !workspace || !plan || workspace_bytes < esr_wgrad_rdb_workspace_bytes(n)) return ESR_E_ARG;
    const bool split = descs[0].x.lo != nullptr;
    const int fmt = descs[0].x.fmt;
    std::vector<RdbArgs> table(n);
    std::vector<int4> map;
    // pixel slices: two per (block, input tile) when that is what it takes to give every CU several workgroups (atomics with two addends
    // into zero-initialised targets are order-independent)
    const int nslices = n * 7 < 3 * 256 ? 2 : 1;
    for (int i = 0; i < n; ++i) {
        const esr_wgrad_rdb_desc& d = descs[i];
        if (!d.x.hi || !d.g.hi || d.x.ncg < 24 || d.g.ncg < 24 || d.B <= 0 || d.H <= 0 || d.W <= 0 || d.lat < 0 || d.lat > 8) return ESR_E_ARG;
        if ((d.x.lo != nullptr) != split || (d.g.lo != nullptr) != split || d.x.fmt != fmt || d.g.fmt != fmt) return ESR_E_ARG;
        if (d.x.H != d.H || d.x.W != d.W || d.g.H != d.H || d.g.W != d.W) return ESR_E_ARG;
        if (d.lat > 0 && (!d.z.hi || d.z.H != d.H || d.z.W != d.W || (d.z.lo != nullptr) != split)) return ESR_E_ARG;
        if (fmt == ESR_FMT_F16 && split) return ESR_E_UNSUPPORTED;
        RdbArgs& a = table[i];
        a.x = to_dview(d.x); a.g = to_dview(d.g); a.z = to_dview(d.z);
        a.lat = d.lat; a.B = d.B; a.H = d.H; a.W = d.W;
        for (int c = 0; c < 5; ++c) {
            if (!d.dw[c]) return ESR_E_ARG;
            a.conv[c] = RdbConv{d.dw[c], d.db[c], d.lat + 64 + 32 * c, d.alpha[c]};
        }
    }
    // largest workgroups first (6 output tiles: input tiles 0, 1 and the latent tile), so that the tail of the launch is made of small ones
    const int order[7] = {0, 1, 6, 2, 3, 4, 5};
    for (int oi = 0; oi < 7; ++oi)
        for (int i = 0; i < n; ++i) {
            if (order[oi] == 6 && descs[i].lat == 0) continue;
            for (int s = 0; s < nslices; ++s) map.push_back(int4{i, order[oi], s, 0});
        }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(workspace, table.data(), (size_t)n * sizeof(RdbArgs), hipMemcpyHostToDevice, s) != hipSuccess) return ESR_E_LAUNCH;
    if (hipMemcpyAsync((char*)workspace + table_bytes(n), map.data(), map.size() * sizeof(int4), hipMemcpyHostToDevice, s) != hipSuccess) return ESR_E_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return ESR_E_LAUNCH;          // the staging vectors go out of scope
    plan->nwg = (int64_t)map.size();
    plan->n = n;
    plan->nslices = nslices;
    plan->split = split ? 1 : 0;
    plan->f16 = fmt == ESR_FMT_F16 ? 1 : 0;
    plan->reserved = 0;
    return ESR_OK;
}

extern "C" int esr_wgrad_rdb_run(const void* workspace, const esr_wgrad_rdb_plan* plan, esr_stream_t stream) {
    if (!workspace || !plan || plan->n <= 0 || plan->nwg <= 0) return ESR_E_ARG;
    const int npl = plan->split ? 2 : 1;
    void (*k)(const RdbArgs*, const int4*, int) = plan->f16 ? wgrad_rdb_kernel<1, 1> : (plan->split ? wgrad_rdb_kernel<2, 0> : wgrad_rdb_kernel<1, 0>);
    const size_t lds = (size_t)2 * npl * (4 * XP + 24 * YPP) * 16;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(k, dim3((unsigned)plan->nwg), dim3(256), lds, (hipStream_t)stream, (const RdbArgs*)workspace,
                       (const int4*)((const char*)workspace + table_bytes(plan->n)), plan->nslices);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_wgrad_rdb_rebase(void* workspace, const esr_wgrad_rdb_plan* plan, int64_t delta_bytes, esr_stream_t stream) {
    if (!workspace || !plan || plan->n <= 0) return ESR_E_ARG;
    if (delta_bytes == 0) return ESR_OK;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(rdb_rebase_kernel, dim3((plan->n + 63) / 64), dim3(64), 0, (hipStream_t)stream, (RdbArgs*)workspace, plan->n, (long long)delta_bytes);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}
