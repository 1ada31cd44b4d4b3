// Device-side building blocks of the conv3x3 kernels (csrc/esr_conv.hip: conv3x3_tile_kernel; csrc/esr_chain.hip: conv3x3_chain_kernel):
// kernel arguments, LDS-DMA copies of a chunk, the MFMAs of a chunk, the epilogue.  Design notes: the head of esr_conv.hip, DESIGN.md section 3.1.
// Everything lives in an anonymous namespace: each translation unit that includes this gets its own copy.
#pragma once
#include "esr_common.h"
#include <type_traits>
#include <vector>

namespace {

constexpr int NW = 4;          // waves per workgroup
constexpr int NTHREADS = 64 * NW;
constexpr int MAXS_BASE = 3;   // activation DMA slots (64 pixel vectors) per wave per plane: NPIX_L <= MAXS*NW*64
// Resident workgroups per CU the single-stage kernels are built (registers) and tiled (LDS) for, per M-tile count.  Measured on MI355X
// (RRDB-23 forward, ms): MT1/MT2 3/2: 76.7, 2/2: 72.9, 1/2: 74.6, 2/1: 85.4, 1/1: 85.5.  The chip is power-limited under this kernel
// (DESIGN.md): beyond the overlap that reaches the power cap, more resident waves cost clock.
constexpr int WGS_MT1 = 2, WGS_MT2 = 2;
// 32-pixel column tiles per wave (R) and activation DMA slots per wave per plane (MAXS), per M-tile count: a workgroup tile holds up to
// NW*R*32 flattened pixels.  (R = 6 for the 32-channel kernels — half the weight copies per pixel, less halo — was measured in rounds 3 and 5:
// no gain at configs[1] / configs[4], -10 % at configs[2]; profiles/r05_r6_tiles_ab.log.  Experiment variants of this file are patches or
// sed-edited scratch copies built to a side library, never switches in here.)
constexpr int r_of(int mt) { return 3; }
constexpr int maxs_of(int mt) { return r_of(mt) > 3 ? MAXS_BASE + 1 : MAXS_BASE; }

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// epilogue feature bits (template parameter EPI)
constexpr int EPI_RES1 = 1, EPI_RES2 = 2, EPI_MASK = 4, EPI_NCHW = 8, EPI_OUT2 = 16;
// residual 1 is a channel-group slice of the conv's own input (RDB conv5: out = 0.2*conv + x, block.py:235): it is added to the
// accumulators from the LDS copy the K loop stages anyway, so the epilogue has no residual loads at all
constexpr int EPI_RESIN = 32;
// pixel-shuffle store (esr_conv3x3_desc.pixel_shuffle): its own instantiations, so that the plain store carries none of its index arithmetic
constexpr int EPI_PS = 64;

struct ConvArgs {
    // ---- what the prologue decodes before it can issue the first copy (one kernarg batch): the tile space, the tile geometry and the
    // divisions by launch constants turned into multiplications on the host (esr_conv3x3: magic numbers, the waves' copy shares)
    int tiles_x, tiles_y, ntiles;
    int xcd_q, xcd_r;               // ntiles = 8 * xcd_q + xcd_r: XCD x sweeps xcd_q (+1 if x < xcd_r) consecutive tiles
    unsigned m_tx, m_ty;            // ceil(2^32 / tiles_x), ceil(2^32 / tiles_y); 0 when the divisor is 1 ...
    unsigned i_tx, i_ty;            // ... and then these are 1: n / d = umulhi(n, m) + n * i, no branch
    int TH, TW, P, NPIX_T, NPIX_L, nslots;      // nslots = 1-KiB copy slots (64 pixel vectors) per plane, the last one partial
    unsigned m_P, m_ups;            // ceil(2^20 / P), ceil(2^16 / ups)
    int H, W, Win_p, ups;           // output interior; padded input row pitch (W_in + 2); input upsample factor
    unsigned share[NW];             // per wave: activation slots | first weight fragment << 8 | weight fragments << 16 (dma_share)
    int ncp, lo_chunks, reverse, B; // chunks [0, lo_chunks) carry a lo activation plane, later ones are hi-only (PARTLO kernels); reverse: walk the tile space backwards (cache-reuse hint)
    DView in0, in1;
    const uint4* wpack;
    // where chunk cp's / tap t's fragments sit in the pack: normally 9 * MT * NPW and MT * NPW fragments apart.  A 64-channel layer of a SMALL
    // launch is run as two 32-channel slices by the MT = 1 kernel out of the same [chunk][tap][M tile][plane] pack: slice s starts s * NPW
    // fragments in and its taps are 2 * NPW fragments apart
    long long wchunk;               // 16-byte vectors between the fragments of consecutive chunks
    long long wslice;               // 16-byte vectors between the weight fragments of consecutive output slices (0: no slices)
    int wtap;                       // fragments between consecutive taps
    int nslices;                    // output slices of this launch (blockIdx.y): cout / 64 when cout > 64; 2 for a 64-channel layer run as two 32-channel halves
    // ---- epilogue
    const float* zero_bias;
    int bias_stride;                // 1, or 0 when `bias` is the zero block (output slices step through a real bias only)
    const float* bias;              // never NULL in the kernel: a launch without a bias points at the library's zero block (zero_bias())
    int cout, ncg_out;              // output channels of one slice, and their groups
    float act_slope, alpha, beta1, beta2;
    DView res1, res2, out, out2, mask;
    float* out_nchw;
    int mask_cg0, mask_cg1;
    float mask_slope;
    int resin_g0;                   // EPI_RESIN: index (in the concatenated in0|in1 group order) of the residual's first group
    float resin_scale;              // beta1 / alpha
    int ps, ps_rg0;                 // pixel-shuffle store: factor r (0 = plain) and the first row group of this launch (esr_hip.h)
    // split K (esr_conv3x3_desc.k_split_ws): blockIdx.z = which run of `ncp` chunks (kz_groups channel groups) of the input this workgroup
    // contracts; its fp32 partial sums go to slab z of the workspace ([B][nchw_ctot][H][W] each, EPI_NCHW store), bias in slab 0 only
    int ksplit, kz_groups;
    long long kz_slab;              // floats between two slabs
    int nchw_ctot;                  // channels of the fp32 NCHW destination (== cout unless the launch covers output slices)
    int stages_hint;                // esr_conv3x3_desc.lds_stages
    unsigned* range_flag;           // esr_conv3x3_desc.range_flag / range_tag (fp16 formats)
    unsigned range_tag;
#ifdef ESR_TRACE
    unsigned long long* trace;   // debug build only: per-workgroup phase timestamps (128 slots each)
#endif
};

// Which instantiation of conv3x3_tile_kernel a launch selected (launch_nst's template arguments), and a chain of up to four launches run as ONE
// (conv3x3_chain_kernel, csrc/esr_chain.hip): the layers' argument blocks as esr_conv3x3 builds them for the separate launches.
struct ConvVariant { int npl, mt, epi, nst, fmt, npw, partlo, tmode, ntile; };
constexpr int CHAIN_MAX = 4;
struct ChainArgs {
    int n, reserved[3];
    ConvArgs l[CHAIN_MAX];
};

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <int FMT>
__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
    if (FMT) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// two floats -> packed 16-bit pair of format FMT (round to nearest even); low half = first argument
template <int FMT>
__device__ __forceinline__ uint32_t cvt_pk(float lo, float hi);

// two floats -> packed bf16x2 (round to nearest even); low half = first argument
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
template <> __device__ __forceinline__ uint32_t cvt_pk<0>(float lo, float hi) { return cvt_pk_bf16(lo, hi); }
template <> __device__ __forceinline__ uint32_t cvt_pk<1>(float lo, float hi) { return f2h(lo) | (f2h(hi) << 16); }

// Asynchronous global -> LDS copy, 16 bytes per lane: LDS destination = (wave-uniform) lds_dst + lane*16; the source is a uniform base
// (SGPR pair) + a per-lane 32-bit byte offset: no 64-bit per-lane address arithmetic per copy (the offsets of a tile's slots are computed once
// per tile, the bases once per chunk).
// Issued through inline asm on purpose: hipcc treats the builtin form as a pending LDS write and drains vmcnt(0) in front of
// every later ds_read, which would serialise the copy of step s+1 with the MFMAs of step s.  Hidden from the compiler, the
// copy is ordered by hand: wait_vm_upto() + barrier before the first read of a stage (see the step loop).
// M0 (the DMA's LDS base) is not preserved by hipcc across statements and no other instruction of this kernel reads it.
__device__ __forceinline__ void glds16s(const uint4* sbase, unsigned voff, unsigned lds_dst) {
    // (readfirstlane: the destination is wave-uniform by construction, but the compiler cannot always prove it and M0 takes a scalar)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

// n / d for a launch constant d: m = ceil(2^32 / d) from the host, exact while n * d < 2^32 (tile indices: n < 2^22, d < 2^10); d = 1 comes
// as m = 0, i = 1 (2^32 does not fit): branch-free, so that nothing in the prologue keeps the kernel-argument loads from being batched
__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned m, unsigned i) { return __umulhi(n, m) + n * i; }

// base pointer (hi or lo) of input channel group g for image b; groups past the end alias group 0 of in1
// (their packed weights are zero, the data only has to be finite)
__device__ __forceinline__ const uint4* in_plane(const ConvArgs& a, int g, int b, bool lo) {
    if (g < a.in0.ncg) return (lo ? a.in0.lo : a.in0.hi) + b * a.in0.bs + g * a.in0.cs;
    int g1 = g - a.in0.ncg;
    if (g1 >= a.in1.ncg) g1 = 0;
    return (lo ? a.in1.lo : a.in1.hi) + b * a.in1.bs + g1 * a.in1.cs;
}

// The activation copies of a tile: up to MAXS slots (64 pixel vectors = 1 KiB each) per wave and plane.  soff = the lane's source BYTE offset
// inside a plane (~0: a lane past the tile's last pixel vector, copies nothing); slot = the LDS slot it fills (uniform).  Pixels of the
// flattened tile that lie outside the padded image read the plane's (0,0) border vector, which is zero.
template <int MAXS>
struct FetchState {
    unsigned soff[MAXS];
    int slot[MAXS];
};

// Flattened-tile pixel p -> (row, column) with the pitch division as a multiplication (m_P = ceil(2^20 / P): exact for p * P < 2^20, and
// p < 1024, P <= 386), nearest-upsample source coordinate (c - 1 + ups) / ups the same way (m_ups = ceil(2^16 / ups): exact for
// coordinates < 2^15, checked by the host; ups = 1: the identity).
template <int MAXS>
__device__ __forceinline__ FetchState<MAXS> setup_tile(const ConvArgs& a, int x0, int y0, int wave, int lane) {
    FetchState<MAXS> f;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        // (a slot index past the tile re-fetches this wave's first slot; dma_chunk never issues it: the wave's share says how many it owns)
        f.slot[s] = (wave + s * NW) < a.nslots ? wave + s * NW : wave;
        const unsigned p = (unsigned)f.slot[s] * 64 + lane;
        const unsigned rr = __umul24(p, a.m_P) >> 20, cc = p - rr * a.P;
        const unsigned Yp = y0 + rr, Xp = x0 + cc;
        const bool inb = (p < (unsigned)a.NPIX_T) && (Yp < (unsigned)a.H + 2) && (Xp < (unsigned)a.W + 2);
        const unsigned sy = __umul24(Yp + a.ups - 1, a.m_ups) >> 16, sx = __umul24(Xp + a.ups - 1, a.m_ups) >> 16;
        // (the pad of the last slot is not copied: it would land in the next plane.  Rounding the planes up to whole slots instead — no
        // per-lane predicate at all — measured +2.6 % on the configs[1] forward: 7 % more bytes into LDS under the power cap)
        f.soff[s] = p >= (unsigned)a.NPIX_L ? ~0u : (inb ? (sy * a.Win_p + sx) * 16 : 0);
    }
    return f;
}

// source bases of one step: the 2*NPL input planes (group-major, hi|lo) of chunk cp in image b, and the chunk's weight fragments
template <int NPL>
struct Bases {
    const uint4* p[2 * NPL];
    const uint4* w;
    int wtap;
};
template <int NPL, int MT, int NPW>
__device__ __forceinline__ Bases<NPL> make_bases(const ConvArgs& a, int cp, int b) {
    Bases<NPL> r;
#pragma unroll
    for (int i = 0; i < 2 * NPL; ++i) r.p[i] = in_plane(a, 2 * cp + i / NPL, b, (i % NPL) == 1);
    r.w = a.wpack + (size_t)cp * a.wchunk;                     // uniform: the lane's 16 bytes are the copy's per-lane offset
    r.wtap = a.wtap;
    return r;
}

// Which copies of a chunk this wave issues.  A chunk is 2*NPL activation planes x `nslots` 1-KiB slots plus NWI 1-KiB weight fragments; wave w
// owns the slots w, w + NW, ... (setup_tile) and a contiguous range of weight fragments sized so that every wave issues the same number of
// copies (+-1): a 1-KiB global_load_lds occupies its in-order wave for 90-150 cycles (profiles/microbench/ingest_paths.hip), the barrier
// behind the copies waits for the slowest wave, and nothing is fetched twice.  Computed on the HOST per launch (dma_share_host; the kernel
// reads its wave's packed word from the kernel arguments).
struct DmaShare {
    int nsl;                   // activation slots of this wave
    int w0, wc;                // its weight fragments [w0, w0 + wc)
};
__device__ __forceinline__ DmaShare unpack_share(unsigned w) { return DmaShare{(int)(w & 0xFF), (int)((w >> 8) & 0xFF), (int)(w >> 16)}; }
// number of copies dma_chunk() issues (for the counted waits of the two-stage kernels)
// TMODE != 0 (tap-masked kernels): only the 4 * MT * NPW fragments of the chunk's live taps are copied, wave k those of the k-th live tap
// (per M tile: its own k-th live tap) — MT * NPW copies per wave whatever the chunk's tap set is
template <int NPL, int MT = 1, int NPW = 1, int TMODE = 0>
__device__ __forceinline__ int dma_count(const DmaShare& d, bool xlo) { return (xlo ? 2 * NPL : 2) * d.nsl + (TMODE != 0 ? MT * NPW : d.wc); }

// tsel (TMODE 1: the chunk's tap-set index (cp >> 1) & 3; TMODE 2: parity of the output slice): the live taps of an embedded stride-2 conv are
// the 2x2 block of taps at (r0, c0): S2D_FWD[q] -> (1 - (q >> 1), 1 - (q & 1)), S2D_FLIP[q] -> (q >> 1, q & 1) with q = 2 * parity + m.  The dead
// taps' fragments (5 of 9: zeros in the pack) are neither copied nor read — on the 512-channel layers the weight copies ARE the launch.
template <int NPL, int MT, int NPW, int TMODE = 0>
__device__ __forceinline__ void dma_chunk(const FetchState<maxs_of(MT)>& f, const Bases<NPL>& bs, const DmaShare& d, unsigned stage, int plane_bytes, bool xlo,
                                          int tsel = 0, int wave = 0) {
    constexpr int MAXS = maxs_of(MT);
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        if (s >= d.nsl) break;                          // wave-uniform
        const unsigned dst = stage + (unsigned)f.slot[s] * 1024;
        if (f.soff[s] != ~0u) {                         // per lane; one predicate per slot, not per copy
#pragma unroll
            for (int cgpl = 0; cgpl < 2 * NPL; ++cgpl) {
                if (!xlo && (cgpl % NPL) == 1) continue;    // this chunk's groups have no lo plane
                glds16s(bs.p[cgpl], f.soff[s], dst + cgpl * plane_bytes);
            }
        }
    }
    const unsigned vlane = (unsigned)(threadIdx.x & 63) * 16;
    if constexpr (TMODE == 0 && MT == 1) {
        // (the pack may be a wider layer's: fragment j = tap * NPW + plane sits (tap * wtap + plane) fragments in — ConvArgs.wtap)
        for (int j = d.w0; j < d.w0 + d.wc; ++j) glds16s(bs.w + ((j / NPW) * bs.wtap + j % NPW) * 64, vlane, stage + 2 * NPL * plane_bytes + j * 1024);
    } else if constexpr (TMODE == 0) {
        for (int j = d.w0; j < d.w0 + d.wc; ++j) glds16s(bs.w + j * 64, vlane, stage + 2 * NPL * plane_bytes + j * 1024);
    } else {
        static_assert(TMODE == 0 || NW == 4, "one live tap per wave");
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r0 = TMODE == 1 ? 1 - (tsel >> 1) : tsel, c0 = TMODE == 1 ? 1 - (tsel & 1) : m;
            const int tap = (r0 + (wave >> 1)) * 3 + c0 + (wave & 1);
#pragma unroll
            for (int pl = 0; pl < NPW; ++pl) {
                const int j = (tap * MT + m) * NPW + pl;
                glds16s(bs.w + j * 64, vlane, stage + 2 * NPL * plane_bytes + j * 1024);
            }
        }
    }
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate).  MAXN = the largest count the calling kernel can ask
// for (its waves' copy shares are bounded by the tile format): the cases above it are not compiled.
template <int MAXN>
__device__ __forceinline__ void wait_vm_upto(int n) {
#define ESR_VMC(k) case k: if constexpr (k <= MAXN) { asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break; }
    switch (n) {
        ESR_VMC(1) ESR_VMC(2) ESR_VMC(3) ESR_VMC(4) ESR_VMC(5) ESR_VMC(6) ESR_VMC(7) ESR_VMC(8) ESR_VMC(9) ESR_VMC(10) ESR_VMC(11) ESR_VMC(12)
        ESR_VMC(13) ESR_VMC(14) ESR_VMC(15) ESR_VMC(16) ESR_VMC(17) ESR_VMC(18) ESR_VMC(19) ESR_VMC(20) ESR_VMC(21) ESR_VMC(22) ESR_VMC(23)
        ESR_VMC(24) ESR_VMC(25) ESR_VMC(26) ESR_VMC(27) ESR_VMC(28) ESR_VMC(29) ESR_VMC(30) ESR_VMC(31) ESR_VMC(32)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;       // 0, or more than the cases cover: wait for everything (always safe)
    }
#undef ESR_VMC
}

// Residual / mask operand of one PAIR of channel groups (cg0, cg0+1) at this lane's pixel, read the way the output is stored:
// lanes 0-31 load the full 16-byte pixel vector of group cg0, lanes 32-63 that of group cg0+1 (one coalesced b128 load per
// plane, uniform per-image base + 32-bit lane offset); res_unpack() then exchanges halves (v_permlane32_swap) into the accumulator
// arrangement: this lane's 4 channels (4*half .. 4*half+3) of both groups.
struct ResRaw { uint4 h, l; };
__device__ __forceinline__ void swap_halves(const uint4& x, uint32_t (&d)[2][2]) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(x.x, x.z, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(x.y, x.w, false, false);
    d[0][0] = s0[0]; d[0][1] = s1[0]; d[1][0] = s0[1]; d[1][1] = s1[1];
}
template <int FMT>
__device__ __forceinline__ void res_unpack(const ResRaw& q, bool has_lo, f32x2 (&rv)[2][2]) {
    uint32_t d[2][2];
    swap_halves(q.h, d);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) rv[k][j] = f32x2{e2f<FMT>(d[k][j] & 0xFFFF), e2f<FMT>(d[k][j] >> 16)};
    if (has_lo) {
        swap_halves(q.l, d);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 2; ++j) rv[k][j] += f32x2{e2f<FMT>(d[k][j] & 0xFFFF), e2f<FMT>(d[k][j] >> 16)};
    }
}

// EPI_RESIN: the two input groups of chunk cp are in the LDS stage right now; if they belong to the residual slice, add this lane's 4
// channels of the centre-tap pixel (exactly hi + lo, in fp32) to the matching accumulator rows.
template <int NPL, int MT, int R, int FMT>
__device__ __forceinline__ void resin_accumulate(f32x16 (&acc)[MT][R], const ConvArgs& a, const unsigned char* stage, int cp, bool xlo, int P,
                                                 int plane_bytes, int wave, int lane) {
#pragma unroll
    for (int sgrp = 0; sgrp < 2; ++sgrp) {
        const int og = 2 * cp + sgrp - a.resin_g0;              // output group fed by this input group (uniform)
        if (og < 0 || og * 8 >= a.cout) continue;
        float x[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned char* const pr = stage + sgrp * NPL * plane_bytes + ((wave + r * NW) * 32 + (lane & 31) + P + 1) * 16 + (lane >> 5) * 8;
            const uint2 h = *(const uint2*)pr;
            x[r][0] = e2f<FMT>(h.x & 0xFFFF); x[r][1] = e2f<FMT>(h.x >> 16); x[r][2] = e2f<FMT>(h.y & 0xFFFF); x[r][3] = e2f<FMT>(h.y >> 16);
            if (NPL == 2 && xlo) {
                const uint2 l = *(const uint2*)(pr + plane_bytes);
                x[r][0] += e2f<FMT>(l.x & 0xFFFF); x[r][1] += e2f<FMT>(l.x >> 16); x[r][2] += e2f<FMT>(l.y & 0xFFFF); x[r][3] += e2f<FMT>(l.y >> 16);
            }
        }
        // (uniform switch with the row group as a compile-time constant per case: accumulator rows are register indices — an if-chain over an
        // unrolled index gets re-rolled into a run-time index, which sends the whole accumulator array to scratch)
        auto add = [&](auto MG) {
            constexpr int mg = decltype(MG)::value;
            if constexpr (mg < MT * 4) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[mg / 4][r][(mg % 4) * 4 + i] = fmaf(a.resin_scale, x[r][i], acc[mg / 4][r][(mg % 4) * 4 + i]);
            }
        };
        switch (og) {
            case 0: add(std::integral_constant<int, 0>{}); break;
            case 1: add(std::integral_constant<int, 1>{}); break;
            case 2: add(std::integral_constant<int, 2>{}); break;
            case 3: add(std::integral_constant<int, 3>{}); break;
            case 4: add(std::integral_constant<int, 4>{}); break;
            case 5: add(std::integral_constant<int, 5>{}); break;
            case 6: add(std::integral_constant<int, 6>{}); break;
            default: add(std::integral_constant<int, 7>{}); break;
        }
    }
}

// The MFMAs of one chunk (2 channel groups x 9 taps) out of one LDS stage, with the fragment reads of tap t+1 interleaved between the
// MFMAs of tap t (sched_barrier-pinned).  XLO: the chunk's activations have a lo plane.  Terms per product, in issue order:
// Wlo*Xhi (if the weights have a lo plane), Whi*Xlo (if XLO), Whi*Xhi.
// TM0 / TM1 (compile time): 9-bit masks of the taps whose weights are not structurally zero for M tile 0 / 1 of this chunk; the unrolled
// loops below drop the dead MFMAs and the fragment reads nobody needs (no run-time branches: those cost more than the MFMAs they save)
template <int NPL, int MT, int R, int NPW, int FMT, bool XLO, int NTERM_CAP, int TM0 = 0x1FF, int TM1 = 0x1FF>
__device__ __forceinline__ void chunk_mfma(f32x16 (&acc)[MT][R], const unsigned char* sa, const unsigned char* sb, int P, int plane_bytes) {
    constexpr int TMU = TM0 | (MT == 2 ? TM1 : 0);      // taps any M tile needs: the activation fragments to read
#define ESR_TAP_LIVE(t, m) ((((m) == 0 ? TM0 : TM1) >> (t)) & 1)
    constexpr int NPB = XLO ? NPL : 1;                                   // activation planes read
    constexpr int NT_FULL = 1 + (NPW == 2 ? 1 : 0) + (NPB == 2 ? 1 : 0);
    constexpr int NTERM = NT_FULL < NTERM_CAP ? NT_FULL : NTERM_CAP;      // NTERM_CAP < 3 only in ablation builds
    constexpr int NM = MT * R * NTERM;
    constexpr int NLA = MT * NPW, NLB = R * NPB, NL = NLA + NLB;
    constexpr int NSLOT = NM > NL ? NM : NL;
    uint4 fa[2][MT][NPW], fb[2][R][NPB];
    // read order inside a tap: [A plane of the first term x MT, B hi x R, then the other A plane x MT (if any), B lo x R (if any)] — what
    // the first MFMAs of the next tap need comes first
    auto load_frag = [&](int t, int k, int buf) {
        const int tapoff = ((t / 3) * P + (t % 3)) * 16;
        if (!((TMU >> t) & 1)) return;               // nobody multiplies this tap
        if (k < MT) {
            const int pl = NPW == 2 ? 1 : 0;
            if (ESR_TAP_LIVE(t, k)) fa[buf][k][pl] = *(const uint4*)(sa + ((t * MT + k) * NPW + pl) * 1024);
        } else if (k < MT + R) {
            fb[buf][k - MT][0] = *(const uint4*)(sb + (k - MT) * NW * 512 + tapoff);
        } else if (NPW == 2 && k < 2 * MT + R) {
            const int idx = k - MT - R;
            if (ESR_TAP_LIVE(t, idx)) fa[buf][idx][0] = *(const uint4*)(sa + ((t * MT + idx) * NPW) * 1024);
        } else {
            const int idx = k - (NPW == 2 ? 2 * MT + R : MT + R);
            fb[buf][idx][NPB - 1] = *(const uint4*)(sb + idx * NW * 512 + tapoff + (NPB - 1) * plane_bytes);
        }
    };
#pragma unroll
    for (int k = 0; k < NL; ++k) load_frag(0, k, 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int cb = t & 1;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            if (i < NM) {
                // the NTERM terms are the LAST NTERM entries of [Wlo*Xhi (needs NPW == 2), Whi*Xlo (needs XLO), Whi*Xhi]
                const int ti = i / (MT * R), rem = i % (MT * R), r = rem % R, m = rem / R;
                constexpr int has0 = NPW == 2 ? 1 : 0, has1 = NPB == 2 ? 1 : 0;
                const int skip = NT_FULL - NTERM;                             // ablation: drop leading terms
                const int idx = ti + skip;                                    // index into the present-term list
                const int term = (idx < has0) ? 0 : ((idx < has0 + has1) ? 1 : 2);
                const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
                if (!ESR_TAP_LIVE(t, m)) {
                    // this tap's weights for M tile m are structurally zero (compile-time: t and m are unrolled constants)
                } else acc[m][r] = mfma<FMT>(fa[cb][m][pa], fb[cb][r][pb], acc[m][r]);
            }
            if (t < 8 && i < NL) load_frag(t + 1, i, cb ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

#undef ESR_TAP_LIVE

// tap masks of the critic's stride-2 convs run as 3x3 convs over the space-to-depth input (esr_hip/critic.py): by parity s = 2 py + px of a
// 32-channel tile, the non-zero taps of the embedded weight (bit 3 ty + tx) — and of its flipped / transposed form (data gradient)
constexpr int S2D_FWD[4] = {432, 216, 54, 27}, S2D_FLIP[4] = {27, 54, 216, 432};

// ---- epilogue.  D layout (32x32 MFMA): lane holds pixel column j = lane&31 and, for register i, output row (i&3) + 8*(i>>2) + 4*(lane>>5):
// i>>2 selects the 8-channel group inside the 32-row tile, (i&3) + 4*(lane>>5) the channel inside the group -> 4 consecutive channels = 8 bytes
// of bf16.  Two groups are paired through v_permlane32_swap so that every lane stores one full 16-byte pixel vector: lanes 0-31 group cg0's,
// lanes 32-63 group cg0+1's (same pixel).
//
// What depends only on (tile, lane) is computed ONCE, in front of the K loop where a lone workgroup waits for its first copies anyway
// (epi_coords): per column tile the lane's byte offset inside an activation plane and `lim` = how many output groups the lane may store
// (0: its pixel is pitch padding or outside the image; ncg_out - half otherwise, so that one compare `cg0 < lim` covers both the pixel and
// the existence of group cg0 + half).  KIND 1 (fp32 NCHW destination): poff = byte offset inside a channel plane, lim without the half
// term (every lane stores its own 4 channels of both groups); KIND 2 (pixel-shuffle store): poff = Y << 16 | X.
template <int R>
struct EpiCoord {
    int lim[R];
    unsigned poff[R];
};
// th: rows of the tile that are this pass's to store (the chain kernel's partial passes); < 0: the tile height
template <int R, int KIND>
__device__ __forceinline__ EpiCoord<R> epi_coords(const ConvArgs& a, int x0, int y0, int wave, int lane, int th = -1) {
    EpiCoord<R> e;
    const int half = lane >> 5;
    const unsigned TH = th < 0 ? (unsigned)a.TH : (unsigned)th;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned q = (wave + r * NW) * 32 + (lane & 31);
        const unsigned rr = __umul24(q, a.m_P) >> 20, cc = q - rr * a.P;
        const unsigned Y = y0 + rr, X = x0 + cc;
        const bool valid = (rr < TH) && (cc < (unsigned)a.TW) && (Y < (unsigned)a.H) && (X < (unsigned)a.W);
        e.lim[r] = valid ? (KIND == 1 ? a.ncg_out : a.ncg_out - half) : 0;
        if (KIND == 1) e.poff[r] = (__umul24(Y, a.W) + X) * 4;
        else if (KIND == 2) e.poff[r] = (Y << 16) | X;
        else e.poff[r] = valid ? (__umul24(Y + 1, a.W + 2) + X + 1) * 16 : 0;
    }
    return e;
}

template <int NPL, int MT, int R, int EPI, int FMT, bool PARTLO>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MT][R], const int b, const EpiCoord<R>& ec, const int lane) {
    constexpr bool HAS_R1 = (EPI & EPI_RES1) != 0, HAS_R2 = (EPI & EPI_RES2) != 0, HAS_MK = (EPI & EPI_MASK) != 0;
    constexpr bool NCHW = (EPI & EPI_NCHW) != 0, OUT2 = (EPI & EPI_OUT2) != 0, PS = (EPI & EPI_PS) != 0;
    const int half = lane >> 5;
    const long long bl = b;
    const f32x2 slope2 = {a.act_slope, a.act_slope}, alpha2 = {a.alpha, a.alpha};
    // per-image plane bases (uniform: SGPR pairs) and the half-wave's group stride (one VGPR per view): every access below is
    // base + 32-bit lane offset (the host checked that a view's image fits 2^32 bytes)
    const char *r1h = nullptr, *r1l = nullptr, *r2h = nullptr, *r2l = nullptr, *mkh = nullptr;
    unsigned h1 = 0, h2 = 0, ho = 0, ho2 = 0;
    if constexpr (HAS_R1) {
        r1h = (const char*)(a.res1.hi + bl * a.res1.bs);
        r1l = a.res1.lo ? (const char*)(a.res1.lo + bl * a.res1.bs) : nullptr;
        h1 = half ? (unsigned)a.res1.cs * 16 : 0;
    }
    if constexpr (HAS_R2) {
        r2h = (const char*)(a.res2.hi + bl * a.res2.bs);
        r2l = a.res2.lo ? (const char*)(a.res2.lo + bl * a.res2.bs) : nullptr;
        h2 = half ? (unsigned)a.res2.cs * 16 : 0;
    }
    if constexpr (HAS_MK) {
        mkh = (const char*)(a.mask.hi + bl * a.mask.bs);
    }
    char *oh = nullptr, *ol = nullptr, *o2h = nullptr, *o2l = nullptr;
    if constexpr (!NCHW) {
        oh = (char*)(a.out.hi + bl * a.out.bs);
        ol = (NPL == 2 && a.out.lo) ? (char*)(a.out.lo + bl * a.out.bs) : nullptr;
        ho = half ? (unsigned)a.out.cs * 16 : 0;
        if constexpr (OUT2) {
            o2h = (char*)(a.out2.hi + bl * a.out2.bs);
            o2l = (NPL == 2 && a.out2.lo) ? (char*)(a.out2.lo + bl * a.out2.bs) : nullptr;       // (a hi-only second destination: the mask stash)
            ho2 = half ? (unsigned)a.out2.cs * 16 : 0;
        }
    }
    // Residual / mask operands: 16-byte loads, all of a column tile's (or, where the registers allow, of the whole tile's) issued in one go
    // before the math that uses them.  A lane without an output reads its view's first vector (always mapped) and drops it.
    constexpr int OPREGS_R = MT * 2 * (((HAS_R1 ? 1 : 0) + (HAS_R2 ? 1 : 0)) * NPL + (HAS_MK ? 1 : 0)) * 4;
    constexpr bool ALL_FIRST = OPREGS_R * R <= 96;
    constexpr int RQ = ALL_FIRST ? R : 1;
    ResRaw q1[HAS_R1 ? RQ : 1][MT * 2], q2[HAS_R2 ? RQ : 1][MT * 2];
    uint4 qm[HAS_MK ? RQ : 1][MT * 2];
    unsigned big = 0;              // fp16 range watch: bit 15 / 31 set once a stored half had magnitude >= 2^15 (exponent field >= 30, inf and NaN included)
    auto issue = [&](const int r) {
        const int rq = ALL_FIRST ? r : 0;
#pragma unroll
        for (int mp = 0; mp < MT * 2; ++mp) {
            const int cg0 = mp * 2;
            const bool ok = cg0 < ec.lim[r];
            if constexpr (HAS_R1) {
                const unsigned off = ok ? ec.poff[r] + h1 + cg0 * ((unsigned)a.res1.cs * 16) : 0;
                q1[rq][mp].h = *(const uint4*)(r1h + off);
                if (NPL == 2 && r1l) q1[rq][mp].l = *(const uint4*)(r1l + off);
            }
            if constexpr (HAS_R2) {
                const unsigned off = ok ? ec.poff[r] + h2 + cg0 * ((unsigned)a.res2.cs * 16) : 0;
                q2[rq][mp].h = *(const uint4*)(r2h + off);
                if (NPL == 2 && r2l) q2[rq][mp].l = *(const uint4*)(r2l + off);
            }
            if constexpr (HAS_MK) {
                const int cgm = cg0 + half - a.mask_cg0;                  // the lane's group inside the mask view
                const bool okm = ok && cgm >= 0 && cg0 + half < a.mask_cg1;
                const unsigned off = okm ? ec.poff[r] + (unsigned)cgm * ((unsigned)a.mask.cs * 16) : 0;
                qm[rq][mp] = *(const uint4*)(mkh + off);
            }
        }
    };
    if constexpr (ALL_FIRST && (HAS_R1 || HAS_R2 || HAS_MK)) {
#pragma unroll
        for (int r = 0; r < R; ++r) issue(r);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if constexpr (!ALL_FIRST && (HAS_R1 || HAS_R2 || HAS_MK)) issue(r);
        const int rq = ALL_FIRST ? r : 0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const int cg0 = m * 4 + gp * 2;                  // this pair: output groups cg0, cg0+1
                if (!(cg0 < ec.lim[r])) continue;                // per lane: pixel inside the image and group cg0 + half exists
                f32x2 v[2][2];                                   // [group k][channel pair]: this lane's 4 channels of both groups
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        // alpha * LeakyReLU(y) = max(alpha * y, alpha * slope * y) for alpha >= 0, 0 < slope <= 1 (checked by the host); the bias is
                        // the accumulators' seed.  (Scaling first makes both operands of the max products: no canonicalising v_max x, x.)
                        v[k][j] = f32x2{acc[m][r][(gp * 2 + k) * 4 + 2 * j], acc[m][r][(gp * 2 + k) * 4 + 2 * j + 1]} * alpha2;
                        v[k][j] = __builtin_elementwise_max(v[k][j], v[k][j] * slope2);
                    }
                if constexpr (HAS_R1) {
                    f32x2 rv[2][2];
                    res_unpack<FMT>(q1[rq][m * 2 + gp], NPL == 2 && a.res1.lo != nullptr, rv);
                    const f32x2 bb = {a.beta1, a.beta1};
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int j = 0; j < 2; ++j) v[k][j] = __builtin_elementwise_fma(bb, rv[k][j], v[k][j]);
                }
                if constexpr (HAS_R2) {
                    f32x2 rv[2][2];
                    res_unpack<FMT>(q2[rq][m * 2 + gp], NPL == 2 && a.res2.lo != nullptr, rv);
                    const f32x2 bb = {a.beta2, a.beta2};
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int j = 0; j < 2; ++j) v[k][j] = __builtin_elementwise_fma(bb, rv[k][j], v[k][j]);
                }
                if constexpr (HAS_MK) {
                    // LeakyReLU' from the stored post-activation value: its sign is the pre-activation's (slope > 0);
                    // x <= 0 -> slope (torch: leaky_relu'(0) = slope).  16-bit elements, two per dword: the low one is positive iff
                    // (int)(d << 16) > 0, the high one iff (int)d > 0xFFFF (sign clear, magnitude bits not all zero) — bf16 and f16 alike
                    uint32_t d[2][2];
                    swap_halves(qm[rq][m * 2 + gp], d);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int cg = cg0 + k;
                        const float ms = (cg < a.mask_cg0 || cg >= a.mask_cg1) ? 1.f : a.mask_slope;      // uniform: groups outside the masked range keep their value
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const f32x2 s = v[k][j] * f32x2{ms, ms};
                            v[k][j].x = (int)(d[k][j] << 16) > 0 ? v[k][j].x : s.x;
                            v[k][j].y = (int)d[k][j] > 0xFFFF ? v[k][j].y : s.y;
                        }
                    }
                }
                if constexpr (NCHW) {
                    // fp32 [B][nchw_ctot][H][W]: this lane's 4 channels of both groups (no exchange)
                    char* const ob = (char*)(a.out_nchw + bl * a.nchw_ctot * a.H * a.W);
                    const unsigned hw4 = (unsigned)(a.H * a.W) * 4;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int ch0 = (cg0 + k) * 8 + half * 4;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (ch0 + i < a.cout) *(float*)(ob + (ec.poff[r] + (unsigned)(ch0 + i) * hw4)) = v[k][i >> 1][i & 1];
                    }
                    continue;                                    // the fp32 NCHW destination replaces the act-layout one
                }
                // to 16-bit hi (+ lo = the rounding residue) elements (v_cvt_pk_bf16_f32 rounds to nearest even), 2 channels per dword.  Rows past
                // cout need no masking: their weights and their bias seed are zero, so is whatever the residual buffers hold there.
                uint32_t hi[2][2], lo[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint32_t h = cvt_pk<FMT>(v[k][j].x, v[k][j].y);
                        hi[k][j] = h;
                        if constexpr (FMT == 1) big |= (h & 0x7FFF7FFFu) + 0x08000800u;      // 15-bit magnitude >= 0x7800 carries into the half's top bit
                        lo[k][j] = 0;
                        if (NPL == 2) lo[k][j] = cvt_pk<FMT>(v[k][j].x - e2f<FMT>(h & 0xFFFF), v[k][j].y - e2f<FMT>(h >> 16));
                    }
                // lanes 0-31 end up with group cg0's 8 channels, lanes 32-63 with group cg0+1's (same pixel)
                const auto s0 = __builtin_amdgcn_permlane32_swap(hi[0][0], hi[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(hi[0][1], hi[1][1], false, false);
                const uint4 hv = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                uint4 lv = hv;
                if (NPL == 2) {
                    const auto t0 = __builtin_amdgcn_permlane32_swap(lo[0][0], lo[1][0], false, false);
                    const auto t1 = __builtin_amdgcn_permlane32_swap(lo[0][1], lo[1][1], false, false);
                    lv = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                }
                unsigned off;
                if constexpr (PS) {
                    // row group -> (output group, sub-position) of the r x r block at (Y, X)
                    const int cgs = cg0 + half, rg = a.ps_rg0 + cgs, r2 = a.ps * a.ps, sp = rg % r2;
                    const int Y = ec.poff[r] >> 16, X = ec.poff[r] & 0xFFFF;
                    off = ((unsigned)(rg / r2) * (unsigned)a.out.cs + (unsigned)(a.ps * Y + sp / a.ps + 1) * (a.ps * a.W + 2) + (a.ps * X + sp % a.ps + 1)) * 16;
                } else off = ec.poff[r] + ho + cg0 * ((unsigned)a.out.cs * 16);
                *(uint4*)(oh + off) = hv;
                if (NPL == 2 && (!PARTLO || ol)) *(uint4*)(ol + off) = lv;
                if constexpr (OUT2) {
                    const unsigned off2 = ec.poff[r] + ho2 + cg0 * ((unsigned)a.out2.cs * 16);
                    *(uint4*)(o2h + off2) = hv;
                    if (NPL == 2 && o2l) *(uint4*)(o2l + off2) = lv;
                }
            }
        }
    }
    if constexpr (FMT == 1 && !NCHW) {
        if (a.range_flag && (big & 0x80008000u)) atomicMin(a.range_flag, a.range_tag);      // (no lane gets here in a pass that stays in range)
    }
}

// The accumulators' seed: the bias of this lane's 16 rows per M tile (esr_conv3x3_desc.bias: MT * 32 floats, zero beyond cout; the library's
// zero block without a bias).  Scalar loads (the constant address space: s_load_dwordx8 per 8-channel block, on their own counter — the hand-counted vmcnt of the
// copies is not involved), then one select per value on the half-wave.
template <int MT>
__device__ __forceinline__ void bias_seed(const float* bias, int half, float (&bz)[MT][16]) {
    typedef const __attribute__((address_space(4))) float* cptr;
    const cptr cb = (cptr)(uintptr_t)bias;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c0 = (m * 4 + j) * 8;
                const float lo = cb[c0 + k], hi = cb[c0 + 4 + k];
                bz[m][j * 4 + k] = half ? hi : lo;
            }
}

}  // namespace

// esr_chain.hip: launches conv3x3_chain_kernel for `variant` (a ConvVariant; ESR_E_UNSUPPORTED when that operand form has no chain instantiation:
// the caller then issues the separate launches; query_only: answer without launching).  chain_args: a ChainArgs (passed untyped: the type lives in each unit's anonymous namespace).
// Hidden: an internal link between two translation units of the library, not part of the C-ABI.
__attribute__((visibility("hidden"))) int esr_internal_chain_launch(const void* chain_args, const int* variant, hipStream_t stream, int query_only);
