// conv3x3 (+bias +LeakyReLU +scaled residuals, fused nearest-upsample of the input) as an implicit GEMM on
// the gfx950 bf16 MFMA pipe, fp32 accumulate, "split-bf16" operands for fp32-class accuracy.
//
// Replaces codes/models/modules/block.py:129-146,230-235,262-270,85-97,293-309 of the reference
// (see include/esr_hip.h).  Design notes live in DESIGN.md §conv3x3; the short version:
//
//   GEMM view      D[cout][pixel] = sum_{tap, cin}  W[cout][tap,cin] * X[tap,cin][pixel]
//                  A = weights (M = cout, 32 per MFMA), B = activations (N = 32 consecutive pixels),
//                  K = 16 = two 8-channel groups at one tap  ->  v_mfma_f32_32x32x16_bf16
//   activations    [B][CG][H+2][W+2][8] bf16 hi (+lo) planes with a zero border in memory: one B fragment is one
//                  aligned ds_read_b128 of an LDS pixel vector, a tap shift is +-16 B, padding needs no branches
//   tile           TH x TW output pixels per workgroup, flattened with pitch P = TW+2 so that the 32-pixel MFMA
//                  columns are 32 CONSECUTIVE LDS vectors for every tap (conflict-free b128 reads); the 2
//                  pitch-padding columns compute garbage that is masked at the store
//   execution      one tile per 4-wave workgroup, K loop over chunks of 2 channel groups x 9 taps: LDS-DMA (global_load_lds) of the
//                  chunk's input tile AND weight fragments -> barrier -> MFMAs out of LDS -> barrier.  Large launches: one LDS
//                  stage, two workgroups per CU cover each other's DMA waits and epilogues (NST = 1).  Launches with no more
//                  tiles than CUs: two LDS stages, the next chunk's DMA in flight under the MFMAs (NST = 2).  A persistent
//                  multi-stage variant and 8-wave workgroups were measured and dropped (DESIGN.md sections 4 and 5).
//   precision      element format FMT (bf16 / f16) x activation planes NPL: bf16 hi+lo = the fp32-class mode below; f16 hi+lo
//                  with single-plane f16 weights = 2 MFMAs; one plane = 1 MFMA (precision table)
//   split-bf16     x = hi + lo (both bf16).  acc += Wlo*Xhi + Whi*Xlo + Whi*Xhi  (3 MFMAs, lo*lo dropped: 2^-16)
//   epilogue       compile-time specialised (EPI bits): bias, LeakyReLU, alpha*y + beta1*r1 + beta2*r2, act' mask
//                  (data-gradient use), re-split to hi/lo via v_cvt_pk_bf16_f32, pairs of channel groups exchanged with
//                  v_permlane32_swap so that every lane stores one full 16-byte pixel vector
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
template <int R, int KIND>
__device__ __forceinline__ EpiCoord<R> epi_coords(const ConvArgs& a, int x0, int y0, int wave, int lane) {
    EpiCoord<R> e;
    const int half = lane >> 5;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned q = (wave + r * NW) * 32 + (lane & 31);
        const unsigned rr = __umul24(q, a.m_P) >> 20, cc = q - rr * a.P;
        const unsigned Y = y0 + rr, X = x0 + cc;
        const bool valid = (rr < (unsigned)a.TH) && (cc < (unsigned)a.TW) && (Y < (unsigned)a.H) && (X < (unsigned)a.W);
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

// One output tile per workgroup.
//   NST == 1: single LDS stage, 2 workgroups resident per CU: latency hiding comes from the co-resident workgroup instead of an
//             in-workgroup pipeline (the persistent multi-stage variant measured slower).  Large launches.
//   NST == 2: two LDS stages, the DMA of chunk c+1 is issued before the MFMAs of chunk c (counted s_waitcnt keeps it in flight).
//             Launches with no more tiles than CUs (small images, the 52x52 training crops), where a workgroup has its CU to itself
//             and nobody else covers its DMA waits.
//   NST == 4: a ring of four stages, three chunks of copies in flight: few, small tiles with a long K axis (the critic's deep layers).
// TMODE: 0 all taps; 1 the K chunks' tap sets follow S2D_FWD by the parity (cp >> 1) & 3 of their channel quad (forward of an embedded stride-2
// conv); 2 the M tiles' tap sets follow S2D_FLIP by the parity of output tile 2 * slice + m (its data gradient)
//
// Order of the prologue (round 5: a lone workgroup per CU pays every instruction in front of its first copy in full): tile decode and the
// slots' source offsets (multiplications by host-made magic numbers, no division), the first chunk's copies — and only then, while those
// are in flight, the bias seed of the accumulators and the epilogue's per-lane coordinates.
template <int NPL, int MT, int EPI, int NST, int FMT, int NPW, bool PARTLO, int TMODE = 0>
__global__ __launch_bounds__(NTHREADS, NST >= 2 ? 1 : (MT == 1 ? WGS_MT1 : WGS_MT2)) void conv3x3_tile_kernel(const ConvArgs a_in) {
    // Output slices (cout > 64; esr_conv3x3_desc): blockIdx.y selects a 64-channel slice of the output — its own weight pack and bias, the
    // same staged input.  All workgroups of all slices are in flight together: a 512-channel layer on an 8x8 map is one launch of
    // 32 x 8 workgroups instead of eight launches of 32.  (Everything below is uniform: the shifts are scalar adds; slice 0 adds zero.)
    ConvArgs a = a_in;
    {
        const long long sl = blockIdx.y;                            // a slice is this kernel's MT * 32 output channels
        a.wpack += sl * a.wslice;
        a.bias += sl * (MT * 32) * a.bias_stride;
        auto shift = [&](DView& v) { if (v.hi) { v.hi += sl * (MT * 4) * v.cs; if (v.lo) v.lo += sl * (MT * 4) * v.cs; } };
        shift(a.out); shift(a.out2); shift(a.res1); shift(a.res2); shift(a.mask);
        if constexpr ((EPI & EPI_RESIN) != 0) a.resin_g0 += (int)sl * (MT * 4);      // the slice's residual groups: further along the input
        if constexpr ((EPI & EPI_NCHW) != 0) {
            a.out_nchw += sl * (MT * 32) * (long long)a.H * a.W;           // (slices with an fp32 destination: the split-K partial sums, B == 1 per image row below)
            if (a.ksplit > 1) {
                const long long kz = blockIdx.z;
                a.in1.hi += kz * a.kz_groups * a.in1.cs;
                if (a.in1.lo) a.in1.lo += kz * a.kz_groups * a.in1.cs;
                a.wpack += kz * a.ncp * (9 * MT * NPW * 64);
                a.out_nchw += kz * a.kz_slab;
                if (kz) a.bias = a.zero_bias;
            }
        }
    }
    // PARTLO: only the first a.lo_chunks chunks of the input carry a lo plane (a dense block's trunk input), the rest are single-plane
    // intermediates; and the output's lo plane is optional.  Non-PARTLO kernels treat every chunk alike.
    static_assert(!PARTLO || NPL == 2, "partial lo needs hi+lo activations");
    // NPW = weight planes: with hi+lo activations, 2 planes = 3 MFMAs per product (Wlo*Xhi + Whi*Xlo + Whi*Xhi), 1 plane = 2
    static_assert(NPW <= NPL, "a lo weight plane needs hi+lo activations");
    constexpr int R = r_of(MT), MAXS = maxs_of(MT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = a.P;
    const int plane_bytes = a.NPIX_L * 16;
    constexpr int NWI = 9 * MT * NPW;                              // weight fragments staged in LDS per chunk
    constexpr int MAXCNT = 2 * NPL * MAXS + (NWI + NW - 1) / NW;      // most copies one wave issues per chunk (dma_share_host balances them)
    const int stage_bytes = 2 * NPL * plane_bytes + NWI * 1024;
    // XCD-aware tile order: workgroup g runs on XCD g%8; each XCD sweeps a contiguous range of the tile space.  The grid is exactly ntiles
    // workgroups (XCD x owns xcd_q tiles, one more if x < xcd_r): no idle workgroup, no early exit — the prologue is branch-free
    const unsigned xcd = blockIdx.x & 7;
    const unsigned tile_f = xcd * a.xcd_q + (xcd < (unsigned)a.xcd_r ? xcd : (unsigned)a.xcd_r) + (blockIdx.x >> 3);
    const unsigned tile = a.reverse ? a.ntiles - 1 - tile_f : tile_f;
    const unsigned trow = udiv_magic(tile, a.m_tx, a.i_tx);         // = image * tiles_y + tile row
    const int b = udiv_magic(trow, a.m_ty, a.i_ty);
    const int x0 = (tile - trow * a.tiles_x) * a.TW, y0 = (trow - b * a.tiles_y) * a.TH;   // tile origin: output interior coords == padded coords of the halo origin
    const DmaShare share = unpack_share(a_in.share[wave]);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#ifdef ESR_TRACE
    unsigned long long* const tr = a.trace ? a.trace + (size_t)blockIdx.x * 128 : nullptr;
    int tslot = 2;
    if (tr && tid == 0) { tr[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4); tr[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20); tr[126] = wall_clock64(); }
#define ESR_TR() do { if (tr && tid == 0 && tslot < 126) tr[tslot++] = __builtin_readcyclecounter(); } while (0)
    ESR_TR();                                    // entry stamp (slot 2): everything up to the first step stamp is the prologue
#else
#define ESR_TR() do { } while (0)
#endif
    const FetchState<MAXS> fs = setup_tile<MAXS>(a, x0, y0, wave, lane);
    ESR_TR();                                    // (slot 3) tile decoded, slot offsets formed
    static_assert(TMODE != 2 || MT == 2, "M-tile tap masks come in pairs");
    const unsigned char* const sb0 = smem + (lane >> 5) * NPL * plane_bytes + (wave * 32 + (lane & 31)) * 16;
    const unsigned char* const sa0 = smem + 2 * NPL * plane_bytes + lane * 16;
    constexpr int NTERM_CAP = 3;
    // which 2x2 block of taps chunk c's weights live in (dma_chunk)
    auto tsel_of = [&](const int c) { return TMODE == 1 ? ((c >> 1) & 3) : (TMODE == 2 ? (int)(blockIdx.y & 1) : 0); };
    auto issue = [&](const int c, const unsigned stage, const bool xlo) {
        const Bases<NPL> bs = make_bases<NPL, MT, NPW>(a, c, b);
        dma_chunk<NPL, MT, NPW, TMODE>(fs, bs, share, stage, plane_bytes, xlo, tsel_of(c), wave);
    };
    if (NST == 4) {                           // ring of four stages: chunks 0, 1, 2 in flight before the first multiply
        static_assert(NST != 4 || !PARTLO, "the four-stage ring counts its copies per chunk: one count for all chunks");
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (c < a.ncp) issue(c, lds0 + c * stage_bytes, true);
    }
    // What waits for nothing: done behind the first copies.  The accumulators start from the bias (their rows' seed); the epilogue's
    // coordinates are formed here when the registers allow (the lone-workgroup forms and the one-plane 32-channel kernels), otherwise after
    // the K loop (the 64-channel kernels of the large launches sit at their 256-register limit and a co-resident workgroup covers it).
    constexpr int EKIND = (EPI & EPI_NCHW) ? 1 : ((EPI & EPI_PS) ? 2 : 0);
    constexpr bool EARLY_COORDS = NST >= 2 || (MT == 1 && NPL == 1);
    f32x16 acc[MT][R];
    EpiCoord<R> ec;
    auto seed = [&]() {
        // (an opaque copy of the lane index: everything per-lane below is loop-invariant and would otherwise be hoisted in front of the copies)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        float bz[MT][16];
        bias_seed<MT>(a.bias, lane_o >> 5, bz);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][r][i] = bz[m][i];
        if constexpr (EARLY_COORDS) ec = epi_coords<R, EKIND>(a, x0, y0, wave, lane_o);
    };
    // The first chunk's copies go out in front of the K loop, the seed right behind them.  (Seeding inside the loop's first pass instead —
    // one copy of the issue code — made the accumulators' loop-carried registers VGPRs: 96 v_accvgpr moves per chunk, +0.35 us per chunk.)
    if (NST != 4) issue(0, lds0, !PARTLO || 0 < a.lo_chunks);
    __builtin_amdgcn_sched_barrier(0);
    ESR_TR();                                    // (slot 4) first chunk's copies issued
    seed();
    ESR_TR();                                    // (slot 5) accumulators seeded, epilogue coordinates formed
    // One chunk: DMA (or prefetch of the next chunk), barrier, MFMAs, barrier.  XLO (compile time): this chunk's activations have a lo
    // plane.  The chunks with a lo plane come first, so the K loop is two loops over the same step with XLO = true / false: a run-time
    // branch between the two MFMA bodies inside ONE loop made the register allocator spill (vgpr_spill 200-500 in the 64-channel kernels).
    auto step = [&](auto XLO_T, const int cp) {
        constexpr bool xlo = decltype(XLO_T)::value;
        const int st = NST == 4 ? (cp & 3) : (NST >= 2 ? (cp & 1) : 0);
        const unsigned char* const sb = sb0 + st * stage_bytes;
        const unsigned char* const sa = sa0 + st * stage_bytes;
        ESR_TR();
        if (NST == 1) {
            if (cp > 0) issue(cp, lds0, xlo);
            ESR_TR();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (NST == 4) {
            // NST == 4 (launches of few, small tiles with a long K axis — the critic's 512-channel layers on 8x8 / 4x4 maps): each chunk is a
            // handful of MFMAs behind a copy round trip, so the round trips of THREE chunks are kept in flight.  The stage refilled now
            // (chunk cp + 3) was last read in iteration cp - 1, closed by its trailing barrier.
            if (cp + 3 < a.ncp) issue(cp + 3, lds0 + ((cp + 3) & 3) * stage_bytes, true);
            const int ahead = a.ncp - 1 - cp < 3 ? a.ncp - 1 - cp : 3;         // chunks behind cp that stay in flight
            ESR_TR();
            wait_vm_upto<3 * MAXCNT < 32 ? 3 * MAXCNT : 32>(ahead * dma_count<NPL, MT, NPW, TMODE>(share, true));
        } else {
            // the other stage was last read in iteration cp-1 (closed by its trailing barrier): refill it now, then wait for
            // everything EXCEPT the copies just issued
            const bool more = cp + 1 < a.ncp;
            const bool xlo_next = !PARTLO || cp + 1 < a.lo_chunks;
            if (more) issue(cp + 1, lds0 + ((cp + 1) & 1) * stage_bytes, xlo_next);
            ESR_TR();
            wait_vm_upto<MAXCNT>(more ? dma_count<NPL, MT, NPW, TMODE>(share, xlo_next) : 0);
        }
        ESR_TR();
        __syncthreads();
        ESR_TR();
        if constexpr (TMODE == 1) {
            switch ((cp >> 1) & 3) {                 // uniform: four copies of the chunk body, each with its own compile-time tap set
                case 0: chunk_mfma<NPL, MT, R, NPW, FMT, xlo, NTERM_CAP, S2D_FWD[0], S2D_FWD[0]>(acc, sa, sb, P, plane_bytes); break;
                case 1: chunk_mfma<NPL, MT, R, NPW, FMT, xlo, NTERM_CAP, S2D_FWD[1], S2D_FWD[1]>(acc, sa, sb, P, plane_bytes); break;
                case 2: chunk_mfma<NPL, MT, R, NPW, FMT, xlo, NTERM_CAP, S2D_FWD[2], S2D_FWD[2]>(acc, sa, sb, P, plane_bytes); break;
                default: chunk_mfma<NPL, MT, R, NPW, FMT, xlo, NTERM_CAP, S2D_FWD[3], S2D_FWD[3]>(acc, sa, sb, P, plane_bytes); break;
            }
        } else if constexpr (TMODE == 2) {
            if (blockIdx.y & 1) chunk_mfma<NPL, MT, R, NPW, FMT, xlo, NTERM_CAP, S2D_FLIP[2], S2D_FLIP[3]>(acc, sa, sb, P, plane_bytes);
            else chunk_mfma<NPL, MT, R, NPW, FMT, xlo, NTERM_CAP, S2D_FLIP[0], S2D_FLIP[1]>(acc, sa, sb, P, plane_bytes);
        } else {
            chunk_mfma<NPL, MT, R, NPW, FMT, xlo, NTERM_CAP>(acc, sa, sb, P, plane_bytes);
        }
        if constexpr ((EPI & EPI_RESIN) != 0) resin_accumulate<NPL, MT, R, FMT>(acc, a, smem + st * stage_bytes, cp, xlo, P, plane_bytes, wave, lane);
        ESR_TR();
        __syncthreads();
    };
    const int lo_end = PARTLO ? (a.lo_chunks < a.ncp ? a.lo_chunks : a.ncp) : a.ncp;
    for (int cp = 0; cp < lo_end; ++cp) step(std::true_type{}, cp);
    if constexpr (PARTLO)
        for (int cp = lo_end; cp < a.ncp; ++cp) step(std::false_type{}, cp);
    ESR_TR();
    if constexpr (!EARLY_COORDS) ec = epi_coords<R, EKIND>(a, x0, y0, wave, lane);
    conv_epilogue<NPL, MT, R, EPI, FMT, PARTLO>(a, acc, b, ec, lane);
    ESR_TR();
#ifdef ESR_TRACE
    if (tr && tid == 0) tr[127] = wall_clock64();
#endif
}

// ---- weight packing: [M][K][3][3] fp32 -> [kstep = cp*9+tap][mtile][hi|lo][lane][8] bf16
__global__ void pack_weights_kernel(const float* __restrict__ w, int dim0, int dim1, const int* __restrict__ kmap, int ncg_in,
                                    const int* __restrict__ mmap, int mtiles, int transposed, int npl, int f16, float scale, uint4* __restrict__ out, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (kstep, mtile, lane)
    if (idx >= total) return;
    const int lane = idx & 63;
    const int m = (idx >> 6) % mtiles;
    const int ks = (idx >> 6) / mtiles;
    const int cp = ks / 9, t = ks % 9;
    const int cg = 2 * cp + (lane >> 5);
    const int mch = mmap[m * 32 + (lane & 31)];
    uint32_t hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kch = cg < ncg_in ? kmap[cg * 8 + e] : -1;
        float v = 0.f;
        if (kch >= 0 && mch >= 0)
            v = transposed ? w[((long long)kch * dim1 + mch) * 9 + (8 - t)] : w[((long long)mch * dim1 + kch) * 9 + t];
        if (f16) { hi[e] = f2h(v * scale); lo[e] = f2h(v * scale - h2f(hi[e])); }
        else split_bf16(v * scale, hi[e], lo[e]);
    }
    uint4* o = out + ((size_t)(ks * mtiles + m) * npl) * 64 + lane;
    o[0] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
    if (npl == 2) o[64] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
}

// many weight tensors in one launch (a training step re-packs every layer after the optimiser's update): block b packs tile map[b].y of entry
// map[b].x — one (K chunk, M tile) = 32 output x 16 input channels x 9 taps.  The tile's 4608 source floats are read ONCE, as the 32 (16 in the
// transposed pack) contiguous runs of 144 (288) floats they are in [M][K][3][3], into LDS; then every thread assembles the 16-byte vectors of
// the pack (lane = output channel x K half, 8 input channels each) from there.  (One thread per output vector gathering its 8 floats 36 bytes apart
// from global memory — every tap another thread, another block — fetched each 32-byte sector eight times: 171 us per configs[2] generator, 0.8 TB/s.)
struct PackEntry {
    const float* w; const int* kmap; const int* mmap; uint4* out;
    int dim0, dim1, ncg_in, mtiles, transposed, npl, f16, total; float scale;
};
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const PackEntry* __restrict__ table, const int2* __restrict__ map) {
    constexpr int PITCH = 16 * 9 + 1;                     // floats per output-channel row of the staged tile: odd, so the 32 rows fall into 32 banks
    __shared__ float tile[32 * PITCH];
    __shared__ int kch_s[16], mch_s[32];
    const int2 m = map[blockIdx.x];
    const PackEntry e = table[m.x];
    const int cp = m.y / e.mtiles, mt = m.y % e.mtiles;
    if (threadIdx.x < 16) {
        const int cg = 2 * cp + (threadIdx.x >> 3);
        kch_s[threadIdx.x] = cg < e.ncg_in ? e.kmap[cg * 8 + (threadIdx.x & 7)] : -1;
    } else if (threadIdx.x >= 32 && threadIdx.x < 64) mch_s[threadIdx.x - 32] = e.mmap[mt * 32 + (threadIdx.x - 32)];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 18; ++j) {
        const int f = j * 256 + threadIdx.x;              // walks the source runs: [32 rows][16 x 9] plain, [16 rows][32 x 9] transposed
        int mr, kc, t;
        if (e.transposed) { kc = f / 288; const int c = f - kc * 288; mr = c / 9; t = 8 - (c - mr * 9); }
        else { mr = f / 144; const int c = f - mr * 144; kc = c / 9; t = c - kc * 9; }
        const int kch = kch_s[kc], mch = mch_s[mr];
        float v = 0.f;
        if (kch >= 0 && mch >= 0) v = e.transposed ? e.w[((long long)kch * e.dim1 + mch) * 9 + (8 - t)] : e.w[((long long)mch * e.dim1 + kch) * 9 + t];
        tile[mr * PITCH + kc * 9 + t] = v;
    }
    __syncthreads();
    for (int v = threadIdx.x; v < 9 * 64; v += 256) {
        const int t = v >> 6, lane = v & 63;
        const float* src = tile + (lane & 31) * PITCH + (lane >> 5) * 72 + t;
        uint32_t hi[8], lo[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float x = src[c * 9];
            if (e.f16) { hi[c] = f2h(x * e.scale); lo[c] = f2h(x * e.scale - h2f(hi[c])); }
            else split_bf16(x * e.scale, hi[c], lo[c]);
        }
        uint4* o = e.out + ((size_t)((cp * 9 + t) * e.mtiles + mt) * e.npl) * 64 + lane;
        o[0] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
        if (e.npl == 2) o[64] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
    }
}

struct TileCfg { int TH, TW, P, NPIX_T, NPIX_L, tiles_x, tiles_y; size_t lds; };

// Choose (TH, TW): minimise the number of workgroup tiles (every wave always runs R column tiles per tile) plus a small
// halo-traffic term, under the LDS budget that keeps `nwg` workgroups resident per CU.
TileCfg pick_tile_search(int H, int W, int npl, int mt, int nwg);
// the search walks up to W column splits: remember the last few geometries per thread (a forward pass repeats three or four of them 351 times)
TileCfg pick_tile(int H, int W, int npl, int mt, int nwg) {
    struct Entry { int H, W, npl, mt, nwg; TileCfg cfg; };
    static thread_local Entry cache[8];
    static thread_local int next = 0;
    for (const Entry& e : cache)
        if (e.H == H && e.W == W && e.npl == npl && e.mt == mt && e.nwg == nwg && e.cfg.P) return e.cfg;
    Entry& e = cache[next];
    next = (next + 1) % 8;
    e = Entry{H, W, npl, mt, nwg, pick_tile_search(H, W, npl, mt, nwg)};
    return e.cfg;
}
TileCfg pick_tile_search(int H, int W, int npl, int mt, int nwg) {
    const int R = r_of(mt), MAXS = maxs_of(mt);
    TileCfg best{};
    double best_cost = -1;
    const size_t budget = 160 * 1024;
    const int max_px = 32 * NW * R;
    for (int ntx = 1; ntx <= W; ++ntx) {
        const int TW = (W + ntx - 1) / ntx;
        const int P = TW + 2;
        if (P > max_px) continue;
        if (ntx > 1 && TW < 6) break;
        int THmax = max_px / P;
        if (THmax > H) THmax = H;
        for (int TH = THmax; TH >= 1 && TH >= THmax - 6; --TH) {
            const int npix_t = (TH + 2) * P;
            int npix_l = max_px + 2 * P + 2;
            if (npix_l < npix_t) npix_l = npix_t;
            if (npix_l > MAXS * NW * 64) continue;
            const size_t lds = nwg * ((size_t)2 * npl * npix_l * 16 + (size_t)9 * mt * npl * 1024);
            if (lds > budget) continue;
            const int nty = (H + TH - 1) / TH;
            const double halo = (double)(TH + 2) * P / ((double)TH * TW);
            const double cost = (double)ntx * nty * (1.0 + 0.05 * halo);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best = TileCfg{TH, TW, P, npix_t, npix_l, ntx, nty, lds};
            }
        }
    }
    return best;
}

#ifdef ESR_TRACE
unsigned long long* g_trace = nullptr;
#endif

// 64 zero floats per device: what ConvArgs.bias points at when the launch has no bias (the kernel seeds its accumulators without a branch)
__device__ float g_zero_bias[64];
const float* zero_bias() {
    static const float* ptr[64] = {};      // per device (benign race: every thread resolves the same address)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const float*& p = ptr[dev & 63];
    if (!p) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero_bias)) != hipSuccess) return nullptr;
        p = (const float*)q;
    }
    return p;
}

// Which copies of a chunk each wave issues (DmaShare): a chunk is 2 * npl activation planes x `nslots` 1-KiB slots plus `nwi` 1-KiB weight
// fragments; wave w owns the slots w, w + NW, ... and a contiguous range of weight fragments sized so that every wave issues the same
// number of copies (+-1).  Packed per wave: slots | first fragment << 8 | fragments << 16.
void dma_share_host(int npl, int nwi, int nslots, unsigned (&out)[NW]) {
    const int target = (2 * npl * nslots + nwi + NW - 1) / NW;
    int start = 0;
    for (int v = 0; v < NW; ++v) {
        const int nv = nslots > v ? (nslots - v + NW - 1) / NW : 0;
        int c = target - 2 * npl * nv;
        c = c < 0 ? 0 : c;
        if (c > nwi - start || v == NW - 1) c = nwi - start;
        out[v] = (unsigned)nv | ((unsigned)start << 8) | ((unsigned)c << 16);
        start += c;
    }
}

template <int NPL, int MT, int EPI, int NST, int FMT, int NPW, bool PARTLO, int TMODE = 0>
int launch_nst(const ConvArgs& a, hipStream_t s) {
    void (*k)(const ConvArgs) = conv3x3_tile_kernel<NPL, MT, EPI, NST, FMT, NPW, PARTLO, TMODE>;
    ESR_ALLOW_160K_LDS(k);
    const int nslices = a.wslice ? a.nslices : 1;
    const size_t stage = (size_t)2 * NPL * a.NPIX_L * 16 + (size_t)9 * MT * NPW * 1024;
    const size_t lds = (NST == 1 ? 1 : (NST == 4 ? 4 : 2)) * stage;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(k, dim3(a.ntiles, nslices, a.ksplit > 1 ? a.ksplit : 1), dim3(NTHREADS), lds, s, a);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

template <int NPL, int MT, int EPI, int FMT, int NPW, bool PARTLO, int TMODE = 0>
int launch(const ConvArgs& a, hipStream_t s) {
    // no more tiles than CUs (+25 %): every workgroup is alone on its CU, so it pipelines its own DMA (two stages fit: the tile
    // geometry is chosen for two resident single-stage workgroups)
    const int ntiles = a.tiles_x * a.tiles_y * a.B * (a.wslice ? a.nslices : 1) * (a.ksplit > 1 ? a.ksplit : 1);
    const int force = a.stages_hint;                 // esr_conv3x3_desc.lds_stages: 0 = by launch size, 1 / 2 = that form
    const bool small = ntiles <= 320;
    // few small tiles, long K: the four-stage ring where it fits (plain bf16 kernels — what the critic's deep layers launch)
    if constexpr ((EPI == 0 || EPI == EPI_NCHW) && !PARTLO && FMT == 0 && MT == 2) {
        const size_t stage = (size_t)2 * NPL * a.NPIX_L * 16 + (size_t)9 * MT * NPW * 1024;
        if (!force && small && a.ncp >= (EPI == EPI_NCHW ? 8 : 16) && 4 * stage <= 160 * 1024) return launch_nst<NPL, MT, EPI, 4, FMT, NPW, PARTLO, TMODE>(a, s);
    }
    // the tap-masked kernels with hi+lo operands (four / two copies of the chunk body with their own tap sets) do not fit the 256 registers of
    // the two-workgroups-per-CU form — they spilled 34-168 VGPRs to scratch: always the two-stage form (one workgroup per CU, 512 registers)
    if constexpr (TMODE != 0 && NPL == 2) return launch_nst<NPL, MT, EPI, 2, FMT, NPW, PARTLO, TMODE>(a, s);
    else {
        const bool two = force ? force == 2 : small;
        return two ? launch_nst<NPL, MT, EPI, 2, FMT, NPW, PARTLO, TMODE>(a, s) : launch_nst<NPL, MT, EPI, 1, FMT, NPW, PARTLO, TMODE>(a, s);
    }
}

// the epilogue combinations the RRDB forward / backward plans use
template <int NPL, int MT, int FMT, int NPW, bool PARTLO = false>
int launch_epi(const ConvArgs& a, int epi, hipStream_t s) {
    if (FMT == 1) {              // f16: the inference forward and the data gradient of 'mixed' (EPI_MASK)
        switch (epi) {
            case EPI_MASK: return launch<NPL, MT, EPI_MASK, FMT, NPW, PARTLO>(a, s);
            case 0: return launch<NPL, MT, 0, FMT, NPW, PARTLO>(a, s);
            case EPI_RES1: return launch<NPL, MT, EPI_RES1, FMT, NPW, PARTLO>(a, s);
            case EPI_RES1 | EPI_RES2: return launch<NPL, MT, EPI_RES1 | EPI_RES2, FMT, NPW, PARTLO>(a, s);
            case EPI_RESIN: return launch<NPL, MT, EPI_RESIN, FMT, NPW, PARTLO>(a, s);
            case EPI_RESIN | EPI_RES2: return launch<NPL, MT, EPI_RESIN | EPI_RES2, FMT, NPW, PARTLO>(a, s);
            case EPI_NCHW: return launch<NPL, MT, EPI_NCHW, FMT, NPW, PARTLO>(a, s);
            case EPI_OUT2: return launch<NPL, MT, EPI_OUT2, FMT, NPW, PARTLO>(a, s);
            case EPI_PS: return launch<NPL, MT, EPI_PS, FMT, NPW, PARTLO>(a, s);
            default: return ESR_E_UNSUPPORTED;
        }
    }
    switch (epi) {
        case 0: return launch<NPL, MT, 0, FMT, NPW, PARTLO>(a, s);
        case EPI_RES1: return launch<NPL, MT, EPI_RES1, FMT, NPW, PARTLO>(a, s);
        case EPI_RES1 | EPI_RES2: return launch<NPL, MT, EPI_RES1 | EPI_RES2, FMT, NPW, PARTLO>(a, s);
        case EPI_RESIN: return launch<NPL, MT, EPI_RESIN, FMT, NPW, PARTLO>(a, s);
        case EPI_RESIN | EPI_RES2: return launch<NPL, MT, EPI_RESIN | EPI_RES2, FMT, NPW, PARTLO>(a, s);
        case EPI_NCHW: return launch<NPL, MT, EPI_NCHW, FMT, NPW, PARTLO>(a, s);
        case EPI_OUT2: return launch<NPL, MT, EPI_OUT2, FMT, NPW, PARTLO>(a, s);
        case EPI_RES1 | EPI_MASK: return launch<NPL, MT, EPI_RES1 | EPI_MASK, FMT, NPW, PARTLO>(a, s);
        case EPI_MASK: return launch<NPL, MT, EPI_MASK, FMT, NPW, PARTLO>(a, s);
        case EPI_PS: return launch<NPL, MT, EPI_PS, FMT, NPW, PARTLO>(a, s);
        default: return ESR_E_UNSUPPORTED;
    }
}

// split K, second launch: out = sum of the S fp32 partial slabs (fixed order: the result does not depend on scheduling), stored as bf16 hi [+ lo]
// at the interior pixels like the conv epilogue does.  One thread per (b, group, y, x): 8 channels x S coalesced reads, one 16-byte store per plane.
__global__ void splitk_finish_kernel(const float* __restrict__ ws, int S, long long slab, DView out, int C, int H, int W, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % W);
    long long t = idx / W;
    const int y = (int)(t % H);
    t /= H;
    const int ncg = C >> 3;
    const int cg = (int)(t % ncg);
    const int b = (int)(t / ncg);
    uint32_t vh[8], vl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float* p = ws + ((long long)(b * C + cg * 8 + e) * H + y) * W + x;
        float v = 0.f;
        for (int k = 0; k < S; ++k) v += p[k * slab];
        split_bf16(v, vh[e], vl[e]);
    }
    const long long o = b * out.bs + cg * out.cs + (long long)(y + 1) * (W + 2) + (x + 1);
    ((uint4*)out.hi)[o] = make_uint4(vh[0] | (vh[1] << 16), vh[2] | (vh[3] << 16), vh[4] | (vh[5] << 16), vh[6] | (vh[7] << 16));
    if (out.lo) ((uint4*)out.lo)[o] = make_uint4(vl[0] | (vl[1] << 16), vl[2] | (vl[3] << 16), vl[4] | (vl[5] << 16), vl[6] | (vl[7] << 16));
}

}  // namespace

#ifdef ESR_TRACE
extern "C" void esr_debug_trace(void* buf) { g_trace = (unsigned long long*)buf; }
#endif

extern "C" size_t esr_conv_wpack_bytes(int ncg_in, int cout, int split) {
    const int ncp = (ncg_in + 1) / 2, mt = (cout + 31) / 32;
    return (size_t)ncp * 9 * mt * ((split == 1 || split == 3) ? 2 : 1) * 64 * 16;
}

extern "C" int esr_pack_conv_weights(const float* w, int cout_w, int cin_w, const int32_t* kmap, int ncg_in, const int32_t* mmap,
                                     int mtiles, int transposed, int split, float scale, void* wpack, esr_stream_t stream) {
    if (!w || !kmap || !mmap || !wpack || ncg_in <= 0 || mtiles <= 0) return ESR_E_ARG;
    const int ncp = (ncg_in + 1) / 2;
    const int total = ncp * 9 * mtiles * 64;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, cout_w, cin_w, kmap,
                       ncg_in, mmap, mtiles, transposed, (split == 1 || split == 3) ? 2 : 1, split >= 2 ? 1 : 0, scale, (uint4*)wpack, total);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

static int64_t pack_batch_blocks(const esr_pack_desc* descs, int n) {
    int64_t nb = 0;
    for (int i = 0; i < n; ++i) {
        if (!descs[i].w || !descs[i].kmap || !descs[i].mmap || !descs[i].wpack || descs[i].ncg_in <= 0 || descs[i].mtiles <= 0) return ESR_E_ARG;
        nb += (int64_t)((descs[i].ncg_in + 1) / 2) * descs[i].mtiles;             // one block per (K chunk, M tile)
    }
    return nb;
}

extern "C" int64_t esr_pack_batch_workspace_bytes(const esr_pack_desc* descs, int n) {
    if (!descs || n <= 0) return ESR_E_ARG;
    const int64_t nb = pack_batch_blocks(descs, n);
    if (nb < 0) return nb;
    return ((int64_t)n * sizeof(PackEntry) + 255) / 256 * 256 + nb * (int64_t)sizeof(int2);
}

extern "C" int64_t esr_pack_batch_upload(const esr_pack_desc* descs, int n, void* workspace, int64_t workspace_bytes, esr_stream_t stream) {
    if (!descs || n <= 0 || !workspace) return ESR_E_ARG;
    const int64_t need = esr_pack_batch_workspace_bytes(descs, n);
    if (need < 0 || workspace_bytes < need) return ESR_E_ARG;
    const int64_t nb = pack_batch_blocks(descs, n);
    std::vector<PackEntry> table(n);
    std::vector<int2> map((size_t)nb);
    int64_t b = 0;
    for (int i = 0; i < n; ++i) {
        PackEntry& e = table[i];
        e.w = descs[i].w; e.kmap = descs[i].kmap; e.mmap = descs[i].mmap; e.out = (uint4*)descs[i].wpack;
        e.dim0 = descs[i].cout_w; e.dim1 = descs[i].cin_w; e.ncg_in = descs[i].ncg_in; e.mtiles = descs[i].mtiles;
        e.transposed = descs[i].transposed; e.npl = (descs[i].split == 1 || descs[i].split == 3) ? 2 : 1; e.f16 = descs[i].split >= 2 ? 1 : 0; e.scale = descs[i].scale;
        e.total = ((descs[i].ncg_in + 1) / 2) * 9 * descs[i].mtiles * 64;
        for (int j = 0; j < ((descs[i].ncg_in + 1) / 2) * descs[i].mtiles; ++j) map[(size_t)b++] = make_int2(i, j);
    }
    const size_t tb = ((size_t)n * sizeof(PackEntry) + 255) / 256 * 256;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(workspace, table.data(), (size_t)n * sizeof(PackEntry), hipMemcpyHostToDevice, s) != hipSuccess) return ESR_E_LAUNCH;
    if (hipMemcpyAsync((char*)workspace + tb, map.data(), (size_t)nb * sizeof(int2), hipMemcpyHostToDevice, s) != hipSuccess) return ESR_E_LAUNCH;
    // (pageable host memory: the runtime stages both copies before hipMemcpyAsync returns, so the vectors may go out of scope; the stream is
    // not synchronised)
    return nb;
}

extern "C" int esr_pack_batch_run(const void* workspace, int n, int64_t nblocks, esr_stream_t stream) {
    if (!workspace || n <= 0 || nblocks <= 0) return ESR_E_ARG;
    const size_t tb = ((size_t)n * sizeof(PackEntry) + 255) / 256 * 256;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const PackEntry*)workspace,
                       (const int2*)((const char*)workspace + tb));
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

extern "C" int esr_conv3x3(const esr_conv3x3_desc* d, esr_stream_t stream) {
    if (!d || !d->in1.hi || !d->wpack || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->cout <= 0) return ESR_E_ARG;
    if (!d->out.hi && !d->out_nchw) return ESR_E_ARG;
    const int ups = d->upsample <= 0 ? 1 : d->upsample;
    if (ups > 1 && d->in0.hi) return ESR_E_UNSUPPORTED;
    if (d->in1.H * ups != d->H || d->in1.W * ups != d->W) return ESR_E_ARG;
    if (d->in0.hi && (d->in0.H != d->H || d->in0.W != d->W)) return ESR_E_ARG;
    const bool split = d->in1.lo != nullptr;
    const bool f16 = d->in1.fmt == ESR_FMT_F16;
    // one element format per launch
    if ((d->in0.hi && d->in0.fmt != d->in1.fmt) || (d->out.hi && d->out.fmt != d->in1.fmt) || (d->out2.hi && d->out2.fmt != d->in1.fmt) ||
        (d->res1.hi && d->res1.fmt != d->in1.fmt) || (d->res2.hi && d->res2.fmt != d->in1.fmt))
        return ESR_E_ARG;      // (mask_src may be in either format: only its sign and zero-ness are read, and those bits coincide)
    if (d->in0.hi && ((d->in0.lo != nullptr) != split)) return ESR_E_ARG;
    // cout > 64: output slices of 64 channels in ONE launch (grid y); the caller packs the weights of slice s (rows 64 s .. 64 s + 63) as a
    // 64-row pack at byte offset s * esr_conv_wpack_bytes(groups, 64, fmt) of `wpack`; bias, out, out2, res1, res2 and mask_src are
    // indexed by absolute output channel
    const int nslices = d->cout > 64 ? d->cout / 64 : 1;
    if (d->cout > 64 && (d->cout % 64 || d->out_nchw || d->pixel_shuffle > 1 || !d->out.hi)) return ESR_E_UNSUPPORTED;
    const int mt = nslices > 1 ? 2 : (d->cout + 31) / 32;
    const int ps = d->pixel_shuffle > 1 ? d->pixel_shuffle : 0;
    if (ps) {
        if (!d->out.hi || d->out_nchw || d->out2.hi || d->res1.hi || d->res2.hi || d->mask_src.hi || d->cout % 8) return ESR_E_UNSUPPORTED;
        if (d->out.H != ps * d->H || d->out.W != ps * d->W || (d->ps_rowgroup0 + d->cout / 8 + ps * ps - 1) / (ps * ps) > d->out.ncg) return ESR_E_ARG;
    } else if (d->out.hi && (d->out.H != d->H || d->out.W != d->W)) return ESR_E_ARG;
    if (!ps && d->out.hi && d->out.ncg * 8 < d->cout) return ESR_E_ARG;
    // a missing lo OUTPUT plane with hi+lo inputs is the single-plane-intermediate case (fp16 formats only, checked below)
    if (d->out.hi && d->out.lo && !split) return ESR_E_ARG;
    if (d->out2.hi && (!d->out.hi || (d->out2.lo && !d->out.lo))) return ESR_E_ARG;      // out2 may drop the lo plane, not add one
    if (d->act_slope <= 0.f || d->act_slope > 1.f || !(d->alpha >= 0.f)) return ESR_E_ARG;      // (the epilogue evaluates alpha * LeakyReLU as a max of two products)

    ConvArgs a{};
    a.in0 = to_dview(d->in0);
    a.in1 = to_dview(d->in1);
    a.ups = ups;
    a.Win_p = d->in1.W + 2;
    a.wpack = (const uint4*)d->wpack;
    a.zero_bias = zero_bias();
    if (!a.zero_bias) return ESR_E_LAUNCH;
    a.bias = d->bias ? d->bias : a.zero_bias;
    a.bias_stride = d->bias ? 1 : 0;
    a.cout = nslices > 1 ? 64 : d->cout;
    a.nslices = nslices;
    a.B = d->B;
    a.H = d->H;
    a.W = d->W;
    const int npl = split ? 2 : 1;
    // A 64-channel layer whose launch would leave every workgroup alone on its CU (no more tiles than CUs: the 52 x 52 training crops, a single
    // image of the Z search) runs as TWO 32-channel output slices of the MT = 1 kernel instead (grid y; same pack, ConvArgs.wtap): twice the
    // workgroups, so that two are resident per CU and cover each other's copy issue and waits — at the price of staging the input tile twice
    // (the second copy comes out of the same XCD's L2).  Same arithmetic per output channel: results are bit-identical to the 64-channel form.
    bool mslice = false;
    auto all_taps = [](const int32_t (&m)[4]) { return (m[0] == 0 || m[0] == 0x1FF) && (m[1] == 0 || m[1] == 0x1FF) && (m[2] == 0 || m[2] == 0x1FF) && (m[3] == 0 || m[3] == 0x1FF); };
    if (nslices == 1 && mt == 2 && d->cout == 64 && !d->out_nchw && !ps && d->lds_stages == 0 && !d->k_split_ws && all_taps(d->tap_mask_k) &&
        all_taps(d->tap_mask_m) && (!d->mask_src.hi || (d->mask_cg0 == 0 && d->mask_cg1 >= 8))) {
        const TileCfg t2 = pick_tile(d->H, d->W, npl, 2, WGS_MT2);
        mslice = t2.TH != 0 && (long long)t2.tiles_x * t2.tiles_y * d->B <= 320;
    }
    const int mt_k = mslice ? 1 : mt;                              // M tiles per workgroup of the kernel that runs
    const int wgs_per_cu = mt_k == 1 ? WGS_MT1 : WGS_MT2;
    const TileCfg t = pick_tile(d->H, d->W, npl, mt_k, wgs_per_cu);
    if (t.TH == 0) return ESR_E_UNSUPPORTED;
    if (mslice) { a.cout = 32; a.nslices = 2; }
    a.TH = t.TH; a.TW = t.TW; a.P = t.P; a.NPIX_T = t.NPIX_T; a.NPIX_L = t.NPIX_L;
    a.tiles_x = t.tiles_x; a.tiles_y = t.tiles_y;
    // the kernel's divisions by launch constants, as multiplications (ConvArgs; conv3x3_tile_kernel's prologue)
    a.ntiles = a.tiles_x * a.tiles_y * a.B;
    a.xcd_q = a.ntiles / 8;
    a.xcd_r = a.ntiles % 8;
    a.m_tx = a.tiles_x == 1 ? 0 : (unsigned)((0x100000000ull + a.tiles_x - 1) / a.tiles_x);     // (divisor 1: 2^32 does not fit — udiv_magic adds n * i instead)
    a.m_ty = a.tiles_y == 1 ? 0 : (unsigned)((0x100000000ull + a.tiles_y - 1) / a.tiles_y);
    a.i_tx = a.tiles_x == 1;
    a.i_ty = a.tiles_y == 1;
    a.m_P = ((1u << 20) + a.P - 1) / a.P;
    a.m_ups = ((1u << 16) + ups - 1) / ups;
    a.nslots = (a.NPIX_L + 63) / 64;
    a.ncg_out = (a.cout + 7) >> 3;
    // (udiv_magic is exact while n * d < 2^32: tile / tiles_x with tile < ntiles, and trow / tiles_y with trow < B * tiles_y)
    if (d->H + 2 >= 32768 || d->W + 2 >= 32768 || ups > 8 || (long long)a.ntiles * a.tiles_x >= 0x100000000ll ||
        (long long)a.B * a.tiles_y * a.tiles_y >= 0x100000000ll)
        return ESR_E_UNSUPPORTED;
    {
        // every per-lane address of the kernel is a uniform per-image base + a 32-bit byte offset
        auto fits = [](const esr_act_view& v) { return !v.hi || (long long)v.ncg * v.cg_stride * 16 < 0x100000000ll; };
        if (!fits(d->in0) || !fits(d->in1) || !fits(d->out) || !fits(d->out2) || !fits(d->res1) || !fits(d->res2) || !fits(d->mask_src)) return ESR_E_UNSUPPORTED;
        if (d->out_nchw && (long long)d->cout * d->H * d->W * 4 >= 0x100000000ll) return ESR_E_UNSUPPORTED;
        // residuals cover every output group (the epilogue reads them for all of them)
        const int ncg_all = (d->cout + 7) >> 3;
        if ((d->res1.hi && d->res1.ncg < ncg_all) || (d->res2.hi && d->res2.ncg < ncg_all)) return ESR_E_ARG;
        if (d->mask_src.hi && d->mask_src.ncg < (d->mask_cg1 < ncg_all ? d->mask_cg1 : ncg_all) - d->mask_cg0) return ESR_E_ARG;
    }
    a.ncp = (a.in0.ncg + a.in1.ncg + 1) / 2;
    a.act_slope = d->act_slope;
    a.alpha = d->alpha;
    a.beta1 = d->beta1;
    a.beta2 = d->beta2;
    a.res1 = to_dview(d->res1);
    a.res2 = to_dview(d->res2);
    a.out = to_dview(d->out);
    a.out2 = to_dview(d->out2);
    a.mask = to_dview(d->mask_src);
    a.out_nchw = d->out_nchw;
    a.nchw_ctot = d->cout;
    a.mask_cg0 = d->mask_cg0;
    a.mask_cg1 = d->mask_cg1;
    a.mask_slope = d->mask_slope;
    a.reverse = d->reverse_order;
    a.stages_hint = (d->lds_stages == 1 || d->lds_stages == 2) ? d->lds_stages : 0;
    a.range_flag = f16 ? d->range_flag : nullptr;
    a.range_tag = d->range_tag;
    a.ps = ps;
    a.ps_rg0 = d->ps_rowgroup0;
#ifdef ESR_TRACE
    a.trace = g_trace;
#endif
    int epi = 0;
    if (d->res1.hi) epi |= EPI_RES1;
    // residual 1 == a channel-group slice of this conv's own main input, linear epilogue: take it from the staged LDS tile
    if (nslices == 1 && d->res1.hi && d->act_slope == 1.f && d->alpha != 0.f && ups == 1 && !d->mask_src.hi && !d->out_nchw &&
        d->res1.batch_stride == d->in1.batch_stride && d->res1.cg_stride == d->in1.cg_stride && ((d->res1.lo != nullptr) == split) &&
        d->in1_lo_groups >= 0) {
        const long long unit = (long long)d->in1.cg_stride * 16;
        const long long off = (const char*)d->res1.hi - (const char*)d->in1.hi;
        const bool lo_ok = !split || ((const char*)d->res1.lo - (const char*)d->in1.lo) == off;
        if (lo_ok && off >= 0 && off % unit == 0 && off / unit + (d->cout + 7) / 8 <= d->in1.ncg) {
            a.resin_g0 = a.in0.ncg + (int)(off / unit);
            a.resin_scale = d->beta1 / d->alpha;
            epi = (epi & ~EPI_RES1) | EPI_RESIN;
        }
    }
    if (d->res2.hi) epi |= EPI_RES2;
    if (d->mask_src.hi) epi |= EPI_MASK;
    if (d->out_nchw) epi |= EPI_NCHW;
    if (d->out2.hi) epi |= EPI_OUT2;
    if (ps) epi |= EPI_PS;
    if ((epi & EPI_NCHW) && d->out.hi) return ESR_E_UNSUPPORTED;     // one destination kind per launch
    hipStream_t s = (hipStream_t)stream;
    int wpl = d->weight_planes;
    if (wpl == 0) wpl = f16 ? 1 : npl;
    if (wpl < 1 || wpl > npl || (!f16 && wpl != npl)) return ESR_E_ARG;
    a.wslice = nslices > 1 ? (long long)a.ncp * 9 * 2 * wpl * 64 : 0;
    a.wchunk = (long long)9 * mt * wpl * 64;
    a.wtap = mt * wpl;
    if (mslice) a.wslice = (long long)wpl * 64;                   // slice s = M tile s of every (chunk, tap) block of the 64-row pack
    dma_share_host(npl, 9 * mt_k * wpl, a.nslots, a.share);
    // which leading chunks of the concatenated input carry a lo plane
    a.lo_chunks = a.ncp;
    bool partlo = false;
    if (split && d->in1_lo_groups > 0 && d->in1_lo_groups < d->in1.ncg) {
        a.lo_chunks = (a.in0.ncg + d->in1_lo_groups + 1) / 2;
        partlo = true;
    }
    if (split && d->in1_lo_groups < 0) {                             // the conv reads hi planes only (lo planes exist but are not operands)
        a.lo_chunks = 0;
        partlo = true;
    }
    if (split && d->out.hi && !d->out.lo) partlo = true;
    if (partlo && !f16) return ESR_E_UNSUPPORTED;                   // single-plane intermediates exist for the fp16 formats only
    // tap masks are a HINT (blocks outside them must be zero in the pack): the two patterns with compiled kernels are honoured, anything
    // else multiplies all nine taps — the same result
    auto all9 = [](const int32_t (&m)[4]) { return (m[0] == 0 || m[0] == 0x1FF) && (m[1] == 0 || m[1] == 0x1FF) && (m[2] == 0 || m[2] == 0x1FF) && (m[3] == 0 || m[3] == 0x1FF); };
    auto same = [](const int32_t (&m)[4], const int (&p)[4]) { return m[0] == p[0] && m[1] == p[1] && m[2] == p[2] && m[3] == p[3]; };
    const bool plain = epi == 0 && !f16 && !partlo && mt == 2 && a.in0.ncg == 0;
    const bool tm1 = plain && same(d->tap_mask_k, S2D_FWD) && d->tap_mask_k_shift == 1 && all9(d->tap_mask_m) && (a.in1.ncg % 4) == 0;
    const bool tm2 = plain && !tm1 && same(d->tap_mask_m, S2D_FLIP) && all9(d->tap_mask_k);
    // split K (k_split_ws): launches of few workgroups with a long K axis — the critic's 256- / 512-channel layers on 16x16 ... 4x4 maps are 24-100
    // workgroups walking 32-128 chunks each on a 256-CU part.  S = 8, 4 or 2 sets of workgroups (grid z) contract 1/S of the input channels each
    // into their own fp32 slab; a second launch adds the slabs in order.  Chosen here, from the tiling: the caller only lends the workspace.
    if (plain && d->k_split_ws && d->act_slope == 1.f && ups == 1 && d->cout % 64 == 0) {
        const long long slab = (long long)d->B * d->cout * d->H * d->W;
        const long long wgs = (long long)a.tiles_x * a.tiles_y * a.B * nslices;
        int S = 1;
        for (int c = 8; c >= 2; c /= 2) {
            if (a.ncp % c || a.ncp / c < 8 || (tm1 && (a.ncp / c) % 8)) continue;
            if (wgs * c > 320 || slab * c > d->k_split_ws_floats) continue;
            S = c;
            break;
        }
        if (S > 1) {
            a.ksplit = S;
            a.ncp /= S;                      // (a.wslice above is the stride of the FULL pack)
            a.kz_groups = 2 * a.ncp;
            a.in1.ncg = a.kz_groups;
            a.kz_slab = slab;
            a.out_nchw = d->k_split_ws;
            int rc;
            if (tm1) rc = split ? launch<2, 2, EPI_NCHW, 0, 2, false, 1>(a, s) : launch<1, 2, EPI_NCHW, 0, 1, false, 1>(a, s);
            else if (tm2) rc = split ? launch<2, 2, EPI_NCHW, 0, 2, false, 2>(a, s) : launch<1, 2, EPI_NCHW, 0, 1, false, 2>(a, s);
            else rc = split ? launch<2, 2, EPI_NCHW, 0, 2, false>(a, s) : launch<1, 2, EPI_NCHW, 0, 1, false>(a, s);
            if (rc != ESR_OK) return rc;
            const long long total = (long long)d->B * (d->cout / 8) * d->H * d->W;
            ESR_CLEAR_ERR();
            hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)d->k_split_ws, S, slab, a.out,
                               d->cout, d->H, d->W, total);
            ESR_CHECK_LAUNCH();
            return ESR_OK;
        }
    }
    if (tm1)
        return split ? launch<2, 2, 0, 0, 2, false, 1>(a, s) : launch<1, 2, 0, 0, 1, false, 1>(a, s);
    if (tm2)
        return split ? launch<2, 2, 0, 0, 2, false, 2>(a, s) : launch<1, 2, 0, 0, 1, false, 2>(a, s);
    if (f16 && split && partlo && wpl == 2) return mt_k == 1 ? launch_epi<2, 1, 1, 2, true>(a, epi, s) : launch_epi<2, 2, 1, 2, true>(a, epi, s);
    if (f16 && split && partlo) return mt_k == 1 ? launch_epi<2, 1, 1, 1, true>(a, epi, s) : launch_epi<2, 2, 1, 1, true>(a, epi, s);
    if (f16 && split && wpl == 2) return mt_k == 1 ? launch_epi<2, 1, 1, 2>(a, epi, s) : launch_epi<2, 2, 1, 2>(a, epi, s);
    if (f16 && split) return mt_k == 1 ? launch_epi<2, 1, 1, 1>(a, epi, s) : launch_epi<2, 2, 1, 1>(a, epi, s);
    if (f16) return mt_k == 1 ? launch_epi<1, 1, 1, 1>(a, epi, s) : launch_epi<1, 2, 1, 1>(a, epi, s);
    if (split) return mt_k == 1 ? launch_epi<2, 1, 0, 2>(a, epi, s) : launch_epi<2, 2, 0, 2>(a, epi, s);
    return mt_k == 1 ? launch_epi<1, 1, 0, 1>(a, epi, s) : launch_epi<1, 2, 0, 1>(a, epi, s);
}
