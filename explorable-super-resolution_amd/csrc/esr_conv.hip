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
#include "esr_conv_dev.h"

namespace {

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
// NTILE > 1 (single-stage kernels of LARGE launches with one MFMA per product, round 6): a workgroup runs up to NTILE consecutive tiles of its
// XCD's sweep — kernel arguments, slice shifts and wave shares read once, the next tile's first chunk of copies issued in front of the current
// tile's epilogue (the K loop's closing barrier has freed the stage), so its landing time hides behind the stores.  Same tiles, same
// arithmetic per tile: bit-identical to NTILE = 1.
template <int NPL, int MT, int EPI, int NST, int FMT, int NPW, bool PARTLO, int TMODE = 0, int NTILE = 1>
__global__ __launch_bounds__(NTHREADS, NST >= 2 ? 1 : (MT == 1 ? WGS_MT1 : WGS_MT2)) void conv3x3_tile_kernel(const ConvArgs a_in) {
    static_assert(NTILE == 1 || NST == 1, "several tiles per workgroup: the single-stage form only");
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
    const unsigned xcd_first = xcd * a.xcd_q + (xcd < (unsigned)a.xcd_r ? xcd : (unsigned)a.xcd_r);      // first tile of this XCD's sweep
    unsigned tile_f = xcd_first + (blockIdx.x >> 3), ntl = 1;
    if constexpr (NTILE > 1) {
        // (grid: 8 x ceil(longest sweep / NTILE) workgroups — the last workgroup of a sweep may get fewer tiles, or none)
        const unsigned qx = a.xcd_q + (xcd < (unsigned)a.xcd_r ? 1u : 0u), j0 = (blockIdx.x >> 3) * NTILE;
        if (j0 >= qx) return;
        ntl = qx - j0 < (unsigned)NTILE ? qx - j0 : (unsigned)NTILE;
        tile_f = xcd_first + j0;
    }
    // tile -> image b, tile origin (x0, y0): output interior coords == padded coords of the halo origin
    struct TilePos { int b, x0, y0; };
    auto decode = [&](const unsigned tf) {
        const unsigned tile = a.reverse ? a.ntiles - 1 - tf : tf;
        const unsigned trow = udiv_magic(tile, a.m_tx, a.i_tx);         // = image * tiles_y + tile row
        const int bb = udiv_magic(trow, a.m_ty, a.i_ty);
        return TilePos{bb, (int)(tile - trow * a.tiles_x) * a.TW, (int)(trow - bb * a.tiles_y) * a.TH};
    };
    TilePos tp = decode(tile_f);
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
    FetchState<MAXS> fs = setup_tile<MAXS>(a, tp.x0, tp.y0, wave, lane);
    ESR_TR();                                    // (slot 3) tile decoded, slot offsets formed
    static_assert(TMODE != 2 || MT == 2, "M-tile tap masks come in pairs");
    const unsigned char* const sb0 = smem + (lane >> 5) * NPL * plane_bytes + (wave * 32 + (lane & 31)) * 16;
    const unsigned char* const sa0 = smem + 2 * NPL * plane_bytes + lane * 16;
    constexpr int NTERM_CAP = 3;
    // which 2x2 block of taps chunk c's weights live in (dma_chunk)
    auto tsel_of = [&](const int c) { return TMODE == 1 ? ((c >> 1) & 3) : (TMODE == 2 ? (int)(blockIdx.y & 1) : 0); };
    auto issue_for = [&](const FetchState<MAXS>& f, const int bb, const int c, const unsigned stage, const bool xlo) {
        const Bases<NPL> bs = make_bases<NPL, MT, NPW>(a, c, bb);
        dma_chunk<NPL, MT, NPW, TMODE>(f, bs, share, stage, plane_bytes, xlo, tsel_of(c), wave);
    };
    auto issue = [&](const int c, const unsigned stage, const bool xlo) { issue_for(fs, tp.b, c, stage, xlo); };
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
    constexpr bool EARLY_COORDS = NST >= 2 || (MT == 1 && NPL == 1) || NTILE > 1;
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
        if constexpr (EARLY_COORDS) ec = epi_coords<R, EKIND>(a, tp.x0, tp.y0, wave, lane_o);
    };
    // The first chunk's copies go out in front of the K loop, the seed right behind them.  (Seeding inside the loop's first pass instead —
    // one copy of the issue code — made the accumulators' loop-carried registers VGPRs: 96 v_accvgpr moves per chunk, +0.35 us per chunk.)
    if (NST != 4) issue(0, lds0, !PARTLO || 0 < a.lo_chunks);
    __builtin_amdgcn_sched_barrier(0);
    ESR_TR();                                    // (slot 4) first chunk's copies issued
    const int lo_end = PARTLO ? (a.lo_chunks < a.ncp ? a.lo_chunks : a.ncp) : a.ncp;
  for (unsigned it = 0;;) {                      // (NTILE == 1: one pass, no loop)
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (NTILE > 1, cp == 0: also the previous tile's stores — issued ahead of these copies' landing)
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
    for (int cp = 0; cp < lo_end; ++cp) step(std::true_type{}, cp);
    if constexpr (PARTLO)
        for (int cp = lo_end; cp < a.ncp; ++cp) step(std::false_type{}, cp);
    ESR_TR();
    if constexpr (NTILE > 1) {
        // the next tile of this workgroup: decoded and its first chunk's copies issued now — every wave is past the K loop's closing barrier, the
        // stage is free — so that they land while the epilogue below stores this tile
        const bool more = it + 1 < ntl;
        TilePos tn = tp;
        FetchState<MAXS> fn = fs;
        if (more) {
            tn = decode(tile_f + it + 1);
            fn = setup_tile<MAXS>(a, tn.x0, tn.y0, wave, lane);
            issue_for(fn, tn.b, 0, lds0, !PARTLO || 0 < a.lo_chunks);
            __builtin_amdgcn_sched_barrier(0);
        }
        conv_epilogue<NPL, MT, R, EPI, FMT, PARTLO>(a, acc, tp.b, ec, lane);
        ESR_TR();
        if (!more) break;
        tp = tn;
        fs = fn;
        ++it;
    } else {
        if constexpr (!EARLY_COORDS) ec = epi_coords<R, EKIND>(a, tp.x0, tp.y0, wave, lane);
        conv_epilogue<NPL, MT, R, EPI, FMT, PARTLO>(a, acc, tp.b, ec, lane);
        ESR_TR();
        break;
    }
  }
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
TileCfg pick_tile_search(int H, int W, int npl, int mt, int nwg, bool full_width);
// the search walks up to W column splits: remember the last few geometries per thread (a forward pass repeats three or four of them 351 times)
// full_width: whole-row tiles only (tiles_x == 1: what the chain kernel's vertical halo recompute needs); TH == 0 when a row does not fit a tile
TileCfg pick_tile(int H, int W, int npl, int mt, int nwg, bool full_width = false) {
    struct Entry { int H, W, npl, mt, nwg; bool fw; TileCfg cfg; };
    static thread_local Entry cache[8];
    static thread_local int next = 0;
    for (const Entry& e : cache)
        if (e.H == H && e.W == W && e.npl == npl && e.mt == mt && e.nwg == nwg && e.fw == full_width && e.cfg.P) return e.cfg;
    const TileCfg cfg = pick_tile_search(H, W, npl, mt, nwg, full_width);
    if (cfg.P) {
        Entry& e = cache[next];
        next = (next + 1) % 8;
        e = Entry{H, W, npl, mt, nwg, full_width, cfg};
    }
    return cfg;
}
TileCfg pick_tile_search(int H, int W, int npl, int mt, int nwg, bool full_width) {
    const int R = r_of(mt), MAXS = maxs_of(mt);
    TileCfg best{};
    double best_cost = -1;
    const size_t budget = 160 * 1024;
    const int max_px = 32 * NW * R;
    for (int ntx = 1; ntx <= (full_width ? 1 : W); ++ntx) {
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

// esr_conv3x3_chain: while set (per thread), esr_conv3x3 goes through all of its validation and launch planning and then hands the argument
// block and the kernel variant it chose to the capture instead of launching
struct ChainCapture { ConvArgs a; ConvVariant v; int captured; };
thread_local ChainCapture* g_capture = nullptr;

template <int NPL, int MT, int EPI, int NST, int FMT, int NPW, bool PARTLO, int TMODE = 0, int NTILE = 1>
int launch_nst(const ConvArgs& a, hipStream_t s) {
    if (g_capture) {
        g_capture->a = a;
        g_capture->v = ConvVariant{NPL, MT, EPI, NST, FMT, NPW, PARTLO ? 1 : 0, TMODE, NTILE};
        g_capture->captured += 1;
        return ESR_OK;
    }
    void (*k)(const ConvArgs) = conv3x3_tile_kernel<NPL, MT, EPI, NST, FMT, NPW, PARTLO, TMODE, NTILE>;
    ESR_ALLOW_160K_LDS(k);
    const int nslices = a.wslice ? a.nslices : 1;
    const size_t stage = (size_t)2 * NPL * a.NPIX_L * 16 + (size_t)9 * MT * NPW * 1024;
    const size_t lds = (NST == 1 ? 1 : (NST == 4 ? 4 : 2)) * stage;
    // NTILE tiles per workgroup: 8 x ceil(longest XCD sweep / NTILE) workgroups (conv3x3_tile_kernel)
    const int grid_x = NTILE == 1 ? a.ntiles : 8 * ((a.xcd_q + (a.xcd_r ? 1 : 0) + NTILE - 1) / NTILE);
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(k, dim3(grid_x, nslices, a.ksplit > 1 ? a.ksplit : 1), dim3(NTHREADS), lds, s, a);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

// Tiles per workgroup of the one-MFMA-per-product single-stage kernels (one-plane weights, 32 output channels) in launches of at least
// NTILE_MIN_TILES tiles: VERDICT r5 item 2 — the per-workgroup set-up is 27 % of a workgroup's life there (profiles/r05_one_mfma_phases.log).
constexpr int NTILE_ONE_MFMA = 2;
constexpr int NTILE_MIN_TILES = 1024;

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
        if constexpr (NTILE_ONE_MFMA > 1 && MT == 1 && NPW == 1 && TMODE == 0 && (EPI & EPI_NCHW) == 0) {
            if (!two && ntiles >= NTILE_MIN_TILES && !a.wslice && a.ksplit <= 1) return launch_nst<NPL, MT, EPI, 1, FMT, NPW, PARTLO, TMODE, NTILE_ONE_MFMA>(a, s);
        }
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
    // (planning a chain — g_capture — asks for whole-row tiles; a row that does not fit one tile ends the plan: chain_prepare reads ESR_E_UNSUPPORTED
    // from a capturing call as "these layers run separately")
    const TileCfg t = pick_tile(d->H, d->W, npl, mt_k, wgs_per_cu, g_capture != nullptr);
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

// ---- a chain of 32-channel layers as one launch (csrc/esr_chain.hip)
namespace {
// byte interval of groups [g0, g0 + ng) of a view's hi plane in image 0
struct Span { const char* lo; const char* hi; long long period; };
Span span_of(const DView& v, int g0, int ng) {
    const char* p = (const char*)v.hi + (long long)g0 * v.cs * 16;
    return Span{p, p + (long long)ng * v.cs * 16, v.bs * 16};
}
// may the two group ranges share a byte in ANY image?  Views of one buffer (same batch stride, bases less than one image apart) repeat with the
// same period, so image 0 decides; anything else is compared over its whole extent across the batch.
bool spans_overlap(const Span& x, const Span& y, int B) {
    if (x.lo == x.hi || y.lo == y.hi) return false;
    const long long d = x.lo - y.lo;
    if (x.period == y.period && d < x.period && -d < x.period) return x.lo < y.hi && y.lo < x.hi;
    return x.lo < y.hi + (long long)(B - 1) * y.period && y.lo < x.hi + (long long)(B - 1) * x.period;
}
}  // namespace

// 1: the layers fuse (c, v0 filled in); 0: they run as separate launches; < 0: a layer's descriptor is invalid (its ESR_E_* code)
static int chain_prepare(const esr_conv3x3_desc* const* descs, int n, ChainArgs& c, ConvVariant& v0) {
    if (!descs || n <= 0) return ESR_E_ARG;
    for (int i = 0; i < n; ++i)
        if (!descs[i]) return ESR_E_ARG;
    auto separately = []() { return 0; };
    const esr_stream_t stream = nullptr;                      // (capturing: nothing is launched)
    if (n < 2 || n > CHAIN_MAX) return separately();
    // only plain 32-channel layers can be chained: one kernel launch each, act-layout destination at the input resolution
    for (int i = 0; i < n; ++i) {
        const esr_conv3x3_desc* d = descs[i];
        if (d->cout > 32 || d->out_nchw || d->pixel_shuffle > 1 || d->upsample > 1 || d->res1.hi || d->res2.hi || d->k_split_ws || d->lds_stages == 1 ||
            !d->out.hi || d->B != descs[0]->B || d->H != descs[0]->H || d->W != descs[0]->W)
            return separately();
    }
    // the launches as esr_conv3x3 would issue them: validation, tiling, kernel variant
    c = ChainArgs{};
    c.n = n;
    v0 = ConvVariant{};
    for (int i = 0; i < n; ++i) {
        ChainCapture cap{};
        g_capture = &cap;
        const int rc = esr_conv3x3(descs[i], stream);
        g_capture = nullptr;
        if (rc == ESR_E_UNSUPPORTED) return separately();          // (no whole-row tile, or a layer the separate launch refuses too: it will say so itself)
        if (rc != ESR_OK) return rc;
        if (cap.captured != 1) return ESR_E_LAUNCH;              // (not reachable for the layers admitted above: they are single launches)
        c.l[i] = cap.a;
        if (i == 0) v0 = cap.v;
        const ConvVariant& v = cap.v;
        if (v.npl != v0.npl || v.mt != v0.mt || v.epi != v0.epi || v.nst != v0.nst || v.fmt != v0.fmt || v.npw != v0.npw || v.partlo != v0.partlo ||
            v.tmode != v0.tmode || v.ntile != v0.ntile)
            return separately();
    }
    const ConvArgs& a0 = c.l[0];
    // the small-launch form only (a lone workgroup per CU, two LDS stages), whole-width tiles (the halo recompute is vertical only)
    if (v0.mt != 1 || v0.nst != 2 || a0.tiles_x != 1 || a0.wslice || a0.ksplit > 1) return separately();
    for (int i = 1; i < n; ++i) {
        const ConvArgs& a = c.l[i];
        if (a.TH != a0.TH || a.TW != a0.TW || a.P != a0.P || a.NPIX_L != a0.NPIX_L || a.nslots != a0.nslots || a.ntiles != a0.ntiles ||
            a.share[0] != a0.share[0] || a.share[1] != a0.share[1] || a.share[2] != a0.share[2] || a.share[3] != a0.share[3] || a.wslice || a.ksplit > 1)
            return separately();
        c.l[i].reverse = a0.reverse;                           // one sweep direction for the launch (a cache hint: results do not depend on it)
    }
    // What the halo recompute relies on.  (1) A layer's output is written by several workgroups at different times: nothing any layer up to and
    // including it reads (inputs, masks) and no other layer's output may alias it.  (2) The first two K chunks (four channel groups) of every
    // layer are not outputs of the chain: they are fetched before the previous layer's stores are known to have landed.
    const int B = a0.B;
    for (int j = 0; j < n; ++j) {
        const ConvArgs& w = c.l[j];
        const Span outs[2] = {span_of(w.out, 0, (w.cout + 7) / 8), w.out2.hi ? span_of(w.out2, 0, (w.cout + 7) / 8) : Span{nullptr, nullptr, 0}};
        for (const Span& o : outs) {
            for (int i = 0; i < n; ++i) {
                const ConvArgs& r = c.l[i];
                if (i <= j && (spans_overlap(o, span_of(r.in0, 0, r.in0.ncg), B) || spans_overlap(o, span_of(r.in1, 0, r.in1.ncg), B))) return separately();
                if (r.mask.hi && spans_overlap(o, span_of(r.mask, 0, r.mask.ncg), B)) return separately();
                if (i != j && (spans_overlap(o, span_of(r.out, 0, (r.cout + 7) / 8), B) ||
                               (r.out2.hi && spans_overlap(o, span_of(r.out2, 0, (r.cout + 7) / 8), B))))
                    return separately();
                if (i > j) {
                    const int g1 = 4 - r.in0.ncg < r.in1.ncg ? 4 - r.in0.ncg : r.in1.ncg;
                    if (spans_overlap(o, span_of(r.in0, 0, r.in0.ncg < 4 ? r.in0.ncg : 4), B) || (g1 > 0 && spans_overlap(o, span_of(r.in1, 0, g1), B))) return separately();
                }
            }
        }
    }
    return esr_internal_chain_launch(&c, (const int*)&v0, nullptr, /*query_only=*/1) == ESR_OK ? 1 : 0;
}

extern "C" int esr_conv3x3_chain_fuses(const esr_conv3x3_desc* const* descs, int n) {
    ChainArgs c;
    ConvVariant v0;
    return chain_prepare(descs, n, c, v0);
}

extern "C" int esr_conv3x3_chain(const esr_conv3x3_desc* const* descs, int n, esr_stream_t stream) {
    ChainArgs c;
    ConvVariant v0;
    const int fuse = chain_prepare(descs, n, c, v0);
    if (fuse < 0) return fuse;
    if (fuse == 1) return esr_internal_chain_launch(&c, (const int*)&v0, (hipStream_t)stream, 0);
    for (int i = 0; i < n; ++i) {
        const int rc = esr_conv3x3(descs[i], stream);
        if (rc != ESR_OK) return rc;
    }
    return ESR_OK;
}

