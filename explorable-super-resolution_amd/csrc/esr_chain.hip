// A CHAIN of 32-channel conv3x3 layers as ONE launch by halo recompute (round 6): the four growing convs of a
// ResidualDenseBlock_5C (reference codes/models/modules/block.py:230-235: x1 = lrelu(conv1(x)), x2 = lrelu(conv2(cat(x, x1))), ...) and the four
// mirrored data-gradient convs of its backward, at the SMALL launch sizes (no more tiles than CUs: the 52 x 52 training crops, single images),
// where a launch is ~12 us of which ~3.3 us are the launch floor and ~4 us prologue + epilogue.
//
//   dependency     layer l+1 reads, through its 3x3 taps, one row of layer l's output above and below its own rows.  A workgroup that owns
//                  output rows [y0, y0 + TH) of the LAST layer therefore computes rows [y0 - e, y0 + TH + e) of layer l, e = n - 1 - l
//                  (clipped to the image, where zero padding replaces the halo): everything it reads of the chain's outputs it has written
//                  itself.  No inter-workgroup flag, no grid barrier — nothing that can hang.
//   identity       every output pixel sees the same MFMA sequence on the same 16-bit operands as in the separate launches (bias seed, chunks in
//                  order, taps in order, terms in order): bit-identical results.  The halo rows a workgroup recomputes are the neighbour's
//                  interior rows: both store the same bits to the same addresses (a benign duplicate; later readers — the closing conv5, the
//                  backward, the weight gradients — see the dense-block buffer exactly as the four launches leave it).
//   visibility     a layer's stores become visible to the workgroup's own later copies by `s_waitcnt vmcnt(0)` + the workgroup barrier in
//                  front of the first dependent chunk: stores and loads of one CU go through the same vector L1 (write-through), which is the
//                  workgroup-scope release / acquire of the gfx9 memory model.  The first two chunks of a layer's K axis are never outputs of
//                  the chain (esr_conv3x3_chain checks), so they are in flight while the previous layer's stores drain.
//   passes         a layer's extent (up to TH + 2e rows) is processed in passes of at most TH rows — the accumulators of one pass are the
//                  tile kernel's (3 column tiles of 32 pixels per wave); a pass with fewer rows runs the K loop instantiated for 2 or 1 column
//                  tiles per wave and copies only the slots it reads.  The next pass's first chunk is issued in front of the current pass's
//                  epilogue.
//   cost           (n = 4, TH = 7) 13 + 11 + 9 + 7 rows instead of 4 x 7: ~1.4 x the K-loop work of the four launches, against three launch
//                  floors, three workgroup dispatches and argument fetches.  Whether that pays is a measurement: profiles/r06_chain_ab.log.
#include "esr_conv_dev.h"

namespace {

// activation copy slots a pass of `th` rows reads: (th + 2) haloed rows of pitch P plus the two vectors the last tap of the last pixel reaches
__device__ __forceinline__ DmaShare pass_share(const DmaShare& full, int th, int P, int nslots, int wave) {
    int ns = ((th + 2) * P + 2 + 63) >> 6;
    ns = ns < nslots ? ns : nslots;
    DmaShare d = full;
    const int mine = ns > wave ? (ns - wave + NW - 1) / NW : 0;
    d.nsl = mine < full.nsl ? mine : full.nsl;
    return d;
}

template <int NPL, int EPI, int FMT, int NPW>
__global__ __launch_bounds__(NTHREADS, 1) void conv3x3_chain_kernel(const ChainArgs c) {
    constexpr int MT = 1, MAXS = maxs_of(MT);
    constexpr bool PARTLO = false;
    static_assert((EPI & ~(EPI_OUT2 | EPI_MASK)) == 0, "chain layers: bias + LeakyReLU (+ second destination / LeakyReLU' mask) epilogues only");
    static_assert(r_of(MT) == 3, "passes are instantiated for 1, 2 and 3 column tiles per wave");
    constexpr int NWI = 9 * MT * NPW;
    constexpr int MAXCNT = 2 * NPL * MAXS + (NWI + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ConvArgs& a0 = c.l[0];                                   // tile space and tile geometry: the same for every layer of the chain
    const int P = a0.P, TH = a0.TH, H = a0.H;
    const int plane_bytes = a0.NPIX_L * 16;
    const int stage_bytes = 2 * NPL * plane_bytes + NWI * 1024;
    const unsigned xcd = blockIdx.x & 7;
    const unsigned tile_f = xcd * a0.xcd_q + (xcd < (unsigned)a0.xcd_r ? xcd : (unsigned)a0.xcd_r) + (blockIdx.x >> 3);
    const unsigned tile = a0.reverse ? a0.ntiles - 1 - tile_f : tile_f;
    const unsigned trow = udiv_magic(tile, a0.m_tx, a0.i_tx);
    const int b = udiv_magic(trow, a0.m_ty, a0.i_ty);
    const int x0 = (tile - trow * a0.tiles_x) * a0.TW, y0 = (trow - b * a0.tiles_y) * TH;      // (tiles_x == 1: x0 == 0)
    const DmaShare share = unpack_share(c.l[0].share[wave]);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned char* const sb0 = smem + (lane >> 5) * NPL * plane_bytes + (wave * 32 + (lane & 31)) * 16;
    const unsigned char* const sa0 = smem + 2 * NPL * plane_bytes + lane * 16;
#ifdef ESR_TRACE
    unsigned long long* const tr = a0.trace ? a0.trace + (size_t)blockIdx.x * 128 : nullptr;
    int tslot = 2;
    if (tr && tid == 0) { tr[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4); tr[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20); tr[126] = wall_clock64(); }
#define ESR_TR() do { if (tr && tid == 0 && tslot < 126) tr[tslot++] = __builtin_readcyclecounter(); } while (0)
    ESR_TR();
#else
#define ESR_TR() do { } while (0)
#endif
    // layer l's argument block.  (A run-time index into the by-value kernel argument would make the compiler keep the whole 2 KiB struct in
    // scratch memory, and everything read from it — every copy's base pointer — would count as divergent: static indices, selected by a uniform switch.)
    auto layer = [&](const int li) __attribute__((always_inline)) {
        switch (li) {
            case 0: return c.l[0];
            case 1: return c.l[1];
            case 2: return c.l[2];
            default: return c.l[3];
        }
    };
    auto issue = [&](const ConvArgs& a, const FetchState<MAXS>& f, const DmaShare& sh, const int cp, const unsigned stage) __attribute__((always_inline)) {
        const Bases<NPL> bs = make_bases<NPL, MT, NPW>(a, cp, b);
        dma_chunk<NPL, MT, NPW, 0>(f, bs, sh, stage, plane_bytes, true, 0, wave);
    };
    // rows [lo, hi) of layer l that this workgroup computes
    auto extent = [&](const int l, int& lo, int& hi) __attribute__((always_inline)) {
        const int e = c.n - 1 - l;
        lo = y0 - e < 0 ? 0 : y0 - e;
        hi = y0 + TH + e > H ? H : y0 + TH + e;
    };
    // ---- the current pass: layer l, rows [ys, ys + th); `drain`: its first chunk waits for the previous layer's stores
    int l = 0, ys, yhi, th;
    extent(0, ys, yhi);
    th = yhi - ys < TH ? yhi - ys : TH;
    bool drain = false;
    FetchState<MAXS> fs = setup_tile<MAXS>(a0, x0, ys, wave, lane);
    DmaShare sh = pass_share(share, th, P, a0.nslots, wave);
    issue(c.l[0], fs, sh, 0, lds0);
    __builtin_amdgcn_sched_barrier(0);
    ESR_TR();
    // ---- the pass behind it
    int ln = 0, ysn = 0, yhin = 0, thn = 0;
    bool more_passes = true, drain_n = false;
    FetchState<MAXS> fsn = fs;
    DmaShare shn = sh;

    auto pass = [&](auto RR_T) __attribute__((always_inline)) {
        constexpr int RR = decltype(RR_T)::value;
        const ConvArgs a = layer(l);
        f32x16 acc[MT][RR];
        EpiCoord<RR> ec;
        {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));           // (keeps the per-lane set-up behind the copies issued above: see conv3x3_tile_kernel)
            float bz[MT][16];
            bias_seed<MT>(a.bias, lane_o >> 5, bz);
#pragma unroll
            for (int r = 0; r < RR; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[0][r][i] = bz[0][i];
            ec = epi_coords<RR, 0>(a, x0, ys, wave, lane_o, th);
        }
        ESR_TR();                                      // per pass: [arguments selected + seeded | K loop done | next pass set up, its copies issued | stored]
        const int ncp = a.ncp;
        for (int cp = 0; cp < ncp; ++cp) {
            const int st = cp & 1;
            const bool more = cp + 1 < ncp;
            if (more) issue(a, fs, sh, cp + 1, lds0 + ((cp + 1) & 1) * stage_bytes);
            // the first chunk of a layer that reads what this workgroup stored a moment ago: everything outstanding — those stores included — has
            // to be complete on every wave before the barrier, and the copies of the first DEPENDENT chunk (index >= 2) are issued behind it
            if (drain && cp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else wait_vm_upto<MAXCNT>(more ? dma_count<NPL, MT, NPW, 0>(sh, true) : 0);
            __syncthreads();
            chunk_mfma<NPL, MT, RR, NPW, FMT, true, 3>(acc, sa0 + st * stage_bytes, sb0 + st * stage_bytes, P, plane_bytes);
            __syncthreads();
        }
        ESR_TR();
        // the next pass (of this layer, or the first of the next): set up and its first chunk's copies in flight under this pass's stores.  Chunk 0
        // of any layer is never an output of the chain (checked on the host), so it may be fetched before the stores below have landed.
        ln = l; ysn = ys + th; yhin = yhi; drain_n = false;
        if (ysn >= yhi) {
            ln = l + 1;
            more_passes = ln < c.n;
            if (more_passes) { extent(ln, ysn, yhin); drain_n = true; }
        }
        if (more_passes) {
            thn = yhin - ysn < TH ? yhin - ysn : TH;
            fsn = setup_tile<MAXS>(a0, x0, ysn, wave, lane);
            shn = pass_share(share, thn, P, a0.nslots, wave);
            const ConvArgs an = layer(ln);
            issue(an, fsn, shn, 0, lds0);
            __builtin_amdgcn_sched_barrier(0);
        }
        ESR_TR();
        conv_epilogue<NPL, MT, RR, EPI, FMT, PARTLO>(a, acc, b, ec, lane);
        ESR_TR();
    };
    while (more_passes) {
        // column tiles of 32 flattened pixels per wave that hold rows of this pass (wave w: tiles w, w + NW, ...)
        const int ncol = (th * P + 31) >> 5;
        const int nr = (ncol + NW - 1) / NW;
        if (nr <= 1) pass(std::integral_constant<int, 1>{});
        else if (nr == 2) pass(std::integral_constant<int, 2>{});
        else pass(std::integral_constant<int, 3>{});
        l = ln; ys = ysn; yhi = yhin; th = thn; drain = drain_n; fs = fsn; sh = shn;
    }
#ifdef ESR_TRACE
    if (tr && tid == 0) tr[127] = wall_clock64();
#endif
}

template <int NPL, int EPI, int FMT, int NPW>
int launch_chain(const ChainArgs& c, hipStream_t s, int query_only) {
    if (query_only) return ESR_OK;
    void (*k)(const ChainArgs) = conv3x3_chain_kernel<NPL, EPI, FMT, NPW>;
    ESR_ALLOW_160K_LDS(k);
    const ConvArgs& a = c.l[0];
    const size_t stage = (size_t)2 * NPL * a.NPIX_L * 16 + (size_t)9 * NPW * 1024;
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(k, dim3(a.ntiles), dim3(NTHREADS), 2 * stage, s, c);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}

template <int NPL, int FMT, int NPW>
int launch_chain_epi(const ChainArgs& c, int epi, hipStream_t s, int q) {
    switch (epi) {
        case 0: return launch_chain<NPL, 0, FMT, NPW>(c, s, q);
        case EPI_OUT2: return launch_chain<NPL, EPI_OUT2, FMT, NPW>(c, s, q);
        case EPI_MASK: return launch_chain<NPL, EPI_MASK, FMT, NPW>(c, s, q);
        default: return ESR_E_UNSUPPORTED;
    }
}

}  // namespace

int esr_internal_chain_launch(const void* chain_args, const int* variant, hipStream_t stream, int query_only) {
    const ChainArgs& c = *(const ChainArgs*)chain_args;
    const ConvVariant& v = *(const ConvVariant*)variant;
    if (c.n < 2 || c.n > CHAIN_MAX || v.mt != 1 || v.nst != 2 || v.tmode != 0 || v.partlo || v.ntile != 1) return ESR_E_UNSUPPORTED;
    if (v.npl == 1 && v.fmt == 0 && v.npw == 1) return launch_chain_epi<1, 0, 1>(c, v.epi, stream, query_only);      // bf16
    if (v.npl == 2 && v.fmt == 0 && v.npw == 2) return launch_chain_epi<2, 0, 2>(c, v.epi, stream, query_only);      // split bf16 (fp32-class)
    if (v.npl == 1 && v.fmt == 1 && v.npw == 1) return launch_chain_epi<1, 1, 1>(c, v.epi, stream, query_only);      // f16
    return ESR_E_UNSUPPORTED;
}
