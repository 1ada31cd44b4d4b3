/*
 * esr_hip.h — C-ABI of libesr_hip.so: the MI355X (gfx950) kernels behind the reference's
 * RRDB generator + Consistency Enforcing Module (CEM) hot path.
 *
 * The reference (YuvalBahat/Explorable-Super-Resolution) is pure Python/PyTorch and has no FFI of
 * its own; the operators it reaches through torch.nn are listed next to each entry point, with
 * the reference file:line they replace.  A maintainer binds these with ctypes;
 * the in-tree binding is explorable-super-resolution_amd/esr_hip/_lib.py.
 *
 * Conventions
 *   - plain C: pointers + sizes only; every pointer is a caller-owned DEVICE pointer unless it is a
 *     descriptor struct (host memory, read during the call).
 *   - return 0 on success, <0 on bad argument / unsupported shape (ESR_E_*); never throws, never
 *     allocates device memory, never synchronises the stream: work is enqueued on `stream`
 *     (hipStream_t).  Two entry points copy a HOST descriptor table to the device
 *     (esr_pack_batch_upload, esr_conv3x3_wgrad_batch[_upload]): pageable memory, staged by the
 *     runtime before the call returns — host-blocking for the duration of that staging and not
 *     capturable in a HIP graph; their *_run counterparts have no host traffic.
 *   - stateless and thread-safe: one process per GPU or several host threads may call concurrently.
 *
 * Internal activation layout ("act view"), used between the conv kernels so that dense-block
 * concatenation is zero-copy and every MFMA operand fragment is one aligned 16-byte read:
 *     [B][CG][H+2][W+2][8]  of bf16, i.e. channels in groups of 8 ("cg"), one 16-byte vector per
 *     pixel per group, with a ONE-PIXEL ZERO BORDER stored in memory (conv zero padding costs no
 *     bounds logic; producers never write the border).
 *   `hi` holds bf16(x); `lo` (optional) holds bf16(x - hi): together ~16 mantissa bits.  With both
 *   planes the conv kernels evaluate hi*hi + hi*lo + lo*hi on the bf16 MFMA pipe with fp32
 *   accumulation ("split-bf16", rel. error ~1e-5 vs fp32); with lo == NULL they run plain bf16.
 */
#ifndef ESR_HIP_H
#define ESR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESR_OK 0
#define ESR_E_ARG (-1)        /* NULL / inconsistent argument */
#define ESR_E_UNSUPPORTED (-2) /* shape outside what the kernels implement */
#define ESR_E_LAUNCH (-3)      /* HIP reported a launch error */

typedef void* esr_stream_t;   /* hipStream_t */

/* Element formats of an activation view.  BF16 with a `lo` plane is the fp32-class mode (split operands, 3 MFMAs per product);
 * BF16 / F16 without `lo` are the single-MFMA reduced-precision modes; F16 with `lo` is the 2-MFMA mode (f16 weights x f16 hi+lo
 * activations).  The reference has no counterpart of these; the F16 modes are inference-only (gradients would need loss scaling).
 * Weight packs use the matching code in their `split` argument: 0 bf16, 1 split bf16 (hi+lo), 2 f16 (one plane), 3 f16 hi+lo. */
#define ESR_FMT_BF16 0
#define ESR_FMT_F16 1

/* A view of `ncg` consecutive channel groups inside an activation buffer. */
typedef struct {
    void* hi;                /* 16-bit planes (bf16 or fp16, see fmt), NULL = view absent */
    void* lo;                /* residual planes (value = hi + lo) or NULL: single-plane view */
    int32_t ncg;             /* number of 8-channel groups in this view */
    int32_t H, W;            /* interior size; the buffer holds (H+2) x (W+2) pixels per group */
    int64_t batch_stride;    /* in 16-byte pixel vectors */
    int64_t cg_stride;       /* in 16-byte pixel vectors (normally (H+2)*(W+2)) */
    int32_t fmt;             /* element format of the planes: ESR_FMT_BF16 (default) or ESR_FMT_F16; both as hi [+ lo] */
} esr_act_view;

/* ---- conv3x3 (+bias, +LeakyReLU, +scaled residuals, + fused nearest upsample of the input) ----
 * Replaces, per call, the reference chain
 *     torch.cat(inputs) -> nn.Conv2d(k=3,s=1,p=1,bias) -> LeakyReLU(0.2) -> .mul(0.2) + residual
 *   codes/models/modules/block.py:129-146 (conv_block), :230-235 (ResidualDenseBlock_5C.forward),
 *   :262-270 (RRDB.forward), :85-97 (ShortcutBlock.forward), :293-309 (Upsampler + upconv_blcok),
 *   codes/models/modules/architecture.py:288-301 (latent re-concatenation before every conv).
 * Also used as the data-gradient of the same conv (weights packed transposed+flipped).
 *
 *   out = alpha * act(conv(in0 ++ in1) + bias) + beta1*res1 + beta2*res2
 *
 * Limits checked by the entry point (ESR_E_ARG / ESR_E_UNSUPPORTED otherwise): 0 < act_slope <= 1 and alpha >= 0 (the epilogue evaluates
 * alpha * LeakyReLU(y) as max(alpha*y, alpha*act_slope*y)); H + 2 and W + 2 below 32768, upsample <= 8; every view's image (ncg * cg_stride * 16
 * bytes) and the fp32 destination's image below 4 GiB (per-lane addresses are a uniform per-image base + a 32-bit offset); res1 / res2 cover
 * every output group (ncg * 8 >= cout); mask_src covers the masked groups.
 *
 * When res1 is a channel-group slice of in1 itself (same strides; the RDB's  conv5*0.2 + x, block.py:235) and act_slope == 1,
 * the library notices it from the pointers and takes the residual from the input tile it stages on chip anyway (no extra reads);
 * the result is the same expression evaluated in fp32.
 */
typedef struct {
    esr_act_view in0;        /* optional leading segment (latent Z group); ncg = 0 when absent */
    esr_act_view in1;        /* main segment; with upsample>1 it is read as in1[y/upsample][x/upsample] */
    int32_t upsample;        /* 1, 2 or 3 */
    const void* wpack;       /* weights packed by esr_pack_conv_weights for (in0.ncg + in1.ncg) groups */
    const float* bias;       /* [mtiles*32] fp32, zero padded; NULL = no bias */
    int32_t cout;            /* real output channels: <= 64, or a multiple of 64 — then the launch covers cout/64 output slices at once and `wpack`
                              * holds one 64-row pack per slice, slice s at byte offset s * esr_conv_wpack_bytes(in0.ncg + in1.ncg, 64, fmt)
                              * (no out_nchw / pixel_shuffle with slices) */
    int32_t B, H, W;         /* output interior size (= input size * upsample) */
    float act_slope;         /* 1.0f: identity; 0.2f: LeakyReLU(0.2) */
    float alpha;
    esr_act_view res1; float beta1;   /* optional (hi == NULL: absent); same H, W as the output */
    esr_act_view res2; float beta2;
    esr_act_view out;        /* act-layout destination (ncg*8 >= cout), or hi == NULL */
    esr_act_view out2;       /* optional second destination (same values; may be hi-only when `out` is hi+lo: a one-plane copy) */
    float* out_nchw;         /* optional fp32 [B][cout][H][W] destination */
    /* data-gradient helper: multiply the result for output groups [mask_cg0, mask_cg1) by
     * act'(mask_src) = (mask_src > 0 ? 1 : mask_slope) AFTER the residual add (LeakyReLU backward of
     * the layer that produced those channels; block.py:18). mask_src.hi == NULL: disabled. */
    esr_act_view mask_src; int32_t mask_cg0, mask_cg1; float mask_slope;
    /* scheduling hint, no effect on the result: walk the tiles (and so the images) last to first.  Consecutive layers of a network
     * re-read what the previous launch just touched; alternating the direction lets the tail of one launch, still in the 256 MB
     * Infinity Cache, be the head of the next. */
    int32_t reverse_order;
    /* planes of the weight pack: 0 = the format's default (bf16: as many as the activations; f16: one), or 1 / 2 explicitly.
     * f16 hi+lo activations with 2 weight planes is the 3-MFMA fp16 form (2^-22 operands) used for the few layers whose weight
     * rounding dominates the output error; with 1 plane it is the 2-MFMA form. */
    int32_t weight_planes;
    /* fp16 formats only: number of LEADING channel groups of in1 whose lo plane carries data (0 = all of them).  The groups behind
     * them are single-plane intermediates (a dense block's conv outputs, consumed only inside the block, where 11 bits suffice): their
     * lo plane is neither read nor multiplied.  Likewise `out.lo == NULL` with hi+lo inputs stores the result as one fp16 plane. */
    int32_t in1_lo_groups;
    /* pixel-shuffle store (codes/models/modules/block.py:278-291: conv to out_nc*r^2 channels, nn.PixelShuffle(r), act): with
     * pixel_shuffle = r > 1 the launch's output rows are taken in groups of 8 ("row groups"); row group g = ps_rowgroup0 + (row / 8)
     * holds the 8 channels of output group g / r^2 at sub-position s = g % r^2, and is stored to out[group][r*y + s / r][r*x + s % r]
     * (out.H == r*H, out.W == r*W).  The caller packs the weights with the matching row map: row -> conv channel
     * ((g / r^2)*8 + row % 8) * r^2 + s.  The activation commutes with the shuffle and is applied before it.  0: plain store. */
    int32_t pixel_shuffle;
    int32_t ps_rowgroup0;
    /* structurally sparse weights, a HINT: 9-bit tap masks (bit 3*dy + dx of the tap as the pack stores it; 0 = all nine taps).  The caller
     * promises that K chunk cp (the cp-th PAIR of input channel groups) has non-zero weights only at the taps of
     * tap_mask_k[(cp >> tap_mask_k_shift) & 3], and the j-th 32-row tile of the output (j = 2 * slice + tile) only at those of
     * tap_mask_m[j & 3]; the library skips blocks outside the masks where it has a kernel for the pattern and multiplies the zeros
     * otherwise — same result.  Patterns with kernels: the critic's 4x4 stride-2 convs (codes/models/modules/architecture.py:452-480) run as
     * 3x3 convs over the space-to-depth input, 16 non-zero blocks of 36 (esr_hip/critic.py): forward (K masks {432, 216, 54, 27}, shift 1) and
     * data gradient (M masks {27, 54, 216, 432}); plain epilogue, bf16 formats, cout a multiple of 64. */
    int32_t tap_mask_k[4];
    int32_t tap_mask_k_shift;
    int32_t tap_mask_m[4];
    /* optional split-K workspace (caller-owned device memory, `k_split_ws_floats` floats; NULL: never split).  A launch of few output tiles
     * with a long K axis (the critic's 256- / 512-channel layers on 16x16 ... 4x4 maps: 24-100 workgroups x 32-128 K chunks on 256 CUs) may be
     * run as S = 2, 4 or 8 sets of workgroups that each contract 1/S of the input channels into an fp32 slab of B*cout*H*W floats, followed by a
     * second launch that adds the slabs in a fixed order and stores `out`.  The library decides from its tiling (S * workgroups <= 320, >= 8 chunks
     * per set, S slabs fit); only plain launches qualify (act_slope 1, no residual / mask / out2 / out_nchw / pixel_shuffle / upsample / in0,
     * bf16 formats, cout a multiple of 64).  The result differs from the unsplit launch by fp32 summation order only. */
    float* k_split_ws;
    int64_t k_split_ws_floats;
    /* scheduling hint, no effect on the result: 0 = the library picks the form by launch size (launches with no more workgroups than CUs
     * pipeline their own LDS-DMA through two stages, larger ones run two single-stage workgroups per CU); 1 / 2 = force that form. */
    int32_t lds_stages;
    /* fp16 formats only, optional (NULL: off): range watch.  The reference computes in fp32 (codes/models/modules/block.py:230-235) and has no
     * magnitude cliff; an fp16 activation plane saturates at 65504.  When a value the launch stores has magnitude >= 32768 (one binade of head
     * room left) or is not finite, the kernel does atomicMin(*range_flag, range_tag) — no extra pass, nothing on the path of a launch that stays
     * in range.  The caller initialises the word to 0xFFFFFFFF, gives every launch of a pass its own tag (its index) and reads the word when it
     * consumes the pass's result: the smallest tag names the first layer that left the range. */
    uint32_t* range_flag;
    uint32_t range_tag;
} esr_conv3x3_desc;

int esr_conv3x3(const esr_conv3x3_desc* d, esr_stream_t stream);

/* Adjoint of the pixel-shuffle store: dst[b][g*r^2 + s][y][x] = src[b][g][r*y + s / r][r*x + s % r] for every group g of `src`
 * (dst.ncg == src.ncg * r^2, src.H == r*dst.H): the gradient of the shuffled tensor laid out in the conv's row-group order
 * (autograd of nn.PixelShuffle, block.py:287). */
int esr_pixel_unshuffle(const esr_act_view* src, int r, const esr_act_view* dst, int B, esr_stream_t stream);

/* Packed-weight size in bytes for `ncg_in` input groups and `cout` output channels
 * (`split` = 1: bf16 hi+lo planes; 0: bf16 hi only; 2: f16, one plane; 3: f16 hi+lo planes). */
size_t esr_conv_wpack_bytes(int ncg_in, int cout, int split);

/* Pack nn.Conv2d weights [cout_w][cin_w][3][3] (fp32, device) into MFMA fragment order.
 *   kmap[ncg_in*8]  : for each (group, lane-in-group) the index into the tensor's "K" channel axis, or -1 (zero)
 *   mmap[mtiles*32] : for each output row the index into the "M" channel axis, or -1 (zero)
 *   transposed = 0 : forward     — M axis = dim 0 (cout_w), K axis = dim 1 (cin_w), tap (dy,dx) as stored
 *   transposed = 1 : data-grad   — M axis = dim 1 (cin_w),  K axis = dim 0 (cout_w), tap flipped (2-dy,2-dx)
 * kmap/mmap are device int32 arrays; every weight is multiplied by `scale` before it is split.
 * The pack is chunk-major ([pair of K groups][tap][mtile][hi|lo][lane]): the K axis of one launch may concatenate several tensors
 * by packing each at byte offset  first_group/2 * esr_conv_wpack_bytes(2, mtiles*32, split)  of one buffer (first_group even) —
 * how the dense-block data gradient sums  W_5^T*dy_5 + ... + W_j^T*dy_j  in a single conv (block.py:230-235 backwards). */
int esr_pack_conv_weights(const float* w, int cout_w, int cin_w, const int32_t* kmap, int ncg_in,
                          const int32_t* mmap, int mtiles, int transposed, int split, float scale, void* wpack,
                          esr_stream_t stream);

/* The same for MANY tensors in one launch (a training step changes every layer's weights).  `descs` is a HOST array; upload once
 * into caller-owned device `workspace` (>= esr_pack_batch_workspace_bytes), then esr_pack_batch_run re-packs all of them from the
 * CURRENT contents of their `w` whenever the parameters changed — no host traffic in steady state.  upload returns the number of
 * workgroups to pass to run (> 0), or ESR_E_*; it must be repeated when any pointer in the descriptors changes. */
typedef struct {
    const float* w; int32_t cout_w, cin_w; const int32_t* kmap; int32_t ncg_in; const int32_t* mmap;
    int32_t mtiles, transposed, split; float scale; void* wpack;
} esr_pack_desc;
int64_t esr_pack_batch_workspace_bytes(const esr_pack_desc* descs, int n);
int64_t esr_pack_batch_upload(const esr_pack_desc* descs, int n, void* workspace, int64_t workspace_bytes, esr_stream_t stream);
int esr_pack_batch_run(const void* workspace, int n, int64_t nblocks, esr_stream_t stream);

/* ---- layout conversion at the module boundary ----
 * fp32 NCHW -> act view.  Writes channels [c0, c0+nc) of `src` ([B][C][h][w]; image b starts at
 * src + b*src_batch_stride floats, 0 = C*h*w — lets the HR-resolution latent be read through the raw
 * `view` the reference applies to it, SRRaGAN_model.py:233 / architecture.py:283) into group 0.. of `dst`
 * (zero-filling unused lanes of the last group and the 1-px border), with
 *   pad  : replicate padding by `pad` pixels on every side (CEM_PyTorch.forward eval-mode LR_padder /
 *          HR_padder, codes/CEM/CEMnet.py:286-295), dst interior = (h+2*pad)/down x (w+2*pad)/down
 *   down : 1, or the bilinear /down of architecture.py:284 (F.interpolate(scale_factor=1/sf,'bilinear',
 *          align_corners=False)) applied AFTER the padding (down divides h+2*pad). */
int esr_pack_nchw(const float* src, int64_t src_batch_stride, int B, int C, int h, int w, int c0, int nc, int pad, int down,
                  const esr_act_view* dst, esr_stream_t stream);
/* act view -> fp32 NCHW [B][nc][H][W] (hi + lo); used by tests and by callers that want features. */
int esr_unpack_nchw(const esr_act_view* src, int B, int nc, float* dst, esr_stream_t stream);
/* zero `n16` 16-byte vectors (border initialisation of freshly allocated activation buffers) */
int esr_zero(void* p, int64_t n16, esr_stream_t stream);

/* ---- Consistency Enforcing Module filters (fixed depth-wise taps, fp32, NCHW) ----
 * codes/CEM/CEMnet.py:243-252 (Filter_Layer) x3 as built in CEM_PyTorch.__init__ (:254-281).
 * `taps` are device fp32 [k][k] arrays exactly as the reference stores them in Filter_OP.weight[0,0]. */

/* DownscaleOP (CEMnet.py:272-275): replicate-pad floor(k/2), correlate with taps (= rot90(ds_kernel,2)),
 * keep [pre::sf, pre::sf].  y: [B][C][sf*h][sf*w] -> d: [B][C][h][w].
 * If lr != NULL the kernel writes  lr_pad - d  instead, where lr_pad is `lr` ([B][C][h-2*lr_pad][w-2*lr_pad])
 * replicate-padded by lr_pad (fuses the LR_padder and the x - D(g) of the projection). */
int esr_cem_downscale(const float* y, int B, int C, int h, int w, int sf, int pre, const float* taps, int k,
                      const float* lr, int lr_pad, float* d, esr_stream_t stream);
/* Conv_LR_with_Inv_hTh_OP (CEMnet.py:261-263): replicate-pad floor(k/2), correlate. [B][C][h][w] -> same. */
int esr_cem_lrfilter(const float* x, int B, int C, int h, int w, const float* taps, int k, float* out,
                     esr_stream_t stream);
/* Upscale_OP (CEMnet.py:264-271): zero-stuff at (pre,pre), replicate-pad floor(k/2), correlate with taps
 * (= ds_kernel*sf^2).  f: [B][C][h][w] -> [B][C][sf*h][sf*w], restricted to the crop window
 * [crop, sf*h-crop) x [crop, sf*w-crop) (HR_unpadder, CEMnet.py:311).
 *   mode 0: out = U(f)
 *   mode 1: out = g + U(f)                      (projection  g + U(K(x - D(g))), CEMnet.py:305-310)
 *   mode 2: out = U(f_x) + tanh(g - U(f_g)) * range      (sigmoid_range_limit; f = f_x, f2 = f_g)
 *   mode 3: out = U(f_x), out2 = g - U(f_g)              (decomposed_output)
 * g/out/out2 are [B][C][sf*h][sf*w] and [B][C][sf*h-2crop][sf*w-2crop] respectively. */
int esr_cem_upscale(const float* f, const float* f2, int B, int C, int h, int w, int sf, int pre,
                    const float* taps, int k, const float* g, int crop, int mode, float range,
                    float* out, float* out2, esr_stream_t stream);

/* Separable variants of the three CEM filters for rank-one taps  taps[a][b] = tv[a] * th[b]  (the bicubic ds_kernel and its inv_hTh are
 * rank one; estimated / anisotropic kernels are not and use the 2-D entry points): a horizontal and a vertical 1-D pass on the tile a
 * workgroup holds on chip, 2k instead of k^2 MACs per output.  Arguments, index conventions, fused forms and results as esr_cem_downscale /
 * esr_cem_lrfilter / esr_cem_upscale (up to fp32 rounding of the tap products); `tv`, `th`: device arrays of k floats.
 * ESR_E_UNSUPPORTED: the tile does not fit on chip for this (sf, k) — use the 2-D entry point. */
int esr_cem_downscale_sep(const float* y, int B, int C, int h, int w, int sf, int pre, const float* tv, const float* th, int k, const float* lr,
                          int lr_pad, float* d, esr_stream_t stream);
int esr_cem_lrfilter_sep(const float* x, int B, int C, int h, int w, const float* tv, const float* th, int k, float* out, esr_stream_t stream);
int esr_cem_upscale_sep(const float* f, const float* f2, int B, int C, int h, int w, int sf, int pre, const float* tv, const float* th, int k,
                        const float* g, int crop, int mode, float range, float* out, float* out2, esr_stream_t stream);
/* esr_cem_lrfilter_sep followed by esr_cem_upscale_sep as ONE launch: every mode of esr_cem_upscale_sep with K(e) (and K(e2)) in place of f (f2),
 * K = the kf-tap separable LR filter tvf x thf with replicate padding (CEM_PyTorch.forward's Conv_LR_with_Inv_hTh_OP in front of Upscale_OP,
 * codes/CEM/CEMnet.py:305-309).  Each tile filters its own window of e on chip, with the separate kernel's passes and summation order: results are
 * bit-identical to the two launches, the LR-sized intermediate and its launch are gone.  ESR_E_UNSUPPORTED: the windows do not fit on chip for this
 * (sf, k, kf) — run the two launches. */
int esr_cem_filter_upscale_sep(const float* e, const float* e2, int B, int C, int h, int w, int sf, int pre, const float* tvf, const float* thf, int kf,
                               const float* tv, const float* th, int k, const float* g, int crop, int mode, float range, float* out, float* out2,
                               esr_stream_t stream);

/* Which kernel form esr_cem_downscale_sep (op 0) / esr_cem_upscale_sep (op 1) / esr_cem_lrfilter_sep (op 2; k = its tap count, sf and pre unused) runs for an image geometry — decided from these arguments alone, never
 * from the batch, so that a batch and its chunks run the same arithmetic: 0 = tile kernel (a workgroup stages a window in LDS), 1 = streaming tile
 * kernel (x8 downscale of small images), 2 = wave-streaming kernel (low-resolution images of at least 64 x 64 pixels, sf 2 / 3 / 4 / 8 with
 * ceil(k / sf) in 4..6; the upscale also needs pre > 0; the LR filter k = 27 or 35 and 128 x 128 pixels): a wave walks down a strip with the vertical pass in registers.  The
 * upscale forms agree to the bit, the downscale and LR-filter forms to fp32 rounding (different order of the two passes).  For tests and documentation; < 0: ESR_E_ARG. */
int esr_cem_sep_form(int op, int sf, int k, int pre, int h, int w);

/* ---- backward-pass helpers (autograd of the reference's torch ops) ----
 * out = alpha*A + beta*sumpool_s(Bv), optionally * LeakyReLU'(mask) — gradient of the nearest upsample
 * (block.py:293-300), of residual sums, and of LeakyReLU (block.py:18).  A / Bv / mask may be NULL. */
int esr_act_combine(const esr_act_view* A, float alpha, const esr_act_view* Bv, float beta, int s, const esr_act_view* mask,
                    float mask_slope, const esr_act_view* out, int B, esr_stream_t stream);
/* Adjoint of esr_pack_nchw (replicate padding folded back onto the edge pixels, bilinear /down taps transposed):
 * act-layout gradient -> fp32 NCHW gradient of channels [c0, c0+nc) of the un-padded source. */
int esr_unpack_grad_nchw(const esr_act_view* G, float* dst, int64_t dst_batch_stride, int B, int C, int h, int w, int c0, int nc,
                         int pad, int down, int accumulate, esr_stream_t stream);
/* Magnitude management of fp16 gradients (the 'mixed' precision's backward; no counterpart in the reference, whose gradients are fp32):
 * esr_grad_absmax: *slot = max(*slot, bit pattern of max|hi|) over the view's groups (fp16 planes; the caller zeroes the slot once).
 * esr_grad_scale : dst = src * f (hi and lo planes; dst may be src), f a power of two taken from device memory:
 *   slot != NULL : f = 2^k with max|hi| * 2^k in [2^(exp-1), 2^exp)  (k = 0 when the view is all zero); scale_out, if given, receives
 *                  scale_in[0] * f — the running scale of the gradients in flight;
 *   slot == NULL : f = scale_in[0] / scale_den[0].
 * Nothing is read back by the host. */
int esr_grad_absmax(const esr_act_view* v, int B, uint32_t* slot, esr_stream_t stream);
int esr_grad_scale(const esr_act_view* src, const esr_act_view* dst, int B, const uint32_t* slot, int exp, const float* scale_in,
                   const float* scale_den, float* scale_out, esr_stream_t stream);
/* Adjoint of a CEM filter (see csrc/esr_cem.hip): y[q] = sum taps * frame[clamp(q*sq+oq + a - p)], unknowns at n*sn+on.
 * tabs: device fp32 [3][3][k][k] prefix/plain/suffix tap tables (esr_hip/cem_ops.py builds them). */
int esr_cem_adjoint(const float* dy, int B, int C, int hq, int wq, int sq, int oq, int Ny, int Nx, const float* tabs, int k,
                    int hn, int wn, int sn, int on, float* dx, int accumulate, esr_stream_t stream);
/* The same adjoint for rank-one taps (= outer(tv, th): the bicubic kernels and their inv_hTh): two 1-D passes through `tmp` (B*C*hq*wn floats).
 * tabs_v / tabs_h: device fp32 [3][k] prefix / plain / suffix sums of the vertical / horizontal factor.  dx = (base ? base : 0) + alpha * adjoint. */
int esr_cem_adjoint_sep(const float* dy, int B, int C, int hq, int wq, int sq, int oq, int Ny, int Nx, const float* tabs_v, const float* tabs_h, int k,
                        int hn, int wn, int sn, int on, float* tmp, const float* base, float alpha, float* dx, esr_stream_t stream);

/* ---- conv3x3 weight / bias gradient (autograd of nn.Conv2d, block.py:141-142) ----
 *   dw[co][lat+ci][dy][dx] += alpha * sum_{b,y,x} dy[b,co,y,x] * x[b,ci,(y+dy-1)/up,(x+dx-1)/up]     (zero padded)
 *   dw[co][e][dy][dx]      += ... with xlat for the latent channels e < lat;   db[co] += alpha * sum dy
 * dw ([cout][lat+cin_main][3][3]) and db ([cout], may be NULL) are fp32 and ACCUMULATED into (zero them first).
 * The pixel sum is split over ~3 workgroups per CU; their partial tiles go through `workspace` (caller-owned device memory of at
 * least esr_conv3x3_wgrad_workspace_floats(d) floats, contents undefined afterwards) and a second small kernel folds them. */
typedef struct {
    esr_act_view dy;         /* gradient w.r.t. the conv's (pre-activation) output, cout channels */
    esr_act_view x;          /* the conv's main input (before the nearest upsample when upsample > 1) */
    esr_act_view xlat;       /* optional latent segment (hi == NULL: absent) */
    int32_t lat;             /* real latent channels (<= 8) */
    int32_t upsample;
    int32_t cout, cin_main;
    int32_t B, H, W;         /* output-resolution size */
    float alpha;
    float* dw;
    float* db;
    float* workspace;
    int64_t workspace_floats;
    /* a HINT, as esr_conv3x3_desc.tap_mask_k: the weights whose gradient this is are structurally zero outside the taps of tap_masks[i & 3] for
     * the i-th 32-channel tile of the main input; dw entries outside the masks receive zeros or are left untouched (0 = all taps) */
    int32_t tap_masks[4];
} esr_wgrad_desc;
int64_t esr_conv3x3_wgrad_workspace_floats(const esr_wgrad_desc* d);   /* depends on B, H, W, cout, cin_main, lat only; <0: bad argument */
int esr_conv3x3_wgrad(const esr_wgrad_desc* d, esr_stream_t stream);
/* The weight gradients of MANY layers in one launch (a whole backward pass): with hundreds of layers there are enough
 * (layer, 32-input-channel tile, 32-output-channel tile) blocks to fill the chip without splitting the pixel sum, so the per-layer
 * reduction traffic and launch tails disappear.  `descs` is a HOST array (its workspace fields are ignored); `workspace` is caller-
 * owned device memory of at least esr_conv3x3_wgrad_batch_workspace_bytes(descs, n) bytes (descriptor table + partial sums).
 * All dy / x buffers must stay alive and unmodified until the launch has run; every layer must use the same operand format. */
int64_t esr_conv3x3_wgrad_batch_workspace_bytes(const esr_wgrad_desc* descs, int n);
int esr_conv3x3_wgrad_batch(const esr_wgrad_desc* descs, int n, void* workspace, int64_t workspace_bytes, esr_stream_t stream);
/* The same in two steps, for callers whose descriptor set repeats from one backward pass to the next (pooled gradient buffers, the
 * allocator handing back the same dW storage): _upload writes the descriptor table into `workspace` (a host-blocking copy from pageable
 * memory: NOT capturable in a HIP graph) and fills `plan`; _run enqueues the launch from the table already on the device (capturable; no
 * host traffic).  esr_conv3x3_wgrad_batch == _upload followed by _run. */
/* (plan fields are the library's own bookkeeping between _upload and _run — opaque to callers: nwg = workgroups of the per-pair launch, `reserved` =
 * block-form items | their LDS KiB << 23 when the experimental block decomposition is enabled, 0 otherwise) */
typedef struct { int64_t nwg, table_bytes; int32_t n, max_red, split, f16, s2d, reserved; } esr_wgrad_batch_plan;
int esr_conv3x3_wgrad_batch_upload(const esr_wgrad_desc* descs, int n, void* workspace, int64_t workspace_bytes, esr_wgrad_batch_plan* plan,
                                   esr_stream_t stream);
int esr_conv3x3_wgrad_batch_run(const void* workspace, const esr_wgrad_batch_plan* plan, esr_stream_t stream);
/* The same launch for a SECOND stream that runs under another stream's kernels (the generator backward: weight gradients of the layers whose
 * data gradients are done, while the data-gradient chain — small launches that leave most of every CU idle — goes on): one-plane operand sets run an
 * instantiation whose waves hold more than half of a SIMD's registers, i.e. ONE workgroup per CU, so that the other stream's workgroups always find
 * room; the results are those of _run.  hi+lo / space-to-depth sets: identical to _run. */
int esr_conv3x3_wgrad_batch_run_side(const void* workspace, const esr_wgrad_batch_plan* plan, esr_stream_t stream);
/* Resident workgroups per CU of the instantiation _run_side launches for one-plane operand sets (f16: the fp16 one, else bf16), as the runtime
 * reports it for the current device (hipOccupancyMaxActiveBlocksPerMultiprocessor): 1 when the register cap holds.  The cap comes from the
 * register allocator honouring a clobber — a property of the toolchain that built the library, not of the source — so callers that schedule a
 * two-stream backward around it (esr_hip/engine.py: wgrad_overlap) ask once and keep the one-stream form when the answer is not 1.  Negative:
 * ESR_E_LAUNCH (no device / query failed).  (The reference has no counterpart: autograd's weight gradients run wherever cuDNN puts them,
 * codes/models/SRRaGAN_model.py:418-499.) */
int esr_conv3x3_wgrad_side_occupancy(int f16);
/* A backward pass's layers as SEVERAL launches (one per gradient bucket of a data-parallel job, so that each bucket's all-reduce starts behind its
 * launch and overlaps the launches that follow — torch.distributed over RCCL; the reference's nn.DataParallel reduces after the whole backward,
 * codes/models/SRRaGAN_model.py:418-499): _unit returns the slicing granule the ONE-launch form would use for the whole set; passing it to
 * _part_workspace_bytes / _part_upload for every part keeps each layer's pixel sum cut exactly as in the one launch — the parts' results are
 * bit-identical to it.  _run is the same for parts. */
int64_t esr_conv3x3_wgrad_batch_unit(const esr_wgrad_desc* descs, int n);
int64_t esr_conv3x3_wgrad_batch_part_workspace_bytes(const esr_wgrad_desc* descs, int n, int64_t unit);
int esr_conv3x3_wgrad_batch_part_upload(const esr_wgrad_desc* descs, int n, void* workspace, int64_t workspace_bytes, esr_wgrad_batch_plan* plan,
                                        int64_t unit, esr_stream_t stream);
/* Every dw / db of an uploaded table moved by the same byte offset (all layers' gradients are views of one flat buffer and the caller got a new
 * flat buffer for this pass): patch the table on the device — one tiny launch, no host traffic, stream-ordered behind the previous run. */
int esr_conv3x3_wgrad_batch_rebase(void* workspace, const esr_wgrad_batch_plan* plan, int64_t delta_bytes, esr_stream_t stream);

/* ---- Z-objective kernels (reference: codes/Z_optimization.py:170-209, SoftHistogramLoss.ComputeSoftHistogram, gray-scale / patch-size-1
 * form).  Soft histogram of n values with K bins whose centres run from lo to hi:  h[k] = (1/n) sum_i exp(-(d(v_i, c_k) + eps)^2 / T), the
 * distance wrapped with period hi as the reference does.  The forward writes one partial (un-normalised) histogram of K doubles per slab of
 * pixels — esr_soft_hist_slabs(n) of them — which the caller sums and divides by n; the backward takes d loss / d h[k] (already divided by n)
 * and writes d loss / d v.  The reference builds the n x K matrix in float64; these kernels never materialise it. */
int64_t esr_soft_hist_slabs(int64_t n);
int esr_soft_hist_fwd(const float* v, int64_t n, int K, float lo, float hi, float T, float eps, double* partial, esr_stream_t stream);
int esr_soft_hist_bwd(const float* v, int64_t n, int K, float lo, float hi, float T, float eps, const float* gh, float* gv, esr_stream_t stream);

/* ---- Adam over many tensors in one launch ----
 * The reference steps both networks with torch.optim.Adam (codes/models/SRRaGAN_model.py:147-160: lr, betas, weight decay from the options);
 * torch's multi-tensor implementation costs ~4 ms of host time per step for the generator's 702 tensors.  Same update rule, same fp32
 * operation order (torch/optim/adam.py, no amsgrad / maximize):
 *     g += weight_decay * p;  m += (1 - beta1) * (g - m);  v = beta2 * v + (1 - beta2) * g * g;
 *     p -= (lr / bias_correction1) * m / (sqrt(v) / bias_correction2_sqrt + eps)
 * with bias_correction1 = 1 - beta1^t and bias_correction2_sqrt = sqrt(1 - beta2^t) computed by the caller for step t.
 * `tensors` is a HOST array; _upload writes it (plus a work table) into caller-owned device `workspace` (>= _workspace_bytes), blocks until
 * the copy is done and returns the chunk count to pass to _run (>= 0) or ESR_E_*; it is repeated only when a pointer changes.
 * esr_adam_table writes the same table into caller-owned HOST memory instead (>= _workspace_bytes; e.g. a pinned buffer) and returns the
 * chunk count: a caller whose gradient tensors move from step to step copies it to the device workspace itself, asynchronously on its
 * stream — no host synchronisation (the training loop of codes/train.py:90-192 has none between optimizer steps either). */
typedef struct { float* p; const float* g; float* m; float* v; int64_t n; } esr_adam_tensor;
int64_t esr_adam_workspace_bytes(const esr_adam_tensor* tensors, int n);
int64_t esr_adam_upload(const esr_adam_tensor* tensors, int n, void* workspace, int64_t workspace_bytes, esr_stream_t stream);
int64_t esr_adam_table(const esr_adam_tensor* tensors, int n, void* host_table, int64_t host_bytes);
int esr_adam_run(const void* workspace, int n, int64_t nchunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                 float bias_correction1, float bias_correction2_sqrt, esr_stream_t stream);

/* Per-image statistics of the SR output and their gradients — the reductions of the Z-search loop and of the L_struct training loss, one pass
 * over the image each (csrc/esr_zobj.hip):
 *   kind 0  masked STD       codes/Z_optimization.py:383-388  torch.std(image * mask, dim=(1,2,3))              sums[b] = (sum v, sum v^2, -)
 *   kind 1  TV_Loss          codes/Z_optimization.py:324-326                                                       sums[b] = (sum |dx|, sum |dy|, -)
 *   kind 2  structure tensor codes/models/modules/loss.py:49-62,141-151 (2x2 difference filters, (H-1) x (W-1) frame)  sums[b] = (sum ix^2, sum iy^2, sum ix iy)
 * x: fp32 [B][C][H][W]; v = clamp(x, 0, 1) when clamp01 (the reference's Output_Batch(within_0_1=True)), times mask [H][W] when given (the GUI's
 * image mask).  sums: [B][3] doubles, zeroed by the caller (the kernels add).  _grad: dx = sum_k coef[b][k] * d sums[b][k] / dx in closed form
 * (the caller folds the scalar ops after the reduction — sqrt, division by N — into coef); accumulate != 0 adds into dx. */
int esr_img_stats(const float* x, int B, int C, int H, int W, const float* mask, int clamp01, int kind, double* sums, esr_stream_t stream);
int esr_img_stats_grad(const float* x, int B, int C, int H, int W, const float* mask, int clamp01, int kind, const float* coef, float* dx, int accumulate,
                       esr_stream_t stream);

/* ---- the critic's glue: BatchNorm2d (training mode) + LeakyReLU, its gradient and the gradient of its gradient ----
 * Reference: Discriminator_VGG_128 (codes/models/modules/architecture.py:446-508): conv_block = nn.Conv2d -> nn.BatchNorm2d(affine, batch
 * statistics while training; block.py:25-35,129-146) -> LeakyReLU(0.2); the WGAN-GP penalty (codes/models/modules/loss.py:260-279)
 * differentiates the critic's input gradient, i.e. back-propagates through the BACKWARD of every one of these layers.  All three levels
 * are closed-form single-pass kernels on the conv kernels' activation layout (csrc/esr_critic.hip states the formulas):
 *   esr_bn_reduce mode 0 : sums[g][c] = (sum y, sum y^2)                                  esr_bn_finalize: mean, rstd, scale, shift, running stats
 *   esr_bn_apply  mode 0 : out0 = lrelu(scale*y + shift)                                  (torch: F.batch_norm(training=True) + leaky_relu)
 *   esr_bn_reduce mode 1 : sums[g][c] = (sum dyb, sum dyb*xhat),  dyb = dz * lrelu'       esr_bn_param_grads: dgamma, dbeta
 *   esr_bn_apply  mode 1 : out0 = d loss / d y                                            (autograd of the above)
 *   esr_bn_reduce mode 2 : sums[g][c] = (sum u, sum u*xhat, sum u*dyb)                    (u: cotangent of mode 1's result)
 *   esr_bn_apply  mode 2 : out0 = cotangent of dz, out1 = cotangent of y                  (torch: batchnorm_double_backward) + g_gamma via
 *                                                                                          esr_bn_param_grads
 * groups: the B images are `groups` runs of B/groups consecutive images, each normalised with its OWN batch statistics (the critic's
 * separate calls on the real, fake and interpolated batches executed as one launch); all per-channel arrays are [groups][C] except gamma /
 * beta / running_* ([C]).  scale == NULL: no normalisation (the first conv block has none) — then const_stats must be 1.  const_stats: the
 * affine map does not depend on y (eval-mode BatchNorm with running statistics, or no norm).  s2d: dz (modes 1, 2), out0 of mode 0 and out0 of
 * mode 2 are stored space-to-depth: logical pixel (y, x) of group cg at pixel (y/2, x/2) of group 4*cg + 2*(y&1) + (x&1) of a view with
 * H/2 x W/2 pixels — the input layout in which the next block's 4x4 stride-2 conv is a 3x3 stride-1 conv over 4x the channels.
 * `sums` targets must be zeroed by the caller (the kernels add); sums2 / sums3: completed results of mode 1 / mode 2. */
typedef struct {
    esr_act_view y;          /* the conv's output (pre-normalisation); defines B x C x H x W */
    esr_act_view dz, u;      /* gradient w.r.t. the activated output (modes 1, 2); cotangent of mode 1's result (mode 2) */
    esr_act_view out0, out1;
    int32_t B, groups, C;
    const float* scale; const float* shift; const float* mean; const float* rstd;      /* [groups][C] */
    const float* gamma;      /* [C] or NULL (ones) */
    const double* sums2; const double* sums3;
    float slope;             /* LeakyReLU negative slope; 1.0f = no activation */
    int32_t const_stats, s2d;
} esr_bn_desc;
int esr_bn_reduce(const esr_bn_desc* d, int mode, double* sums, esr_stream_t stream);
int esr_bn_apply(const esr_bn_desc* d, int mode, esr_stream_t stream);
/* mean / rstd / scale (= gamma*rstd) / shift (= beta - scale*mean) from mode-0 sums, group by group IN ORDER; running_mean / running_var
 * (may be NULL) are updated once per group as nn.BatchNorm2d does per call: r = (1-momentum)*r + momentum*stat, variance unbiased. */
int esr_bn_finalize(const double* sums, int groups, int C, int64_t n_per_group, float eps, float momentum, const float* gamma, const float* beta,
                    float* mean, float* rstd, float* scale, float* shift, float* running_mean, float* running_var, esr_stream_t stream);
/* dgamma[c] = sum_g sums2[g][c][1], dbeta[c] = sum_g sums2[g][c][0]; g_gamma (double backward; needs sums3, rstd) — any output may be NULL */
int esr_bn_param_grads(const double* sums2, const double* sums3, const float* rstd, int groups, int C, int64_t n_per_group, float* dgamma, float* dbeta,
                       float* g_gamma, esr_stream_t stream);

/* ---- launch lists: a whole pass of the generator with ONE call ----
 * The reference dispatches every layer from Python (RRDBNet.forward's module loop, codes/models/modules/architecture.py:278-302, and
 * autograd's node-by-node backward); a port that keeps one FFI call per launch pays ~19 us of host time for each of the ~1,100 launches
 * of a training step.  The launch plan of a pass is static per (network, shape, precision, buffer set): the caller records it ONCE as an
 * array of esr_cmd — each entry is the argument block of one of the entry points above — and replays it with esr_run; between replays it
 * only patches the few pointers that change from call to call (the NCHW input / output tensors).  esr_run enqueues the commands in
 * order on `stream` and stops at the first one that fails: the return value is that command's ESR_E_* code and *failed (if given) its
 * index; ESR_OK and -1 otherwise.  Nothing is copied or retained: descriptors are read during the call only. */
enum {
    ESR_OP_CONV3X3 = 1, ESR_OP_PACK_NCHW = 2, ESR_OP_UNPACK_GRAD_NCHW = 3, ESR_OP_ACT_COMBINE = 4, ESR_OP_PIXEL_UNSHUFFLE = 5,
    ESR_OP_GRAD_ABSMAX = 6, ESR_OP_GRAD_SCALE = 7, ESR_OP_WGRAD_BATCH_RUN = 8, ESR_OP_PACK_BATCH_RUN = 9, ESR_OP_ZERO = 10,
    ESR_OP_UNPACK_NCHW = 11, ESR_OP_WGRAD = 12, ESR_OP_BN_REDUCE = 13, ESR_OP_BN_APPLY = 14, ESR_OP_BN_FINALIZE = 15, ESR_OP_BN_PARAM_GRADS = 16,
    ESR_OP_BN_FINALIZE_APPLY = 17
};
typedef struct { const float* src; int64_t src_batch_stride; int32_t B, C, h, w, c0, nc, pad, down; esr_act_view dst; } esr_cmd_pack_nchw;
typedef struct { esr_act_view G; float* dst; int64_t dst_batch_stride; int32_t B, C, h, w, c0, nc, pad, down, accumulate; } esr_cmd_unpack_grad_nchw;
/* A / Bv / mask with hi == NULL are absent (NULL in esr_act_combine) */
typedef struct { esr_act_view A; float alpha; esr_act_view Bv; float beta; int32_t s; esr_act_view mask; float mask_slope; esr_act_view out; int32_t B; } esr_cmd_act_combine;
typedef struct { esr_act_view src; int32_t r; esr_act_view dst; int32_t B; } esr_cmd_pixel_unshuffle;
typedef struct { esr_act_view v; int32_t B; uint32_t* slot; } esr_cmd_grad_absmax;
typedef struct { esr_act_view src, dst; int32_t B; const uint32_t* slot; int32_t exp; const float* scale_in; const float* scale_den; float* scale_out; } esr_cmd_grad_scale;
typedef struct { const void* workspace; esr_wgrad_batch_plan plan; } esr_cmd_wgrad_batch_run;
typedef struct { const void* workspace; int32_t n; int64_t nblocks; } esr_cmd_pack_batch_run;
typedef struct { void* p; int64_t n16; } esr_cmd_zero;
typedef struct { esr_act_view src; int32_t B, nc; float* dst; } esr_cmd_unpack_nchw;
typedef struct { esr_bn_desc d; int32_t mode; double* sums; } esr_cmd_bn;                 /* esr_bn_reduce (sums) / esr_bn_apply */
typedef struct { const double* sums; int32_t groups, C; int64_t n_per_group; float eps, momentum; const float* gamma; const float* beta;
                 float* mean; float* rstd; float* scale; float* shift; float* running_mean; float* running_var; } esr_cmd_bn_finalize;
typedef struct { const double* sums2; const double* sums3; const float* rstd; int32_t groups, C; int64_t n_per_group; float* dgamma; float* dbeta;
                 float* g_gamma; } esr_cmd_bn_param_grads;
/* esr_bn_finalize(f...) and esr_bn_apply(d, 0) as ONE launch: the normalise + activate kernel derives its affine from the mode-0 sums itself
 * (same fp64 arithmetic) and its first block per channel group stores mean / rstd / scale / shift and moves the running statistics.  d->scale,
 * d->shift, d->mean, d->rstd are not read (f's are written); f->groups, f->C, f->n_per_group must agree with d.  Nine launches fewer per
 * forward of Discriminator_VGG_128 (codes/models/modules/architecture.py:446-508). */
typedef struct { esr_bn_desc d; esr_cmd_bn_finalize f; } esr_cmd_bn_finalize_apply;
int esr_bn_finalize_apply(const esr_bn_desc* d, const esr_cmd_bn_finalize* f, esr_stream_t stream);
typedef struct {
    int32_t op;              /* ESR_OP_* */
    int32_t reserved;
    union {
        esr_conv3x3_desc conv;
        esr_cmd_pack_nchw pack_nchw;
        esr_cmd_unpack_grad_nchw unpack_grad_nchw;
        esr_cmd_act_combine act_combine;
        esr_cmd_pixel_unshuffle pixel_unshuffle;
        esr_cmd_grad_absmax grad_absmax;
        esr_cmd_grad_scale grad_scale;
        esr_cmd_wgrad_batch_run wgrad_batch_run;
        esr_cmd_pack_batch_run pack_batch_run;
        esr_cmd_zero zero;
        esr_cmd_unpack_nchw unpack_nchw;
        esr_wgrad_desc wgrad;
        esr_cmd_bn bn;
        esr_cmd_bn_finalize bn_finalize;
        esr_cmd_bn_param_grads bn_param_grads;
        esr_cmd_bn_finalize_apply bn_finalize_apply;
    } u;
} esr_cmd;
int esr_run(const esr_cmd* cmds, int n, int* failed, esr_stream_t stream);
int64_t esr_cmd_bytes(void);       /* sizeof(esr_cmd): lets a binding check its struct layout */

int esr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ESR_HIP_H */
