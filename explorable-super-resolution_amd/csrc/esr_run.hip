// Launch lists: a whole forward / backward pass of the generator replayed with ONE C-ABI call (esr_run, include/esr_hip.h).
// Host code only: every command dispatches to the entry point of the same name; what this file removes is the per-launch trip through
// the caller's FFI (ctypes: ~19 us per call, ~1,100 calls per training step).
#include "esr_common.h"

extern "C" int esr_run(const esr_cmd* cmds, int n, int* failed, esr_stream_t stream) {
    if (failed) *failed = -1;
    if (n < 0 || (n > 0 && !cmds)) return ESR_E_ARG;
    for (int i = 0; i < n; ++i) {
        const esr_cmd& c = cmds[i];
        int rc;
        switch (c.op) {
            case ESR_OP_CONV3X3: rc = esr_conv3x3(&c.u.conv, stream); break;
            case ESR_OP_PACK_NCHW: {
                const esr_cmd_pack_nchw& a = c.u.pack_nchw;
                rc = esr_pack_nchw(a.src, a.src_batch_stride, a.B, a.C, a.h, a.w, a.c0, a.nc, a.pad, a.down, &a.dst, stream);
            } break;
            case ESR_OP_UNPACK_GRAD_NCHW: {
                const esr_cmd_unpack_grad_nchw& a = c.u.unpack_grad_nchw;
                rc = esr_unpack_grad_nchw(&a.G, a.dst, a.dst_batch_stride, a.B, a.C, a.h, a.w, a.c0, a.nc, a.pad, a.down, a.accumulate, stream);
            } break;
            case ESR_OP_ACT_COMBINE: {
                const esr_cmd_act_combine& a = c.u.act_combine;
                rc = esr_act_combine(a.A.hi ? &a.A : nullptr, a.alpha, a.Bv.hi ? &a.Bv : nullptr, a.beta, a.s, a.mask.hi ? &a.mask : nullptr, a.mask_slope,
                                     &a.out, a.B, stream);
            } break;
            case ESR_OP_PIXEL_UNSHUFFLE: rc = esr_pixel_unshuffle(&c.u.pixel_unshuffle.src, c.u.pixel_unshuffle.r, &c.u.pixel_unshuffle.dst, c.u.pixel_unshuffle.B, stream); break;
            case ESR_OP_GRAD_ABSMAX: rc = esr_grad_absmax(&c.u.grad_absmax.v, c.u.grad_absmax.B, c.u.grad_absmax.slot, stream); break;
            case ESR_OP_GRAD_SCALE: {
                const esr_cmd_grad_scale& a = c.u.grad_scale;
                rc = esr_grad_scale(&a.src, &a.dst, a.B, a.slot, a.exp, a.scale_in, a.scale_den, a.scale_out, stream);
            } break;
            case ESR_OP_WGRAD_BATCH_RUN: rc = esr_conv3x3_wgrad_batch_run(c.u.wgrad_batch_run.workspace, &c.u.wgrad_batch_run.plan, stream); break;
            case ESR_OP_PACK_BATCH_RUN: rc = esr_pack_batch_run(c.u.pack_batch_run.workspace, c.u.pack_batch_run.n, c.u.pack_batch_run.nblocks, stream); break;
            case ESR_OP_ZERO: rc = esr_zero(c.u.zero.p, c.u.zero.n16, stream); break;
            case ESR_OP_UNPACK_NCHW: rc = esr_unpack_nchw(&c.u.unpack_nchw.src, c.u.unpack_nchw.B, c.u.unpack_nchw.nc, c.u.unpack_nchw.dst, stream); break;
            case ESR_OP_WGRAD: rc = esr_conv3x3_wgrad(&c.u.wgrad, stream); break;
            case ESR_OP_BN_REDUCE: rc = esr_bn_reduce(&c.u.bn.d, c.u.bn.mode, c.u.bn.sums, stream); break;
            case ESR_OP_BN_APPLY: rc = esr_bn_apply(&c.u.bn.d, c.u.bn.mode, stream); break;
            case ESR_OP_BN_FINALIZE: {
                const esr_cmd_bn_finalize& a = c.u.bn_finalize;
                rc = esr_bn_finalize(a.sums, a.groups, a.C, a.n_per_group, a.eps, a.momentum, a.gamma, a.beta, a.mean, a.rstd, a.scale, a.shift, a.running_mean,
                                     a.running_var, stream);
            } break;
            case ESR_OP_BN_PARAM_GRADS: {
                const esr_cmd_bn_param_grads& a = c.u.bn_param_grads;
                rc = esr_bn_param_grads(a.sums2, a.sums3, a.rstd, a.groups, a.C, a.n_per_group, a.dgamma, a.dbeta, a.g_gamma, stream);
            } break;
            case ESR_OP_BN_FINALIZE_APPLY: rc = esr_bn_finalize_apply(&c.u.bn_finalize_apply.d, &c.u.bn_finalize_apply.f, stream); break;
            default: rc = ESR_E_ARG;
        }
        if (rc != ESR_OK) {
            if (failed) *failed = i;
            return rc;
        }
    }
    return ESR_OK;
}

extern "C" int64_t esr_cmd_bytes(void) { return (int64_t)sizeof(esr_cmd); }
