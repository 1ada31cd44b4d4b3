"""Loads libesr_hip.so (built in-tree by csrc/Makefile) and declares the C-ABI of include/esr_hip.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = 'libesr_hip.so'


class EsrError(RuntimeError):
    pass


class ActView(C.Structure):
    """esr_act_view (include/esr_hip.h)."""
    _fields_ = [('hi', C.c_void_p), ('lo', C.c_void_p), ('ncg', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('batch_stride', C.c_int64), ('cg_stride', C.c_int64), ('fmt', C.c_int32)]


class Conv3x3Desc(C.Structure):
    """esr_conv3x3_desc (include/esr_hip.h)."""
    _fields_ = [('in0', ActView), ('in1', ActView), ('upsample', C.c_int32), ('wpack', C.c_void_p), ('bias', C.c_void_p),
                ('cout', C.c_int32), ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('act_slope', C.c_float), ('alpha', C.c_float),
                ('res1', ActView), ('beta1', C.c_float), ('res2', ActView), ('beta2', C.c_float),
                ('out', ActView), ('out2', ActView), ('out_nchw', C.c_void_p),
                ('mask_src', ActView), ('mask_cg0', C.c_int32), ('mask_cg1', C.c_int32), ('mask_slope', C.c_float),
                ('reverse_order', C.c_int32), ('weight_planes', C.c_int32), ('in1_lo_groups', C.c_int32),
                ('pixel_shuffle', C.c_int32), ('ps_rowgroup0', C.c_int32), ('tap_mask_k', C.c_int32 * 4), ('tap_mask_k_shift', C.c_int32),
                ('tap_mask_m', C.c_int32 * 4), ('k_split_ws', C.c_void_p), ('k_split_ws_floats', C.c_int64), ('lds_stages', C.c_int32),
                ('range_flag', C.c_void_p), ('range_tag', C.c_uint32)]


class WgradDesc(C.Structure):
    """esr_wgrad_desc (include/esr_hip.h)."""
    _fields_ = [('dy', ActView), ('x', ActView), ('xlat', ActView), ('lat', C.c_int32), ('upsample', C.c_int32), ('cout', C.c_int32),
                ('cin_main', C.c_int32), ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('alpha', C.c_float), ('dw', C.c_void_p),
                ('db', C.c_void_p), ('workspace', C.c_void_p), ('workspace_floats', C.c_int64), ('tap_masks', C.c_int32 * 4)]


class WgradBatchPlan(C.Structure):
    """esr_wgrad_batch_plan (include/esr_hip.h)."""
    _fields_ = [('nwg', C.c_int64), ('table_bytes', C.c_int64), ('n', C.c_int32), ('max_red', C.c_int32), ('split', C.c_int32), ('f16', C.c_int32), ('s2d', C.c_int32), ('reserved', C.c_int32)]


class PackDesc(C.Structure):
    """esr_pack_desc (include/esr_hip.h)."""
    _fields_ = [('w', C.c_void_p), ('cout_w', C.c_int32), ('cin_w', C.c_int32), ('kmap', C.c_void_p), ('ncg_in', C.c_int32), ('mmap', C.c_void_p),
                ('mtiles', C.c_int32), ('transposed', C.c_int32), ('split', C.c_int32), ('scale', C.c_float), ('wpack', C.c_void_p)]


class BnDesc(C.Structure):
    """esr_bn_desc (include/esr_hip.h)."""
    _fields_ = [('y', ActView), ('dz', ActView), ('u', ActView), ('out0', ActView), ('out1', ActView), ('B', C.c_int32), ('groups', C.c_int32),
                ('C', C.c_int32), ('scale', C.c_void_p), ('shift', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p), ('gamma', C.c_void_p),
                ('sums2', C.c_void_p), ('sums3', C.c_void_p), ('slope', C.c_float), ('const_stats', C.c_int32), ('s2d', C.c_int32)]


class AdamTensor(C.Structure):
    """esr_adam_tensor (include/esr_hip.h)."""
    _fields_ = [('p', C.c_void_p), ('g', C.c_void_p), ('m', C.c_void_p), ('v', C.c_void_p), ('n', C.c_int64)]


# ---- launch lists (esr_cmd / esr_run, include/esr_hip.h)
OP_CONV3X3, OP_PACK_NCHW, OP_UNPACK_GRAD_NCHW, OP_ACT_COMBINE, OP_PIXEL_UNSHUFFLE, OP_GRAD_ABSMAX, OP_GRAD_SCALE, OP_WGRAD_BATCH_RUN, \
    OP_PACK_BATCH_RUN, OP_ZERO, OP_UNPACK_NCHW, OP_WGRAD, OP_BN_REDUCE, OP_BN_APPLY, OP_BN_FINALIZE, OP_BN_PARAM_GRADS, \
    OP_BN_FINALIZE_APPLY = range(1, 18)


class CmdPackNchw(C.Structure):
    _fields_ = [('src', C.c_void_p), ('src_batch_stride', C.c_int64), ('B', C.c_int32), ('C', C.c_int32), ('h', C.c_int32), ('w', C.c_int32),
                ('c0', C.c_int32), ('nc', C.c_int32), ('pad', C.c_int32), ('down', C.c_int32), ('dst', ActView)]


class CmdUnpackGradNchw(C.Structure):
    _fields_ = [('G', ActView), ('dst', C.c_void_p), ('dst_batch_stride', C.c_int64), ('B', C.c_int32), ('C', C.c_int32), ('h', C.c_int32),
                ('w', C.c_int32), ('c0', C.c_int32), ('nc', C.c_int32), ('pad', C.c_int32), ('down', C.c_int32), ('accumulate', C.c_int32)]


class CmdActCombine(C.Structure):
    _fields_ = [('A', ActView), ('alpha', C.c_float), ('Bv', ActView), ('beta', C.c_float), ('s', C.c_int32), ('mask', ActView),
                ('mask_slope', C.c_float), ('out', ActView), ('B', C.c_int32)]


class CmdPixelUnshuffle(C.Structure):
    _fields_ = [('src', ActView), ('r', C.c_int32), ('dst', ActView), ('B', C.c_int32)]


class CmdGradAbsmax(C.Structure):
    _fields_ = [('v', ActView), ('B', C.c_int32), ('slot', C.c_void_p)]


class CmdGradScale(C.Structure):
    _fields_ = [('src', ActView), ('dst', ActView), ('B', C.c_int32), ('slot', C.c_void_p), ('exp', C.c_int32), ('scale_in', C.c_void_p),
                ('scale_den', C.c_void_p), ('scale_out', C.c_void_p)]


class CmdWgradBatchRun(C.Structure):
    _fields_ = [('workspace', C.c_void_p), ('plan', WgradBatchPlan)]


class CmdPackBatchRun(C.Structure):
    _fields_ = [('workspace', C.c_void_p), ('n', C.c_int32), ('nblocks', C.c_int64)]


class CmdZero(C.Structure):
    _fields_ = [('p', C.c_void_p), ('n16', C.c_int64)]


class CmdUnpackNchw(C.Structure):
    _fields_ = [('src', ActView), ('B', C.c_int32), ('nc', C.c_int32), ('dst', C.c_void_p)]


class CmdBn(C.Structure):
    _fields_ = [('d', BnDesc), ('mode', C.c_int32), ('sums', C.c_void_p)]


class CmdBnFinalize(C.Structure):
    _fields_ = [('sums', C.c_void_p), ('groups', C.c_int32), ('C', C.c_int32), ('n_per_group', C.c_int64), ('eps', C.c_float), ('momentum', C.c_float),
                ('gamma', C.c_void_p), ('beta', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('running_mean', C.c_void_p), ('running_var', C.c_void_p)]


class CmdBnFinalizeApply(C.Structure):
    _fields_ = [('d', BnDesc), ('f', CmdBnFinalize)]


class CmdBnParamGrads(C.Structure):
    _fields_ = [('sums2', C.c_void_p), ('sums3', C.c_void_p), ('rstd', C.c_void_p), ('groups', C.c_int32), ('C', C.c_int32), ('n_per_group', C.c_int64),
                ('dgamma', C.c_void_p), ('dbeta', C.c_void_p), ('g_gamma', C.c_void_p)]


class CmdUnion(C.Union):
    _fields_ = [('conv', Conv3x3Desc), ('pack_nchw', CmdPackNchw), ('unpack_grad_nchw', CmdUnpackGradNchw), ('act_combine', CmdActCombine),
                ('pixel_unshuffle', CmdPixelUnshuffle), ('grad_absmax', CmdGradAbsmax), ('grad_scale', CmdGradScale),
                ('wgrad_batch_run', CmdWgradBatchRun), ('pack_batch_run', CmdPackBatchRun), ('zero', CmdZero), ('unpack_nchw', CmdUnpackNchw),
                ('wgrad', WgradDesc), ('bn', CmdBn), ('bn_finalize', CmdBnFinalize), ('bn_param_grads', CmdBnParamGrads), ('bn_finalize_apply', CmdBnFinalizeApply)]


class Cmd(C.Structure):
    """esr_cmd (include/esr_hip.h)."""
    _fields_ = [('op', C.c_int32), ('reserved', C.c_int32), ('u', CmdUnion)]


CMD_MEMBER = {OP_CONV3X3: 'conv', OP_PACK_NCHW: 'pack_nchw', OP_UNPACK_GRAD_NCHW: 'unpack_grad_nchw', OP_ACT_COMBINE: 'act_combine',
              OP_PIXEL_UNSHUFFLE: 'pixel_unshuffle', OP_GRAD_ABSMAX: 'grad_absmax', OP_GRAD_SCALE: 'grad_scale',
              OP_WGRAD_BATCH_RUN: 'wgrad_batch_run', OP_PACK_BATCH_RUN: 'pack_batch_run', OP_ZERO: 'zero', OP_UNPACK_NCHW: 'unpack_nchw',
              OP_WGRAD: 'wgrad', OP_BN_REDUCE: 'bn', OP_BN_APPLY: 'bn', OP_BN_FINALIZE: 'bn_finalize', OP_BN_PARAM_GRADS: 'bn_param_grads', OP_BN_FINALIZE_APPLY: 'bn_finalize_apply'}


_SIGS = {
    'esr_bn_reduce': (C.c_int, [C.POINTER(BnDesc), C.c_int, C.c_void_p, C.c_void_p]),
    'esr_bn_apply': (C.c_int, [C.POINTER(BnDesc), C.c_int, C.c_void_p]),
    'esr_bn_finalize_apply': (C.c_int, [C.POINTER(BnDesc), C.POINTER(CmdBnFinalize), C.c_void_p]),
    'esr_bn_finalize': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'esr_bn_param_grads': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'esr_adam_workspace_bytes': (C.c_int64, [C.POINTER(AdamTensor), C.c_int]),
    'esr_adam_upload': (C.c_int64, [C.POINTER(AdamTensor), C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    'esr_adam_table': (C.c_int64, [C.POINTER(AdamTensor), C.c_int, C.c_void_p, C.c_int64]),
    'esr_adam_run': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'esr_run': (C.c_int, [C.POINTER(Cmd), C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    'esr_cmd_bytes': (C.c_int64, []),
    'esr_version': (C.c_int, []),
    'esr_conv3x3': (C.c_int, [C.POINTER(Conv3x3Desc), C.c_void_p]),
    'esr_pixel_unshuffle': (C.c_int, [C.POINTER(ActView), C.c_int, C.POINTER(ActView), C.c_int, C.c_void_p]),
    'esr_conv_wpack_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'esr_pack_conv_weights': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.c_void_p, C.c_void_p]),
    'esr_pack_batch_workspace_bytes': (C.c_int64, [C.POINTER(PackDesc), C.c_int]),
    'esr_pack_batch_upload': (C.c_int64, [C.POINTER(PackDesc), C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    'esr_pack_batch_run': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    'esr_pack_nchw': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.POINTER(ActView), C.c_void_p]),
    'esr_unpack_nchw': (C.c_int, [C.POINTER(ActView), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'esr_zero': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    'esr_act_combine': (C.c_int, [C.POINTER(ActView), C.c_float, C.POINTER(ActView), C.c_float, C.c_int, C.POINTER(ActView), C.c_float,
                                  C.POINTER(ActView), C.c_int, C.c_void_p]),
    'esr_grad_absmax': (C.c_int, [C.POINTER(ActView), C.c_int, C.c_void_p, C.c_void_p]),
    'esr_grad_scale': (C.c_int, [C.POINTER(ActView), C.POINTER(ActView), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'esr_unpack_grad_nchw': (C.c_int, [C.POINTER(ActView), C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p]),
    'esr_cem_adjoint': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'esr_cem_adjoint_sep': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    'esr_conv3x3_wgrad': (C.c_int, [C.POINTER(WgradDesc), C.c_void_p]),
    'esr_conv3x3_wgrad_workspace_floats': (C.c_int64, [C.POINTER(WgradDesc)]),
    'esr_conv3x3_wgrad_batch_workspace_bytes': (C.c_int64, [C.POINTER(WgradDesc), C.c_int]),
    'esr_conv3x3_wgrad_batch': (C.c_int, [C.POINTER(WgradDesc), C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    'esr_conv3x3_wgrad_batch_upload': (C.c_int, [C.POINTER(WgradDesc), C.c_int, C.c_void_p, C.c_int64, C.POINTER(WgradBatchPlan), C.c_void_p]),
    'esr_conv3x3_wgrad_batch_unit': (C.c_int64, [C.POINTER(WgradDesc), C.c_int]),
    'esr_conv3x3_wgrad_batch_part_workspace_bytes': (C.c_int64, [C.POINTER(WgradDesc), C.c_int, C.c_int64]),
    'esr_conv3x3_wgrad_batch_part_upload': (C.c_int, [C.POINTER(WgradDesc), C.c_int, C.c_void_p, C.c_int64, C.POINTER(WgradBatchPlan), C.c_int64, C.c_void_p]),
    'esr_conv3x3_wgrad_batch_run': (C.c_int, [C.c_void_p, C.POINTER(WgradBatchPlan), C.c_void_p]),
    'esr_conv3x3_wgrad_batch_run_side': (C.c_int, [C.c_void_p, C.POINTER(WgradBatchPlan), C.c_void_p]),
    'esr_conv3x3_wgrad_side_occupancy': (C.c_int, [C.c_int]),
    'esr_conv3x3_wgrad_batch_rebase': (C.c_int, [C.c_void_p, C.POINTER(WgradBatchPlan), C.c_int64, C.c_void_p]),
    'esr_soft_hist_slabs': (C.c_int64, [C.c_int64]),
    'esr_soft_hist_fwd': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'esr_soft_hist_bwd': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'esr_img_stats': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'esr_img_stats_grad': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'esr_cem_downscale': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'esr_cem_lrfilter': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'esr_cem_upscale': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'esr_cem_downscale_sep': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'esr_cem_lrfilter_sep': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'esr_cem_upscale_sep': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'esr_cem_filter_upscale_sep': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'esr_cem_sep_form': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def library_path():
    # ESR_HIP_LIBRARY: an alternative build of the same C-ABI (e.g. an instrumented one); still no fallback if it is missing
    return os.environ.get('ESR_HIP_LIBRARY') or os.path.join(_HERE, _LIB_NAME)


def load_library():
    """Load the in-tree shared library; raises EsrError (never falls back) when it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7; whichever copy of that SONAME is loaded first serves the whole process, and device
    # pointers / streams are only meaningful inside ONE runtime instance — so torch's must be in place before ours is resolved
    import torch  # noqa: F401
    path = library_path()
    if not os.path.exists(path):
        raise EsrError('%s not found: build it with `make -C explorable-super-resolution_amd/csrc` (or __graft_entry__.build()). '
                       'There is no CPU fallback for the RRDB/CEM kernels.' % path)
    try:
        h = C.CDLL(path)
    except OSError as e:
        raise EsrError('cannot load %s: %s' % (path, e))
    for name, (res, args) in _SIGS.items():
        try:
            f = getattr(h, name)
        except AttributeError:
            raise EsrError('%s does not export %s (stale build?)' % (path, name))
        f.restype = res
        f.argtypes = args
    if h.esr_cmd_bytes() != C.sizeof(Cmd):
        raise EsrError('%s: esr_cmd is %d bytes in the library, %d in this binding (stale build?)' % (path, h.esr_cmd_bytes(), C.sizeof(Cmd)))
    _lib = h
    return h


class _LazyLib:
    def __getattr__(self, name):
        return getattr(load_library(), name)


lib = _LazyLib()

ESR_OK, ESR_E_ARG, ESR_E_UNSUPPORTED, ESR_E_LAUNCH = 0, -1, -2, -3
_ERR = {-1: 'ESR_E_ARG (bad argument)', -2: 'ESR_E_UNSUPPORTED (unsupported shape)', -3: 'ESR_E_LAUNCH (HIP launch error)'}


def check(rc, what):
    if rc != 0:
        raise EsrError('%s failed: %s' % (what, _ERR.get(rc, rc)))
