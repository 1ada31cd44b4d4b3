"""Host wrappers of the Z-objective kernels (csrc/esr_zobj.hip): the soft histogram behind the reference's SoftHistogramLoss
(codes/Z_optimization.py:170-209), as a differentiable torch function."""
import ctypes as C

import torch

from . import _lib
from ._lib import check
from .act import require_gpu, stream_ptr


class _SoftHist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, K, lo, hi, T, eps):
        require_gpu(v, 'image values')
        v = v.detach().reshape(-1).float().contiguous()
        n = v.numel()
        slabs = int(_lib.lib.esr_soft_hist_slabs(n))
        partial = torch.empty(slabs, K, dtype=torch.float64, device=v.device)
        check(_lib.lib.esr_soft_hist_fwd(v.data_ptr(), n, K, lo, hi, T, eps, partial.data_ptr(), stream_ptr()), 'esr_soft_hist_fwd')
        ctx.save_for_backward(v)
        ctx.args = (K, lo, hi, T, eps)
        return partial.sum(0) / n                        # [K] float64, like the reference's

    @staticmethod
    def backward(ctx, gh):
        v, = ctx.saved_tensors
        K, lo, hi, T, eps = ctx.args
        g = (gh.double() / v.numel()).float().contiguous()
        gv = torch.empty_like(v)
        check(_lib.lib.esr_soft_hist_bwd(v.data_ptr(), v.numel(), K, lo, hi, T, eps, g.data_ptr(), gv.data_ptr(), stream_ptr()), 'esr_soft_hist_bwd')
        return gv, None, None, None, None, None


def soft_histogram(values, bins, lo, hi, temperature, eps=1e-7):
    """h[k] = mean_i exp(-(d(v_i, c_k) + eps)^2 / temperature), c_k = linspace(lo, hi, bins), distance wrapped with period hi.
    values: any shape (flattened), on the GPU; returns [bins] float64; differentiable w.r.t. values."""
    shape_grad = values
    out = _SoftHist.apply(values.reshape(-1), int(bins), float(lo), float(hi), float(temperature), float(eps))
    return out
