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


# ------------------------------------------------------------------------------------------------ per-image statistics (esr_img_stats)
class _ImgStat(torch.autograd.Function):
    """kind 0: unbiased std per image; kind 1: TV_Loss per image; kind 2: structure tensor [3, B] — of v = clamp(x, 0, 1) * mask as asked.
    One reduction kernel forward, one element-wise kernel backward; the scalar algebra in between runs on [B]-sized tensors."""

    @staticmethod
    def forward(ctx, x, mask, clamp01, kind):
        require_gpu(x, 'image batch')
        xd = x.detach()
        if xd.dtype != torch.float32 or not xd.is_contiguous():
            xd = xd.float().contiguous()
        B, Cc, H, W = xd.shape
        m = None
        if mask is not None:
            m = mask.detach().to(device=xd.device, dtype=torch.float32).expand(H, W).contiguous()
        sums = torch.zeros(B, 3, dtype=torch.float64, device=xd.device)
        check(_lib.lib.esr_img_stats(xd.data_ptr(), B, Cc, H, W, m.data_ptr() if m is not None else None, 1 if clamp01 else 0, kind, sums.data_ptr(), stream_ptr()),
              'esr_img_stats')
        ctx.save_for_backward(xd, m, sums)
        ctx.clamp01, ctx.kind = clamp01, kind
        n = Cc * H * W
        if kind == 0:
            mean = sums[:, 0] / n
            var = (sums[:, 1] - sums[:, 0] * mean) / (n - 1)
            return torch.sqrt(var.clamp_min(0)).float()
        if kind == 1:
            return (sums[:, 0] / (Cc * H * (W - 1)) + sums[:, 1] / (Cc * (H - 1) * W)).float()
        return (sums / (Cc * (H - 1) * (W - 1))).t().float().contiguous()             # [3, B]

    @staticmethod
    def backward(ctx, g):
        xd, m, sums = ctx.saved_tensors
        B, Cc, H, W = xd.shape
        n = Cc * H * W
        g = g.detach().double()
        coef = torch.zeros(B, 3, dtype=torch.float64, device=xd.device)
        if ctx.kind == 0:
            mean = sums[:, 0] / n
            std = torch.sqrt(((sums[:, 1] - sums[:, 0] * mean) / (n - 1)).clamp_min(0))
            c0 = g / ((n - 1) * std)
            coef[:, 0], coef[:, 1] = c0, -mean * c0
        elif ctx.kind == 1:
            coef[:, 0], coef[:, 1] = g / (Cc * H * (W - 1)), g / (Cc * (H - 1) * W)
        else:
            nn_ = Cc * (H - 1) * (W - 1)
            coef[:, 0], coef[:, 1], coef[:, 2] = 2 * g[0] / nn_, 2 * g[1] / nn_, g[2] / nn_
        coef = coef.float().contiguous()
        dx = torch.empty_like(xd)
        check(_lib.lib.esr_img_stats_grad(xd.data_ptr(), B, Cc, H, W, m.data_ptr() if m is not None else None, 1 if ctx.clamp01 else 0, ctx.kind, coef.data_ptr(),
                                          dx.data_ptr(), 0, stream_ptr()), 'esr_img_stats_grad')
        return dx, None, None, None


def image_std(x, mask=None, clamp01=False):
    """torch.std(clamp(x, 0, 1) * mask, dim=(1, 2, 3)) per image (reference Z_optimization.py:383-388, Masked_STD) -> [B]"""
    return _ImgStat.apply(x, mask, clamp01, 0)


def tv_loss(x, mask=None, clamp01=False):
    """TV_Loss of clamp(x, 0, 1) * mask per image (reference Z_optimization.py:324-326) -> [B]"""
    return _ImgStat.apply(x, mask, clamp01, 1)


def structure_tensor(x):
    """[mean ix^2, mean iy^2, mean ix iy] per image from forward differences on the (H-1) x (W-1) frame (reference loss.py:49-62,141-151) -> [3, B]"""
    return _ImgStat.apply(x, None, False, 2)
