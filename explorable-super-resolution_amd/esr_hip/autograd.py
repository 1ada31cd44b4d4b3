"""torch.autograd glue: the differentiable entry points of the HIP path.  Forward and backward are both HIP launches planned by
esr_hip/engine.py (generator) and esr_hip/cem_ops.py (CEM filters); torch only carries the graph."""
import numpy as np
import torch

from . import act as A
from . import cem_ops


# ------------------------------------------------------------------------------------------------ generator
class _RRDBFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, pad, x, *params):
        g, bufs = engine.run_forward(x, pad, keep=engine.keep_mode())
        ctx.engine, ctx.pad, ctx.bufs, ctx.x_shape = engine, pad, bufs, tuple(x.shape)
        ctx.params = params
        return g

    @staticmethod
    def backward(ctx, dg):
        need_dx = ctx.needs_input_grad[2]
        need_dw = any(ctx.needs_input_grad[3:])
        dx, dws = ctx.engine.run_backward(ctx.x_shape, ctx.pad, ctx.bufs, dg, need_dx=need_dx, need_dw=need_dw)
        grads = [None] * len(ctx.params)
        if need_dw:
            for i, p in enumerate(ctx.params):
                if ctx.needs_input_grad[3 + i]:
                    grads[i] = dws.get(p)
        return (None, None, dx) + tuple(grads)


def rrdb_forward_with_grad(engine, x, pad):
    params = engine.parameters()
    return _RRDBFunction.apply(engine, pad, x, *params)


# ------------------------------------------------------------------------------------------------ stand-alone conv (block-level calls)
class _ConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act_slope, split):
        y = A.conv3x3_nchw(x, weight, bias, act_slope, split)
        ctx.save_for_backward(x, weight, y)
        ctx.act_slope, ctx.split, ctx.has_bias = act_slope, split, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = dy * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, ctx.act_slope)) if ctx.act_slope != 1.0 else dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = A.conv3x3_dgrad_nchw(dy, weight, ctx.split)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = A.conv3x3_wgrad_nchw(dy, x, weight.shape, ctx.split)
            if not ctx.has_bias:
                db = None
        return dx, dw, db, None, None


def conv3x3_function(x, weight, bias, act_slope, split):
    return _ConvFunction.apply(x, weight, bias, act_slope, split)


# ------------------------------------------------------------------------------------------------ CEM filters
def tap_tables(taps):
    """[3][3][k][k] prefix / plain / suffix cumulative tap tables for esr_cem_adjoint (see csrc/esr_cem.hip)."""
    t = taps.detach().to(torch.float64)
    rows = [torch.cumsum(t, 0), t, torch.flip(torch.cumsum(torch.flip(t, [0]), 0), [0])]
    out = []
    for r in rows:
        out.append(torch.stack([torch.cumsum(r, 1), r, torch.flip(torch.cumsum(torch.flip(r, [1]), 1), [1])]))
    return torch.stack(out).to(torch.float32).contiguous()


def tap_tables_1d(t):
    """[3][k] prefix / plain / suffix sums of one 1-D factor of rank-one taps (esr_cem_adjoint_sep)."""
    t = t.detach().to(torch.float64)
    return torch.stack([torch.cumsum(t, 0), t, torch.flip(torch.cumsum(torch.flip(t, [0]), 0), [0])]).to(torch.float32).contiguous()


class AdjointTables:
    """What the adjoint of one CEM filter reads: the 2-D cumulative tables and, for rank-one taps (cem_ops._sep: the separable kernels'
    factors), the per-axis ones."""

    def __init__(self, taps):
        self.full = tap_tables(taps)
        self.k = taps.shape[-1]
        sep = cem_ops._sep(taps, self.full.device) if self.full.is_cuda else None
        self.v, self.h = (tap_tables_1d(sep[0]), tap_tables_1d(sep[1])) if sep is not None else (None, None)


def _tabs_for(taps):
    """The adjoint's cumulative tap tables, cached next to the taps' device copy ON the tensor that owns the storage (the frozen
    Filter_OP.weight; see cem_ops._taps_entry — a cache keyed on data pointers would go stale when the allocator recycles an address)."""
    owner = taps._base if taps._base is not None else taps
    key = ('tabs', owner._version, tuple(taps.shape), taps.storage_offset(), str(taps.device), cem_ops.USE_SEPARABLE)
    cache = getattr(owner, '_esr_taps', None)
    if cache is None:
        cache = {}
        try:
            owner._esr_taps = cache
        except Exception:
            return AdjointTables(taps)
    hit = cache.get(key)
    if hit is None:
        hit = cache[key] = AdjointTables(taps)
    return hit


class CemLinear(torch.autograd.Function):
    """One fixed CEM filter (kind: 'downscale' | 'lr_filter' | 'upscale'); backward = the adjoint gather."""

    @staticmethod
    def forward(ctx, x, taps, kind, sf, pre):
        ctx.kind, ctx.sf, ctx.pre, ctx.shape = kind, sf, pre, tuple(x.shape)
        ctx.tabs = _tabs_for(taps)
        if kind == 'downscale':
            return cem_ops.downscale_raw(x, taps, sf, pre)
        if kind == 'lr_filter':
            return cem_ops.lr_filter_raw(x, taps)
        return cem_ops.upscale_raw(x, taps, sf, pre)

    @staticmethod
    def backward(ctx, dy):
        return cem_ops.adjoint_raw(dy, ctx.tabs, ctx.kind, ctx.sf, ctx.pre, ctx.shape), None, None, None, None


def _pad_adjoint(dxp, m):
    """Adjoint of replicate padding by m (NCHW): fold the pad ring back onto the edge pixels."""
    if m == 0:
        return dxp
    d = dxp[:, :, m:-m, :].clone()
    d[:, :, 0] += dxp[:, :, :m].sum(2)
    d[:, :, -1] += dxp[:, :, -m:].sum(2)
    e = d[:, :, :, m:-m].clone()
    e[:, :, :, 0] += d[:, :, :, :m].sum(3)
    e[:, :, :, -1] += d[:, :, :, -m:].sum(3)
    return e


class _CemProject(torch.autograd.Function):
    """out = crop(g + U(K(pad(x) - D(g))))  — the plain CEM projection (reference CEMnet.py:303-311) as one node."""

    @staticmethod
    def forward(ctx, lr, g, taps_down, taps_inv, taps_up, sf, pre, lr_pad, crop):
        e = cem_ops.downscale_raw(g, taps_down, sf, pre, lr=lr, lr_pad=lr_pad)
        out = cem_ops.filter_upscale_raw(e, taps_inv, taps_up, sf, pre, g=g, crop=crop, mode=1)
        ctx.cfg = (sf, pre, lr_pad, crop, tuple(g.shape), tuple(e.shape))
        ctx.tabs = (_tabs_for(taps_down), _tabs_for(taps_inv), _tabs_for(taps_up))
        return out

    @staticmethod
    def backward(ctx, dout):
        sf, pre, lr_pad, crop, gshape, eshape = ctx.cfg
        td, ti, tu = ctx.tabs
        dfull = torch.nn.functional.pad(dout, (crop,) * 4) if crop else dout
        dfull = dfull.contiguous()
        df = cem_ops.adjoint_raw(dfull, tu, 'upscale', sf, pre, eshape)            # U^T
        de = cem_ops.adjoint_raw(df, ti, 'lr_filter', 1, 0, eshape)               # K^T
        dlr = _pad_adjoint(de, lr_pad) if ctx.needs_input_grad[0] else None
        dg = None
        if ctx.needs_input_grad[1]:
            dg = cem_ops.adjoint_raw(de, td, 'downscale', sf, pre, gshape, base=dfull, alpha=-1.0)   # g enters directly and through -D
        return dlr, dg, None, None, None, None, None, None, None


def cem_project_with_grad(lr, g, taps_down, taps_inv, taps_up, sf, pre, lr_pad, crop, sigmoid_range, decomposed):
    if sigmoid_range is None and not decomposed:
        return _CemProject.apply(lr, g, taps_down, taps_inv, taps_up, sf, pre, lr_pad, crop)
    # option variants: composed from the differentiable filters (elementwise glue stays in torch)
    lr_p = torch.nn.functional.pad(lr, (lr_pad,) * 4, mode='replicate') if lr_pad else lr
    ortho_x = CemLinear.apply(CemLinear.apply(lr_p, taps_inv, 'lr_filter', 1, 0), taps_up, 'upscale', sf, pre)
    dg = CemLinear.apply(g, taps_down, 'downscale', sf, pre)
    ns = g - CemLinear.apply(CemLinear.apply(dg, taps_inv, 'lr_filter', 1, 0), taps_up, 'upscale', sf, pre)
    if sigmoid_range is not None:
        ns = torch.tanh(ns) * sigmoid_range
    if decomposed:
        return [ortho_x, ns]
    out = ortho_x + ns
    return out[:, :, crop:out.shape[2] - crop, crop:out.shape[3] - crop] if crop else out
