"""Host wrappers of the CEM kernels (csrc/esr_cem.hip).  fp32 NCHW in/out.  Differentiable: every op is linear with
fixed taps, so its backward is the adjoint filter (esr_hip/autograd.py)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check
from .act import require_gpu, stream_ptr


def _prep(x, what):
    require_gpu(x, what)
    x = x if x.dtype == torch.float32 else x.float()
    return x.contiguous()


USE_SEPARABLE = True          # module attribute: the tests compare the separable kernels with the 2-D ones by clearing it
# project(): batches whose generator output exceeds PROJECT_CHUNK_BYTES run the three kernels per chunk of this many images (0: never).
# OFF by default — measured on MI355X at configs[1] (tools/experiments/cem_time.py, profiles/r04_cem_chunk{0,8}_pmc.json): the whole batch
# 0.225 ms, chunks of 16 / 8 / 4 images 0.267 / 0.283 / 0.370 ms, and the counters see the same 0.64 GB per projection either way (FETCH_SIZE
# counts L2 misses; what the Infinity Cache absorbs is invisible to it) — four times the launches of kernels that are 30-100 us each cost more
# than the second reading of `g` from HBM.  Kept as a switch (results are bit-identical, tests/test_gpu_parity.py).
PROJECT_CHUNK_IMAGES = 0
PROJECT_CHUNK_BYTES = 64 << 20


def _taps_entry(t, dev):
    """Device copy of a 2-D tap array and, when it is rank one (sigma_2 <= 1e-6 sigma_1: the bicubic ds_kernel and its inv_hTh, SURVEY.md
    7.3), its 1-D factors taps = outer(tv, th) for the separable kernels.  The taps are construction-time constants (a view of the frozen
    Filter_OP.weight): the entry is cached ON the tensor that owns the storage (the Parameter), keyed by its version counter and the target
    device, so steady-state calls do no host work and no host->device copy — and the cache dies with the module (a cache keyed by data
    pointers would be handed stale taps when the allocator recycles a freed module's storage)."""
    owner = t._base if t._base is not None else t          # taps() hands out a view of the frozen Filter_OP.weight Parameter
    t = t.detach()
    key = (owner._version, tuple(t.shape), t.storage_offset(), tuple(t.stride()), str(dev))
    cache = getattr(owner, '_esr_taps', None)
    if cache is None:
        cache = {}
        try:
            owner._esr_taps = cache
        except Exception:          # a tensor type that takes no attributes: no caching
            pass
    hit = cache.get(key)
    if hit is None:
        assert t.dim() == 2 and t.shape[0] == t.shape[1] and t.shape[0] % 2 == 1, 'CEM filters are odd square 2-D arrays'
        d = t.to(device=dev, dtype=torch.float32).contiguous().clone()
        a = t.double().cpu().numpy()
        u, sv, vt = np.linalg.svd(a)
        sep = None
        if sv[0] > 0 and (len(sv) == 1 or sv[1] <= 1e-6 * sv[0]):
            sgn = 1.0 if u[:, 0].sum() >= 0 else -1.0
            tv, th = sgn * u[:, 0] * np.sqrt(sv[0]), sgn * vt[0] * np.sqrt(sv[0])
            sep = (torch.tensor(tv, dtype=torch.float32, device=dev), torch.tensor(th, dtype=torch.float32, device=dev))
        if len(cache) > 8:
            cache.clear()
        hit = cache[key] = (d, sep)
    return hit


def _taps(t, dev):
    return _taps_entry(t, dev)[0]


def _sep(t, dev):
    return _taps_entry(t, dev)[1] if USE_SEPARABLE else None


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in ts)


def downscale_raw(y, taps, sf, pre, lr=None, lr_pad=0):
    y = _prep(y, 'HR image')
    sep, taps = _sep(taps, y.device), _taps(taps, y.device)
    B, Cc, Hh, Wh = y.shape
    assert Hh % sf == 0 and Wh % sf == 0, 'HR size must be divisible by the scale factor'
    h, w = Hh // sf, Wh // sf
    out = torch.empty(B, Cc, h, w, dtype=torch.float32, device=y.device)
    if lr is not None:
        lr = _prep(lr, 'LR image')
        assert lr.shape == (B, Cc, h - 2 * lr_pad, w - 2 * lr_pad), 'LR / HR sizes do not match'
    rc = _lib.ESR_E_UNSUPPORTED
    if sep is not None:
        rc = _lib.lib.esr_cem_downscale_sep(y.data_ptr(), B, Cc, h, w, sf, pre, sep[0].data_ptr(), sep[1].data_ptr(), taps.shape[0],
                                            lr.data_ptr() if lr is not None else None, lr_pad, out.data_ptr(), stream_ptr())
    if rc == _lib.ESR_E_UNSUPPORTED:
        rc = _lib.lib.esr_cem_downscale(y.data_ptr(), B, Cc, h, w, sf, pre, taps.data_ptr(), taps.shape[0],
                                        lr.data_ptr() if lr is not None else None, lr_pad, out.data_ptr(), stream_ptr())
    check(rc, 'esr_cem_downscale')
    return out


def lr_filter_raw(x, taps):
    x = _prep(x, 'LR image')
    sep, taps = _sep(taps, x.device), _taps(taps, x.device)
    B, Cc, h, w = x.shape
    out = torch.empty_like(x)
    rc = _lib.ESR_E_UNSUPPORTED
    if sep is not None:
        rc = _lib.lib.esr_cem_lrfilter_sep(x.data_ptr(), B, Cc, h, w, sep[0].data_ptr(), sep[1].data_ptr(), taps.shape[0], out.data_ptr(), stream_ptr())
    if rc == _lib.ESR_E_UNSUPPORTED:
        rc = _lib.lib.esr_cem_lrfilter(x.data_ptr(), B, Cc, h, w, taps.data_ptr(), taps.shape[0], out.data_ptr(), stream_ptr())
    check(rc, 'esr_cem_lrfilter')
    return out


def upscale_raw(f, taps, sf, pre, f2=None, g=None, crop=0, mode=0, rng=0.0, out=None):
    f = _prep(f, 'LR image')
    sep, taps = _sep(taps, f.device), _taps(taps, f.device)
    B, Cc, h, w = f.shape
    Ho, Wo = sf * h - 2 * crop, sf * w - 2 * crop
    if out is None:
        out = torch.empty(B, Cc, Ho, Wo, dtype=torch.float32, device=f.device)
    assert out.shape == (B, Cc, Ho, Wo) and out.dtype == torch.float32 and out.is_contiguous()
    out2 = torch.empty_like(out) if mode == 3 else None
    if f2 is not None:
        f2 = _prep(f2, 'LR image')
    if g is not None:
        g = _prep(g, 'generated image')
        assert g.shape == (B, Cc, sf * h, sf * w)
    tail = (g.data_ptr() if g is not None else None, crop, mode, float(rng if rng is not None else 0.0), out.data_ptr(), out2.data_ptr() if out2 is not None else None,
            stream_ptr())
    head = (f.data_ptr(), f2.data_ptr() if f2 is not None else None, B, Cc, h, w, sf, pre)
    rc = _lib.ESR_E_UNSUPPORTED
    if sep is not None:
        rc = _lib.lib.esr_cem_upscale_sep(*head, sep[0].data_ptr(), sep[1].data_ptr(), taps.shape[0], *tail)
    if rc == _lib.ESR_E_UNSUPPORTED:
        rc = _lib.lib.esr_cem_upscale(*head, taps.data_ptr(), taps.shape[0], *tail)
    check(rc, 'esr_cem_upscale')
    return (out, out2) if mode == 3 else out


# K(.) folded into the upscale launch (esr_cem_filter_upscale_sep; both tap sets rank one) — for launches of at most FOLD_MAX_TILES 64 x 64 output
# tiles.  Measured on MI355X (tools/experiments/cem_fold_ab.py, profiles/r06_cem_fold_ab.log): every tile re-filters its own (23 + 26)^2 window of
# the LR operand, 1.8 x the multiply-adds of the separate filter launch per LR pixel, inside a kernel that is instruction-issue bound — the
# projection gets SLOWER where the launches are large (configs[1], 8,256 tiles: 0.200 vs 0.186 ms; configs[4]: 1.27 vs 1.15 ms) and faster only
# where a launch is all latency (configs[0] on the GPU, 48 tiles: 557.7 vs 562.7 us per forward).  True / False force it on / off (tests).
FUSE_FILTER_UPSCALE = None
FOLD_MAX_TILES = 512


def filter_upscale_raw(e, taps_inv, taps_up, sf, pre, e2=None, g=None, crop=0, mode=0, rng=0.0, out=None):
    """upscale_raw(lr_filter_raw(e, taps_inv), taps_up, ..., f2=lr_filter_raw(e2, taps_inv)) — as ONE launch (esr_cem_filter_upscale_sep: every tile
    filters its own window of e on chip, bit-identical to the two launches) when both filters are separable and the windows fit, else as the two
    (three) launches."""
    e = _prep(e, 'LR image')
    sep_i, sep_u = _sep(taps_inv, e.device), _sep(taps_up, e.device)
    B, Cc, h, w = e.shape
    Ho, Wo = sf * h - 2 * crop, sf * w - 2 * crop
    fold = FUSE_FILTER_UPSCALE
    if fold is None:
        # small launches only — and only where the separate filter launch is the TILE kernel, whose arithmetic the fold reproduces to the bit: images that
        # take the wave-streaming filter (esr_cem_sep_form) are never folded, so that a batch and any shard of it run the same arithmetic whatever their size
        fold = B * Cc * ((Ho + 63) // 64) * ((Wo + 63) // 64) <= FOLD_MAX_TILES and \
            _lib.lib.esr_cem_sep_form(2, sf, int(_taps(taps_inv, e.device).shape[0]), pre, h, w) == 0
    if fold and sep_i is not None and sep_u is not None:
        o = out if out is not None else torch.empty(B, Cc, Ho, Wo, dtype=torch.float32, device=e.device)
        assert o.shape == (B, Cc, Ho, Wo) and o.dtype == torch.float32 and o.is_contiguous()
        o2 = torch.empty_like(o) if mode == 3 else None
        e2p = _prep(e2, 'LR image') if e2 is not None else None
        gp = _prep(g, 'generated image') if g is not None else None
        if gp is not None:
            assert gp.shape == (B, Cc, sf * h, sf * w)
        rc = _lib.lib.esr_cem_filter_upscale_sep(e.data_ptr(), e2p.data_ptr() if e2p is not None else None, B, Cc, h, w, sf, pre, sep_i[0].data_ptr(),
                                                 sep_i[1].data_ptr(), _taps(taps_inv, e.device).shape[0], sep_u[0].data_ptr(), sep_u[1].data_ptr(),
                                                 _taps(taps_up, e.device).shape[0], gp.data_ptr() if gp is not None else None, crop, mode,
                                                 float(rng if rng is not None else 0.0), o.data_ptr(), o2.data_ptr() if o2 is not None else None, stream_ptr())
        if rc != _lib.ESR_E_UNSUPPORTED:
            check(rc, 'esr_cem_filter_upscale_sep')
            return (o, o2) if mode == 3 else o
    f = lr_filter_raw(e, taps_inv)
    f2 = lr_filter_raw(e2, taps_inv) if e2 is not None else None
    return upscale_raw(f, taps_up, sf, pre, f2=f2, g=g, crop=crop, mode=mode, rng=rng, out=out)


def adjoint_raw(dy, tabs, kind, sf, pre, in_shape, base=None, alpha=1.0):
    """Transpose of one CEM filter: dy (the op's output gradient) -> base + alpha * (gradient w.r.t. the op's input of shape `in_shape`).
    tabs: autograd.AdjointTables — rank-one taps run the two 1-D passes (esr_cem_adjoint_sep), anything else the 2-D gather."""
    dy = _prep(dy, 'gradient')
    B, Cc = in_shape[0], in_shape[1]
    k = tabs.k
    dx = torch.empty(in_shape, dtype=torch.float32, device=dy.device)
    hq, wq = dy.shape[2], dy.shape[3]
    hn, wn = in_shape[2], in_shape[3]
    if kind == 'downscale':      # frame = HR (the input); outputs q at sf*q+pre; unknowns = every HR pixel
        args = (sf, pre, hn, wn, 1, 0)
    elif kind == 'lr_filter':    # frame = LR
        args = (1, 0, hn, wn, 1, 0)
    else:                        # upscale: frame = HR (the output); unknowns = LR samples at sf*n+pre
        args = (1, 0, hq, wq, sf, pre)
    sq, oq, Ny, Nx, sn, on = args
    if base is not None:
        base = _prep(base, 'gradient')
        assert tuple(base.shape) == tuple(in_shape)
    if tabs.v is not None:
        tmp = torch.empty(B * Cc * hq * wn, dtype=torch.float32, device=dy.device)
        check(_lib.lib.esr_cem_adjoint_sep(dy.data_ptr(), B, Cc, hq, wq, sq, oq, Ny, Nx, tabs.v.data_ptr(), tabs.h.data_ptr(), k, hn, wn, sn, on, tmp.data_ptr(),
                                           base.data_ptr() if base is not None else None, float(alpha), dx.data_ptr(), stream_ptr()), 'esr_cem_adjoint_sep')
        return dx
    check(_lib.lib.esr_cem_adjoint(dy.data_ptr(), B, Cc, hq, wq, sq, oq, Ny, Nx, tabs.full.data_ptr(), k, hn, wn, sn, on, dx.data_ptr(), 0,
                                   stream_ptr()), 'esr_cem_adjoint')
    if base is not None or alpha != 1.0:
        dx = dx * alpha if base is None else torch.add(base, dx, alpha=alpha)
    return dx


# ---- public, differentiable entry points -------------------------------------------------------------------------
def downscale(y, taps, sf, pre):
    if _needs_grad(y):
        from . import autograd as AG
        return AG.CemLinear.apply(y, taps, 'downscale', sf, pre)
    return downscale_raw(y, taps, sf, pre)


def lr_filter(x, taps):
    if _needs_grad(x):
        from . import autograd as AG
        return AG.CemLinear.apply(x, taps, 'lr_filter', 1, 0)
    return lr_filter_raw(x, taps)


def upscale(x, taps, sf, pre):
    if _needs_grad(x):
        from . import autograd as AG
        return AG.CemLinear.apply(x, taps, 'upscale', sf, pre)
    return upscale_raw(x, taps, sf, pre)


def project(lr, g, taps_down, taps_inv, taps_up, sf, pre, lr_pad=0, crop=0, sigmoid_range=None, decomposed=False):
    """CEM_PyTorch.forward after the generator call (reference CEMnet.py:303-311).
    lr: un-padded LR [B,C,h0,w0]; g: generator output on the (replicate-padded by lr_pad) LR frame."""
    if _needs_grad(lr, g):
        from . import autograd as AG
        return AG.cem_project_with_grad(lr, g, taps_down, taps_inv, taps_up, sf, pre, lr_pad, crop, sigmoid_range, decomposed)
    if sigmoid_range is None and not decomposed:
        B = g.shape[0]
        nb = min(PROJECT_CHUNK_IMAGES, PROJECT_CHUNK_BYTES // max(g[0].numel() * 4, 1))
        if nb >= 2 and B > nb and g.is_contiguous() and lr.is_contiguous() and g.dtype == torch.float32 and lr.dtype == torch.float32 and \
                PROJECT_CHUNK_BYTES < g.numel() * 4:
            # `g` is read twice — by the strided downscale and again by the upscale that adds the correction to it.  A whole batch of it
            # (configs[1]: 135 MB) is gone from the 256 MB Infinity Cache by the time the upscale comes back to it behind the other two kernels'
            # traffic; a chunk of a few images (8 x 4.2 MB at 592 x 592) is still there.  Same kernels, same values, image by image.
            out = torch.empty(B, g.shape[1], g.shape[2] - 2 * crop, g.shape[3] - 2 * crop, dtype=torch.float32, device=g.device)
            for b0 in range(0, B, nb):
                gc, lc = g[b0:b0 + nb], lr[b0:b0 + nb]
                e = downscale_raw(gc, taps_down, sf, pre, lr=lc, lr_pad=lr_pad)
                filter_upscale_raw(e, taps_inv, taps_up, sf, pre, g=gc, crop=crop, mode=1, out=out[b0:b0 + nb])
            return out
        e = downscale_raw(g, taps_down, sf, pre, lr=lr, lr_pad=lr_pad)       # x - D(g) on the padded frame
        return filter_upscale_raw(e, taps_inv, taps_up, sf, pre, g=g, crop=crop, mode=1)       # crop(g + U(K(x - D g)))
    lr_p = torch.nn.functional.pad(lr, (lr_pad,) * 4, mode='replicate') if lr_pad else lr
    dg = downscale_raw(g, taps_down, sf, pre)
    if decomposed:
        ortho, ns = filter_upscale_raw(lr_p, taps_inv, taps_up, sf, pre, e2=dg, g=g, crop=crop, mode=3)
        if sigmoid_range is not None:
            ns = torch.tanh(ns) * sigmoid_range
        return [ortho, ns]
    return filter_upscale_raw(lr_p, taps_inv, taps_up, sf, pre, e2=dg, g=g, crop=crop, mode=2, rng=sigmoid_range)
