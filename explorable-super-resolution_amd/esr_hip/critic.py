"""The critic on the library's kernels: Discriminator_VGG_128 (reference codes/models/modules/architecture.py:446-508) forward, backward and
the backward of its backward (the WGAN-GP penalty, codes/models/modules/loss.py:260-279, differentiates d critic / d input), with torch
only carrying the autograd graph.

  * activations live in the conv kernels' layout ([planes][B][CG][H+2][W+2][8] bf16, zero border; one plane in 'bf16', hi+lo in 'split')
  * every 3x3 stride-1 conv is one esr_conv3x3 launch; wide outputs (128-512 channels) run as 64-channel output slices of ONE launch
  * every 4x4 stride-2 conv is the SAME 3x3 stride-1 kernel on the space-to-depth input: with S[(py,px,c)][i][j] = X[c][2i+py][2j+px],
        sum_{dy,dx} W[dy][dx] X[2oy+dy-1][2ox+dx-1]  =  sum_{ty,tx,py,px} E[(py,px)][ty][tx] S[(py,px)][oy+ty-1][ox+tx-1],   dy = 2ty+py-1
    i.e. a 3x3 conv over 4x the channels whose weight E holds the 16 taps of W in 16 of its 36 (tap, parity) slots (zeros elsewhere); the
    producing layer's normalise+activate kernel stores S directly (esr_bn_apply, s2d), so no tensor is ever re-laid-out
  * BatchNorm (batch statistics) + LeakyReLU, their gradient and the gradient of their gradient are the fused closed-form kernels of
    csrc/esr_critic.hip
  * data / weight gradients of the convs are the generator's kernels (transposed + flipped packs; esr_conv3x3_wgrad)
Each op is a torch.autograd.Function whose backward is built from the other Functions, so `create_graph=True` (the penalty) differentiates
through the backward pass with the same kernels: conv <-> data-gradient are each other's adjoint, the weight gradient is bilinear.
The two Linear layers of the classifier stay on torch (rocBLAS): 51 MFLOP of the critic's 2.2 GFLOP per image."""
import ctypes as C
import os

import torch

from . import _lib
from . import act as A
from ._lib import ActView, BnDesc, EsrError, check

SLOPE = 0.2
_state = {'input_grad_only': 0}


class input_grad_only:
    """with input_grad_only(): ... — backward passes inside compute data gradients only (the penalty's d critic / d input: autograd cannot
    tell the conv nodes that the weight gradients it would also hand back are not wanted by torch.autograd.grad(inputs=[interp]))."""

    def __enter__(self):
        _state['input_grad_only'] += 1

    def __exit__(self, *exc):
        _state['input_grad_only'] -= 1



# ------------------------------------------------------------------------------------------------ activation tensors
def new_at(planes, B, ncg, H, W, device):
    """Uninitialised: conv outputs and data gradients are only ever read at interior pixels; everything that becomes a conv (or weight-
    gradient) INPUT is produced by pack_nchw or esr_bn_apply, which write the one-pixel zero border themselves."""
    return torch.empty(planes, B, ncg, H + 2, W + 2, 8, dtype=torch.bfloat16, device=device)


def _tap_masks():
    """Per parity s = 2 py + px of the space-to-depth input: the taps (bit 3 ty + tx) of the embedded 3x3 weight that are non-zero — rows
    ty in {1 - py, 2 - py}, columns tx in {1 - px, 2 - px} (module docstring: dy = 2 ty + py - 1) — and the same for the flipped taps of the
    data-gradient pack (tap (2 - ty, 2 - tx))."""
    fwd, flipped = [], []
    for s in range(4):
        py, px = s >> 1, s & 1
        m = f = 0
        for ty in (1 - py, 2 - py):
            for tx in (1 - px, 2 - px):
                m |= 1 << (3 * ty + tx)
                f |= 1 << (3 * (2 - ty) + (2 - tx))
        fwd.append(m)
        flipped.append(f)
    return fwd, flipped


MASK_FWD, MASK_FLIPPED = _tap_masks()
if os.environ.get('ESR_CRITIC_MASKS') == '0':          # experiments: multiply the structural zeros too (same results)
    MASK_FWD = MASK_FLIPPED = None


def view_of(t, cg0=0, ncg=None):
    P, B, CG, Hp, Wp, _ = t.shape
    n = CG - cg0 if ncg is None else ncg
    cs = Hp * Wp
    off = cg0 * cs * 16
    hi = t.data_ptr() + off
    lo = hi + t.stride(0) * 2 if P == 2 else None
    return ActView(hi, lo, n, Hp - 2, Wp - 2, CG * cs, cs, 0)


class _Layer:
    pass


class CriticEngine:
    """Launch planner of one Discriminator_VGG_128.  precision: 'bf16' (one plane, one MFMA per product: what configs[2] names) or
    'split' (bf16 hi+lo, three MFMAs: fp32-class)."""

    def __init__(self, netD, precision='split'):
        self.net = netD
        self.precision = None
        self.layers = []
        mods = list(netD.features)
        i = 0
        while i < len(mods):
            conv = mods[i]
            if not isinstance(conv, torch.nn.Conv2d):
                raise EsrError('unexpected module %r in Discriminator_VGG_128.features' % (conv,))
            k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            if (k, s, p) not in ((3, 1, 1), (4, 2, 1)) or conv.kernel_size[0] != conv.kernel_size[1] or conv.groups != 1 or conv.bias is None:
                raise EsrError('critic conv %r: the kernels implement 3x3 stride 1 and 4x4 stride 2 (padding 1)' % (conv,))
            L = _Layer()
            L.conv, L.strided, L.bn = conv, s == 2, None
            i += 1
            if i < len(mods) and isinstance(mods[i], torch.nn.BatchNorm2d):
                L.bn = mods[i]
                i += 1
            elif i < len(mods) and not isinstance(mods[i], torch.nn.LeakyReLU):
                raise EsrError('critic norm %r: BatchNorm2d or none' % (mods[i],))
            if not (i < len(mods) and isinstance(mods[i], torch.nn.LeakyReLU) and abs(mods[i].negative_slope - SLOPE) < 1e-12):
                raise EsrError('critic activation: LeakyReLU(0.2) after every conv block')
            i += 1
            L.cin, L.cout = conv.in_channels, conv.out_channels
            if L.cout % 64 or (L.strided and L.cin % 32):
                raise EsrError('critic widths: multiples of 64 output channels (and of 32 input channels in front of a stride-2 conv)')
            L.index = len(self.layers)
            self.layers.append(L)
        if self.layers[0].strided or not getattr(netD, 'last_FC_layers', True):
            raise EsrError('critic layout: a 3x3 first conv and the Linear classifier (num_2_strides = 5)')
        self._fp = None
        self._batch = A.PackBatch()
        self._wgb = {}
        self.set_precision(precision)

    # ------------------------------------------------------------------ weights
    def set_precision(self, precision):
        assert precision in ('bf16', 'split')
        if precision == self.precision:
            return
        self.precision, self.planes, self.split = precision, (2 if precision == 'split' else 1), precision == 'split'
        self._fp = None
        for L in self.layers:
            L.fwd = L.tr = L.E = None

    @staticmethod
    def _embed_index(cout, cin, device):
        """Flat index into E [cout][4 cin][3][3] of every element of a 4x4 weight [cout][cin][4][4] (module docstring; channel order of
        the space-to-depth input as esr_bn_apply stores it: group g, parity s = 2 py + px, lane e -> channel (16 (g // 4) + 4 s + g % 4) * 8
        + e — four consecutive groups, i.e. one 32-channel MFMA tile, share a parity)."""
        co = torch.arange(cout, device=device).view(-1, 1, 1, 1)
        c = torch.arange(cin, device=device).view(1, -1, 1, 1)
        d = torch.arange(4, device=device)
        t, par = (d + 1) // 2, (d + 1) % 2            # dy = 2 ty + py - 1
        ty, py = t.view(1, 1, -1, 1), par.view(1, 1, -1, 1)
        tx, px = t.view(1, 1, 1, -1), par.view(1, 1, 1, -1)
        g = c // 8
        ch = ((g // 4) * 16 + (py * 2 + px) * 4 + g % 4) * 8 + c % 8
        return (((co * (4 * cin) + ch) * 3 + ty) * 3 + tx).reshape(-1)

    def _build_packs(self, L):
        dev = L.conv.weight.device
        fmt = self.split
        if L.strided:
            L.E = torch.zeros(L.cout, 4 * L.cin, 3, 3, dtype=torch.float32, device=dev)
            L.E_index = self._embed_index(L.cout, L.cin, dev)
            w, cin_e = L.E, 4 * L.cin
        else:
            w, cin_e = L.conv.weight, L.cin
        L.cin_e, L.wsrc = cin_e, w
        ncg_in = (cin_e + 7) // 8
        # forward: one 64-row pack per output slice, back to back in one buffer (esr_conv3x3_desc.cout > 64)
        nsl = L.cout // 64
        per = _lib.lib.esr_conv_wpack_bytes(ncg_in, 64, A.fmt_code(fmt))
        buf = torch.empty(nsl * per, dtype=torch.uint8, device=dev)
        L.fwd_packs = []
        for s in range(nsl):
            pk = A.PackedConv(w, None, 0, split=fmt, rows=list(range(64 * s, 64 * s + 64)))
            pk.wpack = buf[s * per:(s + 1) * per]
            L.fwd_packs.append(pk)
        L.fwd = _SlicedPack(buf, L.conv.bias, fmt)
        # data gradient: transposed + flipped, one pack per 64 INPUT channels
        ncg_k = L.cout // 8
        m = min(cin_e, 64)
        nsl_t = max(cin_e // 64, 1)
        per_t = _lib.lib.esr_conv_wpack_bytes(ncg_k, m, A.fmt_code(fmt))
        buf_t = torch.empty(nsl_t * per_t, dtype=torch.uint8, device=dev)
        L.tr_packs = []
        for s in range(nsl_t):
            pk = A.PackedConv(w, None, 0, split=fmt, transposed=True, m_slice=(64 * s, min(cin_e, 64 * s + 64)))
            pk.wpack = buf_t[s * per_t:(s + 1) * per_t]
            L.tr_packs.append(pk)
        L.tr = _SlicedPack(buf_t, None, fmt)

    def refresh(self):
        """Weight packs follow the parameters (one batched re-pack launch when any conv weight changed)."""
        fp = tuple((L.conv.weight.data_ptr(), L.conv.weight._version, L.conv.bias.data_ptr()) for L in self.layers)
        if fp == self._fp:
            return
        packs = []
        for L in self.layers:
            A.require_gpu(L.conv.weight, 'critic weight')
            if L.fwd is None:
                self._build_packs(L)
            if L.strided:
                L.E.view(-1)[L.E_index] = L.conv.weight.detach().float().reshape(-1)
            L.fwd.bias = L.conv.bias.detach()
            packs += L.fwd_packs + L.tr_packs
        self._batch.run(packs)
        self._fp = fp

    # ------------------------------------------------------------------ launches (all take / return activation tensors)
    def conv_fwd(self, L, x, use_bias=True):
        P, B, _, Hp, Wp, _ = x.shape
        y = new_at(P, B, L.cout // 8, Hp - 2, Wp - 2, x.device)
        kw = dict(tap_mask_k=MASK_FWD, tap_mask_k_shift=1) if (L.strided and MASK_FWD) else {}      # chunk cp = group pair: parity of quad cp >> 1
        A.conv3x3(L.fwd, view_of(x), B, Hp - 2, Wp - 2, L.cout, out=view_of(y), use_bias=use_bias, reverse=False, **kw)
        return y

    def conv_dgrad(self, L, dy):
        P, B, _, Hp, Wp, _ = dy.shape
        dx = new_at(P, B, (L.cin_e + 7) // 8, Hp - 2, Wp - 2, dy.device)
        kw = dict(tap_mask_m=MASK_FLIPPED) if (L.strided and MASK_FLIPPED) else {}                     # 32-row output tile j = input quad j
        A.conv3x3(L.tr, view_of(dy), B, Hp - 2, Wp - 2, L.cin_e, out=view_of(dx), use_bias=False, reverse=False, **kw)
        return dx

    def conv_wgrad(self, L, dy, x):
        P, B, _, Hp, Wp, _ = dy.shape
        dw, db = A.conv3x3_wgrad(view_of(dy), view_of(x), None, 0, (L.cout, L.cin_e, 3, 3), B, Hp - 2, Wp - 2, 1.0, 1, dy.device,
                                 tap_masks=MASK_FWD if L.strided else None)
        if L.strided:
            dw = dw.view(-1)[L.E_index].view(L.cout, L.cin, 4, 4)
        return dw, db

    def _bn_desc(self, L, y, st, s2d, dz=None, u=None, out0=None, out1=None, sums2=None, sums3=None):
        d = BnDesc()
        d.y = view_of(y)
        for name, t in (('dz', dz), ('u', u), ('out0', out0), ('out1', out1)):
            setattr(d, name, view_of(t) if t is not None else A.NO_VIEW)
        d.B, d.groups, d.C = y.shape[1], 1, L.cout
        ptr = lambda t: t.data_ptr() if t is not None else None
        d.scale, d.shift, d.mean, d.rstd = ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.rstd)
        d.gamma = ptr(st.gamma)
        d.sums2, d.sums3 = ptr(sums2), ptr(sums3)
        d.slope, d.const_stats, d.s2d = SLOPE, 1 if st.const else 0, 1 if s2d else 0
        return d


class _SlicedPack:
    """What A.conv3x3 needs of a weight pack: the packed bytes (64-row slices back to back), the bias array and the operand format."""

    def __init__(self, wpack, bias, split):
        self.wpack, self.bias, self.split = wpack, (bias.detach() if bias is not None else None), split


class _Stats:
    """Per-layer normalisation state of one forward call (plain tensors: constants of the autograd graph; the dependence of the batch
    statistics on the conv output is inside the closed-form gradients)."""
    scale = shift = mean = rstd = gamma = None
    const = True


# ------------------------------------------------------------------------------------------------ autograd
class _PackIn(torch.autograd.Function):
    """fp32 NCHW -> activation tensor (and back: _UnpackOut); each is the other's adjoint."""

    @staticmethod
    def forward(ctx, x, planes):
        ctx.nc, ctx.planes = x.shape[1], planes
        x = x.detach().float().contiguous()
        B, Cc, H, W = x.shape
        t = new_at(planes, B, (Cc + 7) // 8, H, W, x.device)
        A.pack_nchw(x, view_of(t), 0, Cc)
        return t

    @staticmethod
    def backward(ctx, dt):
        return _UnpackOut.apply(dt, ctx.nc), None


class _UnpackOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, nc):
        ctx.planes = t.shape[0]
        P, B, CG, Hp, Wp, _ = t.shape
        out = torch.empty(B, nc, Hp - 2, Wp - 2, dtype=torch.float32, device=t.device)
        v = view_of(t, 0, (nc + 7) // 8)
        check(_lib.lib.esr_unpack_nchw(C.byref(v), B, nc, out.data_ptr(), A.stream_ptr()), 'esr_unpack_nchw')
        ctx.ncg = CG
        return out

    @staticmethod
    def backward(ctx, dout):
        t = _PackIn.apply(dout, ctx.planes)
        if t.shape[2] != ctx.ncg:
            raise EsrError('unpack adjoint: channel-group count mismatch')
        return t, None


class _Conv(torch.autograd.Function):
    """y = conv(x; W) + b (no activation).  Adjoint pair with _ConvT; _WGrad is the bilinear form both differentiate into."""

    @staticmethod
    def forward(ctx, eng, L, x, w, b):
        ctx.eng, ctx.L = eng, L
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return eng.conv_fwd(L, x.detach(), use_bias=b is not None)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = _ConvT.apply(ctx.eng, ctx.L, dy, w) if ctx.needs_input_grad[2] else None
        dw = db = None
        if (ctx.needs_input_grad[3] or (ctx.has_b and ctx.needs_input_grad[4])) and not _state['input_grad_only']:
            dw, db = _WGrad.apply(ctx.eng, ctx.L, dy, x)
        return None, None, dx, dw, (db if ctx.has_b else None)


class _ConvT(torch.autograd.Function):
    """dx = conv_T(dy; W): the data gradient of _Conv."""

    @staticmethod
    def forward(ctx, eng, L, dy, w):
        ctx.eng, ctx.L = eng, L
        ctx.save_for_backward(dy, w)
        return eng.conv_dgrad(L, dy.detach())

    @staticmethod
    def backward(ctx, ddx):
        dy, w = ctx.saved_tensors
        g_dy = _Conv.apply(ctx.eng, ctx.L, ddx, w, None) if ctx.needs_input_grad[2] else None
        g_w = _WGrad.apply(ctx.eng, ctx.L, dy, ddx)[0] if ctx.needs_input_grad[3] else None
        return None, None, g_dy, g_w


class _WGrad(torch.autograd.Function):
    """(dW, db) = (sum dy (x) x, sum dy).  Leaf of every graph this module builds (nothing here needs third derivatives)."""

    @staticmethod
    def forward(ctx, eng, L, dy, x):
        dw, db = eng.conv_wgrad(L, dy.detach(), x.detach())
        ctx.mark_non_differentiable(dw, db)
        return dw, db

    @staticmethod
    def backward(ctx, *g):
        raise EsrError('third-order derivatives of the critic are not implemented')


class _BNAct(torch.autograd.Function):
    """z = LeakyReLU(BatchNorm(y)) (batch statistics in training mode; no norm for the first block), stored space-to-depth when the next
    conv strides."""

    @staticmethod
    def forward(ctx, eng, L, y, gamma, beta, s2d, training):
        yd = y.detach()
        P, B, CG, Hp, Wp, _ = yd.shape
        H, W = Hp - 2, Wp - 2
        st = _Stats()
        dev = yd.device
        if L.bn is not None:
            Cc = L.cout
            bn = L.bn
            st.gamma = gamma.detach() if gamma is not None else None
            if training:
                st.const = False
                buf = torch.zeros(2 * Cc * 8 + 4 * Cc * 4, dtype=torch.uint8, device=dev)      # [sums: C x 2 doubles | mean | rstd | scale | shift]
                sums = buf[:Cc * 16].view(torch.float64)
                f = buf[Cc * 16:].view(torch.float32)
                st.mean, st.rstd, st.scale, st.shift = f[:Cc], f[Cc:2 * Cc], f[2 * Cc:3 * Cc], f[3 * Cc:]
                d = eng._bn_desc(L, yd, st, False)
                check(_lib.lib.esr_bn_reduce(C.byref(d), 0, sums.data_ptr(), A.stream_ptr()), 'esr_bn_reduce')
                mom = bn.momentum if bn.momentum is not None else 0.1
                track = bn.track_running_stats and bn.running_mean is not None
                check(_lib.lib.esr_bn_finalize(sums.data_ptr(), 1, Cc, B * H * W, bn.eps, mom, gamma.data_ptr() if gamma is not None else None,
                                               beta.data_ptr() if beta is not None else None, st.mean.data_ptr(), st.rstd.data_ptr(), st.scale.data_ptr(),
                                               st.shift.data_ptr(), bn.running_mean.data_ptr() if track else None,
                                               bn.running_var.data_ptr() if track else None, A.stream_ptr()), 'esr_bn_finalize')
                if track and bn.num_batches_tracked is not None:
                    bn.num_batches_tracked += 1
            else:
                rstd = torch.rsqrt(bn.running_var.float() + bn.eps)
                g = gamma.detach().float() if gamma is not None else torch.ones_like(rstd)
                bt = beta.detach().float() if beta is not None else torch.zeros_like(rstd)
                st.scale = (g * rstd).contiguous()
                st.shift = (bt - st.scale * bn.running_mean.float()).contiguous()
        Ho, Wo = (H // 2, W // 2) if s2d else (H, W)
        if s2d and (H % 2 or W % 2):
            raise EsrError('critic: odd feature-map size %dx%d in front of a stride-2 conv' % (H, W))
        z = new_at(P, B, CG * 4 if s2d else CG, Ho, Wo, dev)
        d = eng._bn_desc(L, yd, st, s2d, out0=z)
        check(_lib.lib.esr_bn_apply(C.byref(d), 0, A.stream_ptr()), 'esr_bn_apply')
        ctx.eng, ctx.L, ctx.st, ctx.s2d = eng, L, st, s2d
        ctx.save_for_backward(y, gamma)
        ctx.has_affine = gamma is not None
        return z

    @staticmethod
    def backward(ctx, dz):
        y, gamma = ctx.saved_tensors
        dy, dgamma, dbeta = _BNActBwd.apply(ctx.eng, ctx.L, ctx.st, ctx.s2d, y, gamma, dz)
        return None, None, dy, (dgamma if ctx.has_affine else None), (dbeta if ctx.has_affine else None), None, None


class _BNActBwd(torch.autograd.Function):
    """(dy, dgamma, dbeta) from (y, gamma, dz): the backward of _BNAct as a differentiable op (csrc/esr_critic.hip, 'backward' and 'double
    backward')."""

    @staticmethod
    def forward(ctx, eng, L, st, s2d, y, gamma, dz):
        ctx.set_materialize_grads(False)
        yd, dzd = y.detach(), dz.detach().contiguous()
        P, B, CG, Hp, Wp, _ = yd.shape
        dev = yd.device
        dy = new_at(P, B, CG, Hp - 2, Wp - 2, dev)
        sums2 = dgamma = dbeta = None
        if not st.const:
            Cc = L.cout
            sums2 = torch.zeros(Cc * 2, dtype=torch.float64, device=dev)
            d = eng._bn_desc(L, yd, st, s2d, dz=dzd)
            check(_lib.lib.esr_bn_reduce(C.byref(d), 1, sums2.data_ptr(), A.stream_ptr()), 'esr_bn_reduce')
            pg = torch.empty(2, Cc, dtype=torch.float32, device=dev)
            dgamma, dbeta = pg[0], pg[1]
            check(_lib.lib.esr_bn_param_grads(sums2.data_ptr(), None, None, 1, Cc, B * (Hp - 2) * (Wp - 2), dgamma.data_ptr(), dbeta.data_ptr(), None,
                                              A.stream_ptr()), 'esr_bn_param_grads')
        d = eng._bn_desc(L, yd, st, s2d, dz=dzd, out0=dy, sums2=sums2)
        check(_lib.lib.esr_bn_apply(C.byref(d), 1, A.stream_ptr()), 'esr_bn_apply')
        ctx.eng, ctx.L, ctx.st, ctx.s2d, ctx.sums2 = eng, L, st, s2d, sums2
        ctx.save_for_backward(y, gamma, dz)
        if dgamma is None:
            dgamma, dbeta = torch.zeros(L.cout, dtype=torch.float32, device=dev), torch.zeros(L.cout, dtype=torch.float32, device=dev)
        ctx.mark_non_differentiable(dgamma, dbeta)
        return dy, dgamma, dbeta

    @staticmethod
    def backward(ctx, u, u_dgamma=None, u_dbeta=None):
        y, gamma, dz = ctx.saved_tensors
        eng, L, st, s2d = ctx.eng, ctx.L, ctx.st, ctx.s2d
        if u is None:
            return (None,) * 7
        yd, dzd, ud = y.detach(), dz.detach().contiguous(), u.detach().contiguous()
        P, B, CG, Hp, Wp, _ = yd.shape
        dev = yd.device
        g_dz = torch.empty_like(dzd)
        g_y = new_at(P, B, CG, Hp - 2, Wp - 2, dev)
        sums3 = g_gamma = None
        if not st.const:
            Cc = L.cout
            sums3 = torch.zeros(Cc * 3, dtype=torch.float64, device=dev)
            d = eng._bn_desc(L, yd, st, s2d, dz=dzd, u=ud)
            check(_lib.lib.esr_bn_reduce(C.byref(d), 2, sums3.data_ptr(), A.stream_ptr()), 'esr_bn_reduce')
            if gamma is not None and ctx.needs_input_grad[5]:
                g_gamma = torch.empty(Cc, dtype=torch.float32, device=dev)
                check(_lib.lib.esr_bn_param_grads(ctx.sums2.data_ptr(), sums3.data_ptr(), st.rstd.data_ptr(), 1, Cc, B * (Hp - 2) * (Wp - 2), None, None,
                                                  g_gamma.data_ptr(), A.stream_ptr()), 'esr_bn_param_grads')
        d = eng._bn_desc(L, yd, st, s2d, dz=dzd, u=ud, out0=g_dz, out1=g_y, sums2=ctx.sums2, sums3=sums3)
        check(_lib.lib.esr_bn_apply(C.byref(d), 2, A.stream_ptr()), 'esr_bn_apply')
        return None, None, None, None, (g_y if not st.const else None), g_gamma, g_dz


def critic_forward(eng, x):
    """Logits [B, 1] of the critic for fp32 NCHW images `x`, differentiable to any order the WGAN-GP step needs."""
    A.require_gpu(x, 'critic input')
    net = eng.net
    eng.refresh()
    training = net.training
    L0 = eng.layers[0]
    if x.shape[1] != L0.cin:
        raise EsrError('critic input: %d channels expected' % L0.cin)
    t = _PackIn.apply(x, eng.planes)
    for i, L in enumerate(eng.layers):
        nxt_strided = i + 1 < len(eng.layers) and eng.layers[i + 1].strided
        y = _Conv.apply(eng, L, t, L.conv.weight, L.conv.bias)
        gamma, beta = (L.bn.weight, L.bn.bias) if L.bn is not None else (None, None)
        t = _BNAct.apply(eng, L, y, gamma, beta, nxt_strided, training)
    feat = _UnpackOut.apply(t, eng.layers[-1].cout)
    return net.classifier(feat.reshape(feat.size(0), -1))


# ================================================================================================ fused passes
# The per-layer Functions above are the readable definition (and what the tests check piece by piece against float64).  Executed that way
# a critic step is ~700 Python-level operations (autograd nodes, FFI calls, allocations) and host-bound: 12 ms of host work for 10 ms of
# kernels.  Below, each of the three passes — forward, backward, backward-of-backward — is ONE launch list (esr_run) over all ten blocks,
# and the graph has two nodes: _CriticFwd (outputs: the features AND every block's pre-normalisation conv output y_l, so that cotangents
# of the y_l can arrive) and _CriticBwd (the backward pass as a differentiable op of (d features, y_l, parameters)).  Same kernels, same
# launch order, bit-identical results (tests/test_gpu_critic.py).
class _Scratch:
    """One zeroed device buffer per pass for the per-channel sums / statistics, handed out as raw pointers."""

    def __init__(self, nbytes, device):
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self.base, self.off = self.buf.data_ptr(), 0

    def take(self, nbytes):
        p = self.base + self.off
        self.off += (nbytes + 255) // 256 * 256
        assert self.off <= self.buf.numel()
        return p


class _State:
    pass


def _desc(L, B, y, st, s2d, dz=None, u=None, out0=None, out1=None):
    d = BnDesc()
    d.y = view_of(y)
    if dz is not None:
        d.dz = view_of(dz)
    if u is not None:
        d.u = view_of(u)
    if out0 is not None:
        d.out0 = view_of(out0)
    if out1 is not None:
        d.out1 = view_of(out1)
    d.B, d.groups, d.C = B, 1, L.cout
    d.scale, d.shift, d.mean, d.rstd, d.gamma = st.scale, st.shift, st.mean, st.rstd, st.gamma
    d.sums2, d.sums3 = st.sums2, st.sums3
    d.slope, d.const_stats, d.s2d = SLOPE, 1 if st.const else 0, 1 if s2d else 0
    return d


def _emit_bn(rec, op, d, mode, sums=None):
    rec.emit(op, _lib.CmdBn(d, mode, sums))


def _fwd_pass(eng, x, training):
    """-> (features fp32 [B, C, h, w], state).  One launch list: pack, then per block conv -> statistics -> normalise + activate."""
    dev = x.device
    x = x.float().contiguous()
    B, Cin, H, W = x.shape
    P = eng.planes
    S = _State()
    S.B, S.planes, S.training, S.in_shape = B, P, training, (B, Cin, H, W)
    S.y, S.z, S.st, S.s2d = [], [], [], []
    nstat = sum(L.cout for L in eng.layers if L.bn is not None)
    S.scratch = _Scratch(nstat * (16 + 16) + 4096 * len(eng.layers), dev)
    rec = A.Recorder({})
    eval_keep = []
    with A.recording(rec):
        t = new_at(P, B, (Cin + 7) // 8, H, W, dev)
        A.pack_nchw(x, view_of(t), 0, Cin)
        S.t0 = t
        h, w = H, W
        tracked = []
        for i, L in enumerate(eng.layers):
            if L.strided:
                h, w = h // 2, w // 2
            y = new_at(P, B, L.cout // 8, h, w, dev)
            kw = dict(tap_mask_k=MASK_FWD, tap_mask_k_shift=1) if (L.strided and MASK_FWD) else {}
            A.conv3x3(L.fwd, view_of(t), B, h, w, L.cout, out=view_of(y), reverse=False, **kw)
            st = _Stats()
            st.sums2 = st.sums3 = None
            s2d = i + 1 < len(eng.layers) and eng.layers[i + 1].strided
            if s2d and (h % 2 or w % 2):
                raise EsrError('critic: odd feature-map size %dx%d in front of a stride-2 conv' % (h, w))
            if L.bn is not None:
                bn, Cc = L.bn, L.cout
                st.gamma = bn.weight.data_ptr() if bn.weight is not None else None
                if training:
                    st.const = False
                    sums = S.scratch.take(Cc * 16)
                    st.mean, st.rstd, st.scale, st.shift = (S.scratch.take(Cc * 4) for _ in range(4))
                    _emit_bn(rec, _lib.OP_BN_REDUCE, _desc(L, B, y, st, False), 0, sums)
                    track = bn.track_running_stats and bn.running_mean is not None
                    rec.emit(_lib.OP_BN_FINALIZE, _lib.CmdBnFinalize(sums, 1, Cc, B * h * w, bn.eps, bn.momentum if bn.momentum is not None else 0.1, st.gamma,
                                                                     bn.bias.data_ptr() if bn.bias is not None else None, st.mean, st.rstd, st.scale, st.shift,
                                                                     bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None))
                    if track and bn.num_batches_tracked is not None:
                        tracked.append(bn.num_batches_tracked)
                else:
                    rstd = torch.rsqrt(bn.running_var.float() + bn.eps)
                    g = bn.weight.detach().float() if bn.weight is not None else torch.ones_like(rstd)
                    bt = bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(rstd)
                    sc = (g * rstd).contiguous()
                    sh = (bt - sc * bn.running_mean.float()).contiguous()
                    eval_keep += [sc, sh]
                    st.scale, st.shift = sc.data_ptr(), sh.data_ptr()
            ho, wo = (h // 2, w // 2) if s2d else (h, w)
            z = new_at(P, B, (L.cout // 8) * (4 if s2d else 1), ho, wo, dev)
            _emit_bn(rec, _lib.OP_BN_APPLY, _desc(L, B, y, st, s2d, out0=z), 0)
            S.y.append(y); S.z.append(z); S.st.append(st); S.s2d.append(s2d)
            t = z
        Cl = eng.layers[-1].cout
        feat = torch.empty(B, Cl, h, w, dtype=torch.float32, device=dev)
        rec.emit(_lib.OP_UNPACK_NCHW, _lib.CmdUnpackNchw(view_of(t), B, Cl, feat.data_ptr()))
    rec.finish().run({})
    S.keep = eval_keep
    if tracked:
        torch._foreach_add_(tracked, 1)
    return feat, S


def _wgrad_batch(eng, S, pairs, dev):
    """Weight / bias gradients of all blocks in one launch (esr_conv3x3_wgrad_batch): pairs = [(layer, dy, x)].  Returns {layer index:
    (dW in the parameter's shape, db)}."""
    sizes = [L.cout * L.cin_e * 9 + L.cout for L, _, _ in pairs]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    descs, out, off = [], {}, 0
    for (L, dy, xin), n in zip(pairs, sizes):
        nw = L.cout * L.cin_e * 9
        dw, db = flat[off:off + nw].view(L.cout, L.cin_e, 3, 3), flat[off + nw:off + n]
        d, _, _ = A.wgrad_desc(view_of(dy), view_of(xin), None, 0, (L.cout, L.cin_e, 3, 3), S.B, dy.shape[3] - 2, dy.shape[4] - 2, 1.0, 1, dev, out=(dw, db))
        descs.append(d)
        out[L.index] = (dw, db)
        off += n
    A.conv3x3_wgrad_batch(descs, dev, cache=eng._wgb)
    for L, _, _ in pairs:
        if L.strided:
            dw, db = out[L.index]
            out[L.index] = (dw.reshape(-1)[L.E_index].view(L.cout, L.cin, 4, 4), db)
    return out


def _bwd_pass(eng, S, d_feat, g_ys, want_dx, want_params):
    """The backward pass as one launch list: per block (last to first) BatchNorm/LeakyReLU gradient -> [+ injected cotangent of y_l] -> data
    gradient.  Saves dy_l / dz_l in the state (the double backward needs them).  -> (d input fp32 or None, {layer: (dW, db)}, {layer:
    (dgamma, dbeta)})"""
    dev = S.t0.device
    B, P = S.B, S.planes
    n = len(eng.layers)
    S.dy, S.dz = [None] * n, [None] * n
    bnl = [L for L in eng.layers if L.bn is not None and S.training]
    scratch = _Scratch(sum(L.cout for L in bnl) * 16 + 4096 * n, dev)
    pg = torch.empty(2 * sum(L.cout for L in bnl), dtype=torch.float32, device=dev) if (want_params and bnl) else None
    pg_off, bn_grads = 0, {}
    rec = A.Recorder({})
    with A.recording(rec):
        dz = torch.empty_like(S.z[-1])
        if d_feat is None:
            rec.emit(_lib.OP_ZERO, _lib.CmdZero(dz.data_ptr(), dz.numel() // 8))
        else:
            d_feat = d_feat.detach().float().contiguous()
            A.pack_nchw(d_feat, view_of(dz), 0, d_feat.shape[1])
        for i in reversed(range(n)):
            L, st, y = eng.layers[i], S.st[i], S.y[i]
            dy = torch.empty_like(y)
            if not st.const:
                st.sums2 = scratch.take(L.cout * 16)
                _emit_bn(rec, _lib.OP_BN_REDUCE, _desc(L, B, y, st, S.s2d[i], dz=dz), 1, st.sums2)
                if pg is not None:
                    dg, db_ = pg[pg_off:pg_off + L.cout], pg[pg_off + L.cout:pg_off + 2 * L.cout]
                    pg_off += 2 * L.cout
                    rec.emit(_lib.OP_BN_PARAM_GRADS, _lib.CmdBnParamGrads(st.sums2, None, None, 1, L.cout, B * (y.shape[3] - 2) * (y.shape[4] - 2), dg.data_ptr(),
                                                                          db_.data_ptr(), None))
                    bn_grads[i] = (dg, db_)
            _emit_bn(rec, _lib.OP_BN_APPLY, _desc(L, B, y, st, S.s2d[i], dz=dz, out0=dy), 1)
            if g_ys is not None and g_ys[i] is not None:
                gy = g_ys[i].detach().contiguous()
                rec.keep.append(gy)
                A.act_combine(view_of(dy), B, A_=view_of(dy), alpha=1.0, Bv=view_of(gy), beta=1.0, s=1)
            S.dy[i], S.dz[i] = dy, dz
            if i > 0 or want_dx:
                xin = S.z[i - 1] if i > 0 else S.t0
                dx = torch.empty_like(xin)
                kw = dict(tap_mask_m=MASK_FLIPPED) if (L.strided and MASK_FLIPPED) else {}
                A.conv3x3(L.tr, view_of(dy), B, dy.shape[3] - 2, dy.shape[4] - 2, L.cin_e, out=view_of(dx), use_bias=False, reverse=False, **kw)
                dz = dx
        dx_in = None
        if want_dx:
            dx_in = torch.empty(S.in_shape, dtype=torch.float32, device=dev)
            rec.emit(_lib.OP_UNPACK_NCHW, _lib.CmdUnpackNchw(view_of(dz), B, S.in_shape[1], dx_in.data_ptr()))
    rec.finish().run({})
    S.bwd_scratch = scratch
    conv_grads = {}
    if want_params:
        conv_grads = _wgrad_batch(eng, S, [(L, S.dy[i], S.z[i - 1] if i > 0 else S.t0) for i, L in enumerate(eng.layers)], dev)
    return dx_in, conv_grads, bn_grads


def _bwd2_pass(eng, S, u, want_params):
    """The backward of the backward pass (u: cotangent of d input), first block to last: per block conv of the incoming cotangent -> gradient
    of the BatchNorm/LeakyReLU gradient.  -> (cotangent of d features, [cotangent of y_l], {layer: dW (second order)}, {layer: g_gamma})"""
    dev = S.t0.device
    B, P = S.B, S.planes
    n = len(eng.layers)
    bnl = [L for L in eng.layers if L.bn is not None and S.training]
    scratch = _Scratch(sum(L.cout for L in bnl) * 24 + 4096 * n, dev)
    gg_all = torch.empty(sum(L.cout for L in bnl), dtype=torch.float32, device=dev) if (want_params and bnl) else None
    gg_off, g_gammas, g_ys, uts = 0, {}, [None] * n, []
    rec = A.Recorder({})
    with A.recording(rec):
        ut = torch.empty_like(S.t0)
        u = u.detach().float().contiguous()
        A.pack_nchw(u, view_of(ut), 0, u.shape[1])
        for i, L in enumerate(eng.layers):
            st, y = S.st[i], S.y[i]
            uts.append(ut)
            gdy = torch.empty_like(y)
            kw = dict(tap_mask_k=MASK_FWD, tap_mask_k_shift=1) if (L.strided and MASK_FWD) else {}
            A.conv3x3(L.fwd, view_of(ut), B, y.shape[3] - 2, y.shape[4] - 2, L.cout, out=view_of(gdy), use_bias=False, reverse=False, **kw)
            if not st.const:
                st.sums3 = scratch.take(L.cout * 24)
                _emit_bn(rec, _lib.OP_BN_REDUCE, _desc(L, B, y, st, S.s2d[i], dz=S.dz[i], u=gdy), 2, st.sums3)
                if gg_all is not None and L.bn.weight is not None:
                    gg = gg_all[gg_off:gg_off + L.cout]
                    gg_off += L.cout
                    rec.emit(_lib.OP_BN_PARAM_GRADS, _lib.CmdBnParamGrads(st.sums2, st.sums3, st.rstd, 1, L.cout, B * (y.shape[3] - 2) * (y.shape[4] - 2), None, None,
                                                                          gg.data_ptr()))
                    g_gammas[i] = gg
            g_dz = torch.empty_like(S.dz[i])
            g_y = torch.empty_like(y)
            _emit_bn(rec, _lib.OP_BN_APPLY, _desc(L, B, y, st, S.s2d[i], dz=S.dz[i], u=gdy, out0=g_dz, out1=g_y), 2)
            rec.keep.append(gdy)
            if not st.const:
                g_ys[i] = g_y
            ut = g_dz
        g_dfeat = torch.empty(B, eng.layers[-1].cout, ut.shape[3] - 2, ut.shape[4] - 2, dtype=torch.float32, device=dev)
        rec.emit(_lib.OP_UNPACK_NCHW, _lib.CmdUnpackNchw(view_of(ut), B, eng.layers[-1].cout, g_dfeat.data_ptr()))
    rec.finish().run({})
    S.bwd2_scratch = scratch
    conv2 = {}
    if want_params:
        conv2 = {k: v[0] for k, v in _wgrad_batch(eng, S, [(L, S.dy[i], uts[i]) for i, L in enumerate(eng.layers)], dev).items()}
    return g_dfeat, g_ys, conv2, g_gammas


def _param_list(eng):
    """[(conv.weight, conv.bias, bn.weight or None, bn.bias or None)] flattened, None entries kept (positions are fixed: 4 per block)."""
    out = []
    for L in eng.layers:
        out += [L.conv.weight, L.conv.bias, L.bn.weight if L.bn is not None else None, L.bn.bias if L.bn is not None else None]
    return out


class _CriticFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, training, x, *params):
        ctx.set_materialize_grads(False)
        feat, S = _fwd_pass(eng, x.detach(), training)
        ctx.eng, ctx.S, ctx.np = eng, S, len(params)
        outs = (feat,) + tuple(S.y)
        ctx.save_for_backward(*[p for p in params if p is not None], *S.y)
        ctx.pmask = [p is not None for p in params]
        return outs

    @staticmethod
    def backward(ctx, d_feat, *g_ys):
        saved = list(ctx.saved_tensors)
        nreal = sum(ctx.pmask)
        it = iter(saved[:nreal])
        params = [next(it) if m else None for m in ctx.pmask]
        ys = saved[nreal:]
        want_dx = ctx.needs_input_grad[2]
        want_params = any(ctx.needs_input_grad[3:]) and not _state['input_grad_only']
        outs = _CriticBwd.apply(ctx.eng, ctx.S, want_dx, want_params, d_feat, len(ys), *g_ys, *ys, *params)
        dx, pgrads = outs[0], outs[1:]
        return (None, None, dx) + tuple(g if (g is not None and ctx.needs_input_grad[3 + k]) else None for k, g in enumerate(pgrads))


class _CriticBwd(torch.autograd.Function):
    """(d input, parameter gradients) = backward pass of the critic, as a function of (d features, cotangents injected at the y_l, the y_l
    themselves, the parameters): differentiable once more (WGAN-GP)."""

    @staticmethod
    def forward(ctx, eng, S, want_dx, want_params, d_feat, n, *rest):
        ctx.set_materialize_grads(False)
        g_ys, ys, params = rest[:n], rest[n:2 * n], rest[2 * n:]
        dx, conv_grads, bn_grads = _bwd_pass(eng, S, d_feat, g_ys if any(g is not None for g in g_ys) else None, want_dx, want_params)
        ctx.eng, ctx.S, ctx.n, ctx.nparams = eng, S, n, len(params)
        ctx.had_dfeat = d_feat is not None
        pg = []
        for i, L in enumerate(eng.layers):
            cw, cb = conv_grads.get(i, (None, None))
            bg, bb = bn_grads.get(i, (None, None))
            pg += [cw, cb, bg if L.bn is not None and L.bn.weight is not None else None, bb if L.bn is not None and L.bn.bias is not None else None]
        ctx.mark_non_differentiable(*[g for g in pg if g is not None])
        if dx is None:
            dx = torch.zeros((), device=S.t0.device)          # placeholder output (never used: the input asked for no gradient)
            ctx.mark_non_differentiable(dx)
        return (dx,) + tuple(pg)

    @staticmethod
    def backward(ctx, u, *u_params):
        n = ctx.n
        if u is None:
            return (None,) * (6 + 2 * n + ctx.nparams)
        want_params = any(ctx.needs_input_grad[6 + 2 * n:]) and not _state['input_grad_only']
        g_dfeat, g_ys, conv2, g_gammas = _bwd2_pass(ctx.eng, ctx.S, u, want_params)
        pg = []
        for i, L in enumerate(ctx.eng.layers):
            pg += [conv2.get(i), None, g_gammas.get(i), None]
        return (None, None, None, None, g_dfeat if ctx.had_dfeat else None, None) + (None,) * n + tuple(g_ys) + tuple(pg)


FUSED = os.environ.get('ESR_CRITIC_FUSED', '1') != '0'
_critic_forward_per_layer = critic_forward


def critic_forward(eng, x):
    """Logits [B, 1] of the critic for fp32 NCHW images `x`, differentiable to the order the WGAN-GP step needs.  Three launch lists per
    forward / backward / double backward (ESR_CRITIC_FUSED=0: one autograd node and several FFI calls per layer — the same kernels)."""
    if not FUSED:
        return _critic_forward_per_layer(eng, x)
    A.require_gpu(x, 'critic input')
    net = eng.net
    eng.refresh()
    if x.shape[1] != eng.layers[0].cin:
        raise EsrError('critic input: %d channels expected' % eng.layers[0].cin)
    outs = _CriticFwd.apply(eng, net.training, x, *_param_list(eng))
    feat = outs[0]
    return net.classifier(feat.reshape(feat.size(0), -1))
