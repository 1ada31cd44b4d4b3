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
import weakref

import torch

from . import _lib
from . import act as A
from ._lib import ActView, BnDesc, EsrError, check

SLOPE = 0.2
_state = {'input_grad_only': 0, 'group': None, 'only': None}


class input_grad_only:
    """with input_grad_only(): ... — backward passes inside compute data gradients only (the penalty's d critic / d input: autograd cannot
    tell the conv nodes that the weight gradients it would also hand back are not wanted by torch.autograd.grad(inputs=[interp])).
    group = g: the cotangent that enters a critic_forward_group() graph inside is non-zero for the g-th input only (the penalty
    differentiates the interpolated batch's logits alone) — the backward pass, and later the backward of that backward pass, then run on
    that input's images only.  A promise the caller makes: rows of the other inputs are not looked at.
    of = a logits tensor returned by critic_forward / critic_forward_group: the hints then apply to THAT call's graph only — any other critic
    graph differentiated inside the context (another model's critic, a second call) computes everything it is asked for."""

    def __init__(self, group=None, of=None):
        self.group, self.only = group, getattr(of, '_esr_critic_state', None)

    def __enter__(self):
        _state['input_grad_only'] += 1
        self.prev, _state['group'], _state['only'] = (_state['group'], _state['only']), self.group, self.only

    def __exit__(self, *exc):
        _state['input_grad_only'] -= 1
        _state['group'], _state['only'] = self.prev


def _hints_for(S):
    """(input gradients only?, group) as they apply to the forward call whose state is S"""
    if not _state['input_grad_only'] or (_state['only'] is not None and _state['only'] is not S):
        return False, None
    return True, _state['group']



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


MASK_FWD, MASK_FLIPPED = _tap_masks()       # (module attributes; None: multiply the structural zeros too — same results)


def view_of(t, cg0=0, ncg=None, b0=0):
    """View of channel groups [cg0, cg0 + ncg) starting at image b0 (the view has no batch size: the launch says how many images)."""
    if getattr(t, '_esr_stacked', False):              # [planes][CG][B][H+2][W+2][8]: see stacked_at()
        P, CG, B, Hp, Wp, _ = t.shape
        n = CG - cg0 if ncg is None else ncg
        cs = B * Hp * Wp
        hi = t.data_ptr() + (cg0 * cs + b0 * Hp * Wp) * 16
        return ActView(hi, hi + t.stride(0) * 2 if P == 2 else None, n, Hp - 2, Wp - 2, Hp * Wp, cs, 0)
    P, B, CG, Hp, Wp, _ = t.shape
    n = CG - cg0 if ncg is None else ncg
    cs = Hp * Wp
    off = (cg0 * cs + b0 * CG * cs) * 16
    hi = t.data_ptr() + off
    lo = hi + t.stride(0) * 2 if P == 2 else None
    return ActView(hi, lo, n, Hp - 2, Wp - 2, CG * cs, cs, 0)


def stacked_at(planes, B, ncg, H, W, device):
    """Group-major activation tensor [planes][CG][B][H+2][W+2][8] for SMALL feature maps: the B images of one channel group are stacked
    vertically, each with its own zero border rows, so that the whole batch is ONE image of B*(H+2) - 2 rows to a conv launch (tall_view):
    a tile then spans several images, and the layer's weights are fetched once per ~384 pixels instead of once per 16- or 64-pixel image.
    The rows between two images are their bottom / top borders: zero in every conv INPUT (the producers write them), garbage in conv
    OUTPUTS (which are only ever read at interior pixels)."""
    t = torch.empty(planes, ncg, B, H + 2, W + 2, 8, dtype=torch.bfloat16, device=device)
    t._esr_stacked = True
    return t


def tall_view(t, b0=0, nb=None):
    """Images [b0, b0 + nb) of the stacked tensor as one image per channel group (B' = 1): (view, rows)."""
    P, CG, B, Hp, Wp, _ = t.shape
    cs = B * Hp * Wp
    hi = t.data_ptr() + b0 * Hp * Wp * 16
    rows = (B - b0 if nb is None else nb) * Hp - 2
    return ActView(hi, hi + t.stride(0) * 2 if P == 2 else None, CG, rows, Wp - 2, cs * CG, cs, 0), rows


def conv_io(t_in, t_out, B, h, w, b0=0):
    """(input view, output view, B', H', W') of a conv launch over images [b0, b0 + B) of `t_in` -> `t_out`: per image, or the stacked
    images as one tall image."""
    if getattr(t_in, '_esr_stacked', False):
        assert getattr(t_out, '_esr_stacked', False)
        vi, rows = tall_view(t_in, b0, B)
        vo, rows_o = tall_view(t_out, b0, B)
        assert rows == rows_o
        return vi, vo, 1, rows, w
    return view_of(t_in, b0=b0), view_of(t_out, b0=b0), B, h, w


STACK_MAX = 8       # feature maps up to this height are stacked (0: never)
# fused passes lend every conv launch an fp32 workspace: the library splits the K axis of the deep layers' launches (24-100 workgroups walking
# 32-128 chunks each) over 2-8 workgroup sets (esr_conv3x3_desc.k_split_ws); 0: never
SPLITK = True
SPLITK_MAX_FLOATS = 16 << 20
# forward of a training-mode BatchNorm block: bn_finalize folded into the normalise + activate launch (esr_bn_finalize_apply); False: two launches
FUSE_FINALIZE = True


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
        self._free_sets = {}
        self.set_precision(precision)

    def unsupported_input(self, H, W):
        """None when the kernels run an H x W input, otherwise the reason (what _BufSet would raise): the stride-2 convs are 3x3 convs over the
        space-to-depth map of their input, which needs even sizes.  (The reference's module floors instead — and then fails in its Linear
        classifier unless the input is 128 x 128, codes/models/modules/architecture.py:446-508.)"""
        h, w = int(H), int(W)
        for L in self.layers:
            if L.strided:
                if h % 2 or w % 2:
                    return 'critic: odd feature-map size %dx%d in front of a stride-2 conv (input %dx%d)' % (h, w, H, W)
                h, w = h // 2, w // 2
        return None

    # ------------------------------------------------------------------ weights
    def set_precision(self, precision):
        assert precision in ('bf16', 'split')
        if precision == self.precision:
            return
        self.precision, self.planes, self.split = precision, (2 if precision == 'split' else 1), precision == 'split'
        self._fp = None
        self._free_sets = {}
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

    def pointer_fingerprint(self):
        """Everything a recorded launch list points at besides the buffer set: parameter / running-statistics storages and the weight packs."""
        out = []
        for L in self.layers:
            out += [L.conv.bias.data_ptr(), L.fwd.wpack.data_ptr() if L.fwd is not None else 0, L.tr.wpack.data_ptr() if L.tr is not None else 0]
            if L.bn is not None:
                out += [t.data_ptr() if t is not None else 0 for t in (L.bn.weight, L.bn.bias, L.bn.running_mean, L.bn.running_var)]
        return tuple(out)

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
        # (the per-layer graph carries no state object: a hint scoped to another graph — input_grad_only(of=...) — must not reach it)
        if (ctx.needs_input_grad[3] or (ctx.has_b and ctx.needs_input_grad[4])) and not (_state['input_grad_only'] and _state['only'] is None):
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
# recorded once per buffer set and replayed; the graph has two nodes: _CriticFwd (outputs: the features AND every block's
# pre-normalisation conv output y_l, so that cotangents of the y_l can arrive) and _CriticBwd (the backward pass as a differentiable op of
# (d features, y_l, parameters)).  Same kernels, same launch order as the per-layer graph; where two cotangents meet on one tensor
# (second-order + first-order on y_l) they are added in fp32 here, as bf16 planes by autograd there.
class _BufSet:
    """Every activation / gradient buffer the three passes of ONE critic call use, for one input shape, plus the launch lists recorded over
    them.  A call takes a free set from the engine (or makes one) and gives it back when its autograd graph is gone: steady state
    allocates nothing and builds no descriptors."""

    def __init__(self, eng, B, Cin, H, W, device, groups=1):
        P = eng.planes
        self.key = (B, Cin, H, W, P, groups)
        self.B, self.in_shape, self.dev = B, (B, Cin, H, W), device
        self.groups, self.Bg = groups, B // groups              # `groups` runs of Bg images, each with its own batch statistics
        mk = lambda ncg, h, w: stacked_at(P, B, ncg, h, w, device) if h <= STACK_MAX else new_at(P, B, ncg, h, w, device)
        self.t0, self.ut0, self.dx0 = mk((Cin + 7) // 8, H, W), mk((Cin + 7) // 8, H, W), mk((Cin + 7) // 8, H, W)
        self.y, self.z, self.dy, self.gdy, self.g_y, self.dz, self.g_dz, self.s2d, self.hw = [], [], [], [], [], [], [], [], []
        h, w = H, W
        n = len(eng.layers)
        for i, L in enumerate(eng.layers):
            if L.strided:
                if h % 2 or w % 2:
                    raise EsrError('critic: odd feature-map size %dx%d in front of a stride-2 conv' % (h, w))
                h, w = h // 2, w // 2
            s2d = i + 1 < n and eng.layers[i + 1].strided
            cg = L.cout // 8
            zg, zh, zw = (cg * 4, h // 2, w // 2) if s2d else (cg, h, w)
            self.y.append(mk(cg, h, w)); self.dy.append(mk(cg, h, w)); self.gdy.append(mk(cg, h, w)); self.g_y.append(mk(cg, h, w))
            self.z.append(mk(zg, zh, zw)); self.dz.append(mk(zg, zh, zw)); self.g_dz.append(mk(zg, zh, zw))
            self.s2d.append(s2d); self.hw.append((h, w))
        self.feat_shape = (B, eng.layers[-1].cout, h, w)
        # split-K workspace: room for 8 slabs of the largest deep-layer output (or input gradient), capped
        self.ksw = None
        if SPLITK:
            # (a slab is B*C*H*W floats; numel / planes also counts the border pixels: an upper bound)
            nch = lambda t: 8 * (t.shape[1] if getattr(t, '_esr_stacked', False) else t.shape[2])
            need = max([8 * (t.numel() // P) for t in self.y + self.dz if nch(t) >= 256] + [0])
            if need:
                self.ksw = torch.empty(min(need, SPLITK_MAX_FLOATS), dtype=torch.float32, device=device)
        # per-channel sums / statistics: [fwd: sums(2 doubles) mean rstd scale shift | bwd: sums2 | bwd2: sums3] per normalised block
        self.st = []
        off = 0
        take = lambda nb: (off_box.__setitem__(0, off_box[0] + (nb + 255) // 256 * 256), off_box[0] - (nb + 255) // 256 * 256)[1]
        off_box = [0]
        lay = []
        for L in eng.layers:
            if L.bn is None:
                lay.append(None)
                continue
            Cc = L.cout * groups                                # every array is [groups][C]
            lay.append(dict(sums=take(Cc * 16), mean=take(Cc * 4), rstd=take(Cc * 4), scale=take(Cc * 4), shift=take(Cc * 4)))
        self.fwd_zero = (0, off_box[0])
        b0 = off_box[0]
        for L, d in zip(eng.layers, lay):
            if d is not None:
                d['sums2'] = take(L.cout * groups * 16)
        self.bwd_zero = (b0, off_box[0] - b0)
        b1 = off_box[0]
        for L, d in zip(eng.layers, lay):
            if d is not None:
                d['sums3'] = take(L.cout * groups * 24)
        self.bwd2_zero = (b1, off_box[0] - b1)
        self.scratch = torch.zeros(max(off_box[0], 256), dtype=torch.uint8, device=device)
        self.lay = lay
        self.eval_affine = {}            # eval-mode BatchNorm: per block (scale, shift) tensors owned by the set (static addresses)
        self.plans, self.plans_fp = {}, None
        self.wg = {}

    _PER_CH = dict(sums=16, mean=4, rstd=4, scale=4, shift=4, sums2=16, sums3=24)

    def ptr(self, i, name, g=0, eng=None):
        """Address of block i's array `name` ([groups][C] ...), at group g."""
        off = g * eng.layers[i].cout * self._PER_CH[name] if g else 0
        return self.scratch.data_ptr() + self.lay[i][name] + off

    def stats(self, eng, i, training, g=0):
        """Statistics pointers of block i, starting at group g (a launch over ONE group passes g and groups = 1)."""
        L = eng.layers[i]
        st = _Stats()
        st.sums2 = st.sums3 = None
        if L.bn is None:
            return st
        st.gamma = L.bn.weight.data_ptr() if L.bn.weight is not None else None
        if training:
            st.const = False
            st.mean, st.rstd, st.scale, st.shift = (self.ptr(i, k, g, eng) for k in ('mean', 'rstd', 'scale', 'shift'))
            st.sums2, st.sums3 = self.ptr(i, 'sums2', g, eng), self.ptr(i, 'sums3', g, eng)
        else:
            sc, sh = self.eval_affine[i]
            st.scale, st.shift = sc.data_ptr(), sh.data_ptr()
        return st


class _State:
    pass


def _desc(L, B, y, st, s2d, dz=None, u=None, out0=None, out1=None, groups=1, b0=0):
    """BatchNorm launch over images [b0, b0 + B) as `groups` runs with their own statistics (st: pointers at the first of those groups)."""
    d = BnDesc()
    d.y = view_of(y, b0=b0)
    if dz is not None:
        d.dz = view_of(dz, b0=b0)
    if u is not None:
        d.u = view_of(u, b0=b0)
    if out0 is not None:
        d.out0 = view_of(out0, b0=b0)
    if out1 is not None:
        d.out1 = view_of(out1, b0=b0)
    d.B, d.groups, d.C = B, groups, L.cout
    d.scale, d.shift, d.mean, d.rstd, d.gamma = st.scale, st.shift, st.mean, st.rstd, st.gamma
    d.sums2, d.sums3 = st.sums2, st.sums3
    d.slope, d.const_stats, d.s2d = SLOPE, 1 if st.const else 0, 1 if s2d else 0
    return d


def _emit_bn(rec, op, d, mode, sums=None):
    rec.emit(op, _lib.CmdBn(d, mode, sums))


def _zero_region(rec, bs, region):
    off, nbytes = region
    if nbytes:
        rec.emit(_lib.OP_ZERO, _lib.CmdZero(bs.scratch.data_ptr() + off, nbytes // 16))


def _acquire(eng, x, groups=1):
    B, Cin, H, W = x.shape
    if B % groups:
        raise EsrError('critic: %d images do not split into %d equal groups' % (B, groups))
    key = (B, Cin, H, W, eng.planes, groups)
    free = eng._free_sets.setdefault(key, [])
    bs = free.pop() if free else _BufSet(eng, B, Cin, H, W, x.device, groups)
    fp = eng.pointer_fingerprint()
    if bs.plans_fp != fp:
        bs.plans, bs.plans_fp, bs.wg = {}, fp, {}
    return bs


def _release(eng, bs):
    free = eng._free_sets.setdefault(bs.key, [])
    if len(free) < 8:
        free.append(bs)


def _replay(bs, key, ext, build):
    plan = bs.plans.get(key)
    if plan is None:
        rec = A.Recorder(ext)
        with A.recording(rec):
            build(rec)
        plan = bs.plans[key] = rec.finish()
    plan.run(ext)


def _fwd_pass(eng, x, training, groups=1):
    """-> (features fp32 [B, C, h, w], state).  One launch list: pack, then per block conv -> statistics -> normalise + activate.  groups > 1:
    x is `groups` batches back to back, each normalised with its own batch statistics (the critic's separate calls as one pass; the running
    statistics are updated group after group, as the separate calls would)."""
    x = x.float().contiguous()
    bs = _acquire(eng, x, groups)
    B, G = bs.B, bs.groups
    S = _State()
    S.bs, S.training = bs, training
    weakref.finalize(S, _release, eng, bs)
    feat = torch.empty(bs.feat_shape, dtype=torch.float32, device=x.device)
    tracked = []
    if not training:
        for i, L in enumerate(eng.layers):
            if L.bn is None:
                continue
            bn = L.bn
            rstd = torch.rsqrt(bn.running_var.float() + bn.eps)
            sc = (bn.weight.detach().float() if bn.weight is not None else torch.ones_like(rstd)) * rstd
            sh = (bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(rstd)) - sc * bn.running_mean.float()
            if i not in bs.eval_affine:
                bs.eval_affine[i] = (torch.empty_like(sc), torch.empty_like(sh))
            bs.eval_affine[i][0].copy_(sc); bs.eval_affine[i][1].copy_(sh)
    else:
        tracked = [L.bn.num_batches_tracked for L in eng.layers if L.bn is not None and L.bn.track_running_stats and L.bn.num_batches_tracked is not None]

    def build(rec):
        Cin = bs.in_shape[1]
        _zero_region(rec, bs, bs.fwd_zero)
        A.pack_nchw(x, view_of(bs.t0), 0, Cin)
        t = bs.t0
        for i, L in enumerate(eng.layers):
            h, w = bs.hw[i]
            y, z = bs.y[i], bs.z[i]
            kw = dict(tap_mask_k=MASK_FWD, tap_mask_k_shift=1) if (L.strided and MASK_FWD) else {}
            vi, vo, Bc, hc, wc = conv_io(t, y, B, h, w)
            A.conv3x3(L.fwd, vi, Bc, hc, wc, L.cout, out=vo, reverse=False, k_split_ws=bs.ksw, **kw)
            st = bs.stats(eng, i, training)
            if L.bn is not None and training:
                # statistics, then ONE launch that turns the sums into the affine (and the stored / running statistics) and applies it
                bn = L.bn
                _emit_bn(rec, _lib.OP_BN_REDUCE, _desc(L, B, y, st, False, groups=G), 0, bs.ptr(i, 'sums'))
                track = bn.track_running_stats and bn.running_mean is not None
                fin = _lib.CmdBnFinalize(bs.ptr(i, 'sums'), G, L.cout, bs.Bg * h * w, bn.eps, bn.momentum if bn.momentum is not None else 0.1,
                                         st.gamma, bn.bias.data_ptr() if bn.bias is not None else None, st.mean, st.rstd, st.scale, st.shift,
                                         bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None)
                if FUSE_FINALIZE:
                    rec.emit(_lib.OP_BN_FINALIZE_APPLY, _lib.CmdBnFinalizeApply(_desc(L, B, y, st, bs.s2d[i], out0=z, groups=G), fin))
                    t = z
                    continue
                rec.emit(_lib.OP_BN_FINALIZE, fin)
            _emit_bn(rec, _lib.OP_BN_APPLY, _desc(L, B, y, st, bs.s2d[i], out0=z, groups=G), 0)
            t = z
        rec.emit(_lib.OP_UNPACK_NCHW, _lib.CmdUnpackNchw(view_of(t), B, bs.feat_shape[1], feat.data_ptr()), ('dst',))
    _replay(bs, ('fwd', training), {'x': x, 'feat': feat}, build)
    if tracked:
        torch._foreach_add_(tracked, G)
    return feat, S


class _WgradSet:
    """The weight-gradient launch of one pass kind over one buffer set: descriptors built once; per call a fresh zeroed flat dW buffer, the
    table re-pointed only if its address moved (engine.WGrad.rebind's scheme)."""

    def __init__(self, eng, bs, pairs, b0=0, nb=None):
        self.eng, self.pairs = eng, pairs
        nb = bs.B - b0 if nb is None else nb                       # the images the sum runs over: [b0, b0 + nb)
        self.sizes = [L.cout * L.cin_e * 9 + L.cout for L, _, _ in pairs]
        self.n = sum(self.sizes)
        flat = torch.zeros(self.n, dtype=torch.float32, device=bs.dev)
        descs, off = [], 0
        for (L, dy, xin), n in zip(pairs, self.sizes):
            nw = L.cout * L.cin_e * 9
            vx, vdy, Bc, hc, wc = conv_io(xin, dy, nb, dy.shape[3] - 2, dy.shape[4] - 2, b0)     # (stacked maps: one tall image; their borders are zero)
            d, _, _ = A.wgrad_desc(vdy, vx, None, 0, (L.cout, L.cin_e, 3, 3), Bc, hc, wc, 1.0, 1, bs.dev,
                                   out=(flat[off:off + nw], flat[off + nw:off + n]), tap_masks=MASK_FWD if (L.strided and MASK_FWD) else None)
            descs.append(d)
            off += n
        self.arr = (_lib.WgradDesc * len(descs))(*descs)
        self.flat_ptr = flat.data_ptr()
        need = _lib.lib.esr_conv3x3_wgrad_batch_workspace_bytes(self.arr, len(descs))
        check(min(need, 0), 'esr_conv3x3_wgrad_batch_workspace_bytes')
        self.ws = torch.empty(int(need), dtype=torch.uint8, device=bs.dev)
        self.plan = None
        self._first = flat
        # the embedded stride-2 layers' 4x4 gradients are gathered out of their 3x3 x 4 cin blocks by ONE index launch for all of them
        idx, self.gather, off, g0 = [], {}, 0, 0
        for (L, _, _), n in zip(pairs, self.sizes):
            if L.strided:
                idx.append(L.E_index + off)
                self.gather[L.index] = (g0, g0 + L.E_index.numel())
                g0 += L.E_index.numel()
            off += n
        self.gather_idx = torch.cat(idx) if idx else None

    def run(self):
        """-> {layer index: (dW in the parameter's shape, db)}"""
        flat, self._first = (self._first, None) if self._first is not None else (torch.zeros(self.n, dtype=torch.float32, device=self.ws.device), None)
        delta = flat.data_ptr() - self.flat_ptr
        if self.plan is None:
            self.plan = _lib.WgradBatchPlan()
            check(_lib.lib.esr_conv3x3_wgrad_batch_upload(self.arr, len(self.arr), self.ws.data_ptr(), self.ws.numel(), C.byref(self.plan), A.stream_ptr()),
                  'esr_conv3x3_wgrad_batch_upload')
        elif delta:            # a new flat buffer: move the table's dW / db pointers on the device (no host copy)
            check(_lib.lib.esr_conv3x3_wgrad_batch_rebase(self.ws.data_ptr(), C.byref(self.plan), delta, A.stream_ptr()), 'esr_conv3x3_wgrad_batch_rebase')
            self.flat_ptr += delta
        check(_lib.lib.esr_conv3x3_wgrad_batch_run(self.ws.data_ptr(), C.byref(self.plan), A.stream_ptr()), 'esr_conv3x3_wgrad_batch_run')
        out, off = {}, 0
        picked = flat[self.gather_idx] if self.gather_idx is not None else None
        for (L, _, _), n in zip(self.pairs, self.sizes):
            nw = L.cout * L.cin_e * 9
            dw, db = flat[off:off + nw], flat[off + nw:off + n]
            dw = picked[self.gather[L.index][0]:self.gather[L.index][1]].view(L.cout, L.cin, 4, 4) if L.strided else dw.view(L.cout, L.cin_e, 3, 3)
            out[L.index] = (dw, db)
            off += n
        return out


def _bwd_pass(eng, S, d_feat, g_ys, want_dx, want_params, group=None):
    """The backward pass as one launch list: per block (last to first) BatchNorm/LeakyReLU gradient -> [+ injected cotangent of y_l] -> data
    gradient.  dy_l / dz_l stay in the buffer set (the double backward reads them).  -> (d input fp32 or None, {layer: (dW, db)}, {layer:
    (dgamma, dbeta)}).  group = g (a grouped forward): only the g-th input's images are processed — the caller promises that d_feat is zero
    elsewhere; the other rows of d input come back zero."""
    bs, training = S.bs, S.training
    dev, n = bs.dev, len(eng.layers)
    if group is not None and bs.groups == 1:
        group = None
    G = bs.groups if group is None else 1                        # statistic groups of this pass's launches
    g0 = 0 if group is None else group
    b0, B = g0 * bs.Bg, (bs.B if group is None else bs.Bg)        # its images: [b0, b0 + B)
    Bg = bs.Bg
    bnl = [i for i, L in enumerate(eng.layers) if L.bn is not None and training]
    pg = torch.empty(2 * sum(eng.layers[i].cout for i in bnl), dtype=torch.float32, device=dev) if (want_params and bnl) else None
    inj = tuple(g is not None for g in g_ys) if g_ys is not None else (False,) * n
    if g_ys is not None:
        for i, g in enumerate(g_ys):         # the cotangents of the y_l come out of this set's own double-backward pass
            if g is not None and g.data_ptr() != bs.g_y[i].data_ptr():
                bs.g_y[i].copy_(g)
    ext = {}
    if d_feat is not None:
        d_feat = d_feat.detach().float().contiguous()[b0:b0 + B]
        ext['d_feat'] = d_feat
    dx_full = (torch.empty if group is None else torch.zeros)(bs.in_shape, dtype=torch.float32, device=dev) if want_dx else None
    dx_in = dx_full[b0:b0 + B] if want_dx else None
    if dx_in is not None:
        ext['dx_in'] = dx_in
    # the images whose y_l cotangents exist: those the double backward pass ran on (it leaves the range on the state), inside this pass's range
    ib0, inb = getattr(S, 'inj_rows', None) or (0, bs.B)
    if group is not None:
        ib0, inb = b0, B
    if pg is not None:
        ext['pg'] = pg
    bn_grads, pg_off = {}, 0
    for i in reversed(bnl):
        if pg is not None:
            c = eng.layers[i].cout
            bn_grads[i] = (pg[pg_off:pg_off + c], pg[pg_off + c:pg_off + 2 * c])
            pg_off += 2 * c

    def build(rec):
        # (a pass over one group leaves the other groups' sums alone: the double backward of another group may still need them)
        if group is None:
            _zero_region(rec, bs, bs.bwd_zero)
        else:
            for i in bnl:
                rec.emit(_lib.OP_ZERO, _lib.CmdZero(bs.ptr(i, 'sums2', g0, eng), eng.layers[i].cout))        # C x 2 doubles = C 16-byte vectors
        dz = bs.dz[n - 1]
        if d_feat is None:
            rec.emit(_lib.OP_ZERO, _lib.CmdZero(dz.data_ptr(), dz.numel() // 8))
        else:
            A.pack_nchw(d_feat, view_of(dz, b0=b0), 0, d_feat.shape[1])
        for i in reversed(range(n)):
            L, y, dy, dz = eng.layers[i], bs.y[i], bs.dy[i], bs.dz[i]
            h, w = bs.hw[i]
            st = bs.stats(eng, i, training, g0)
            if not st.const:
                _emit_bn(rec, _lib.OP_BN_REDUCE, _desc(L, B, y, st, bs.s2d[i], dz=dz, groups=G, b0=b0), 1, st.sums2)
                if pg is not None:
                    rec.emit(_lib.OP_BN_PARAM_GRADS, _lib.CmdBnParamGrads(st.sums2, None, None, G, L.cout, Bg * h * w, bn_grads[i][0].data_ptr(),
                                                                          bn_grads[i][1].data_ptr(), None), ('dgamma', 'dbeta'))
            _emit_bn(rec, _lib.OP_BN_APPLY, _desc(L, B, y, st, bs.s2d[i], dz=dz, out0=dy, groups=G, b0=b0), 1)
            if inj[i]:
                A.act_combine(view_of(dy, b0=ib0), inb, A_=view_of(dy, b0=ib0), alpha=1.0, Bv=view_of(bs.g_y[i], b0=ib0), beta=1.0, s=1)
            if i > 0 or want_dx:
                dx = bs.dz[i - 1] if i > 0 else bs.dx0
                kw = dict(tap_mask_m=MASK_FLIPPED) if (L.strided and MASK_FLIPPED) else {}
                vi, vo, Bc, hc, wc = conv_io(dy, dx, B, h, w, b0)
                A.conv3x3(L.tr, vi, Bc, hc, wc, L.cin_e, out=vo, use_bias=False, reverse=False, k_split_ws=bs.ksw, **kw)
        if want_dx:
            rec.emit(_lib.OP_UNPACK_NCHW, _lib.CmdUnpackNchw(view_of(bs.dx0, b0=b0), B, bs.in_shape[1], dx_in.data_ptr()), ('dst',))
    _replay(bs, ('bwd', training, d_feat is not None, inj, want_dx, pg is not None, group, (ib0, inb)), ext, build)
    conv_grads = {}
    if want_params:
        wk = ('bwd', group)
        if wk not in bs.wg:
            bs.wg[wk] = _WgradSet(eng, bs, [(L, bs.dy[i], bs.z[i - 1] if i > 0 else bs.t0) for i, L in enumerate(eng.layers)], b0, B)
        conv_grads = bs.wg[wk].run()
    return dx_full, conv_grads, bn_grads


def _bwd2_pass(eng, S, u, want_params, group=None):
    """The backward of the backward pass (u: cotangent of d input), first block to last: per block conv of the incoming cotangent -> gradient
    of the BatchNorm/LeakyReLU gradient.  -> (cotangent of d features, [cotangent of y_l], {layer: dW (second order)}, {layer: g_gamma}).
    group = g: the backward pass it differentiates ran on the g-th input's images only (and u is zero elsewhere): so does this one."""
    bs, training = S.bs, S.training
    dev, n = bs.dev, len(eng.layers)
    if group is not None and bs.groups == 1:
        group = None
    G = bs.groups if group is None else 1
    g0 = 0 if group is None else group
    b0, B = g0 * bs.Bg, (bs.B if group is None else bs.Bg)
    Bg = bs.Bg
    bnl = [i for i, L in enumerate(eng.layers) if L.bn is not None and training and L.bn.weight is not None]
    gg_all = torch.empty(sum(eng.layers[i].cout for i in bnl), dtype=torch.float32, device=dev) if (want_params and bnl) else None
    u = u.detach().float().contiguous()[b0:b0 + B]
    g_dfeat_full = (torch.empty if group is None else torch.zeros)(bs.feat_shape, dtype=torch.float32, device=dev)
    g_dfeat = g_dfeat_full[b0:b0 + B]
    ext = {'u': u, 'g_dfeat': g_dfeat}
    g_gammas, off = {}, 0
    if gg_all is not None:
        ext['gg'] = gg_all
        for i in bnl:
            g_gammas[i] = gg_all[off:off + eng.layers[i].cout]
            off += eng.layers[i].cout

    def build(rec):
        if group is None:
            _zero_region(rec, bs, bs.bwd2_zero)
        else:
            for i, L in enumerate(eng.layers):
                if L.bn is not None and training:
                    rec.emit(_lib.OP_ZERO, _lib.CmdZero(bs.ptr(i, 'sums3', g0, eng), (L.cout * 24 + 15) // 16))
        A.pack_nchw(u, view_of(bs.ut0, b0=b0), 0, u.shape[1])
        ut = bs.ut0
        for i, L in enumerate(eng.layers):
            y, gdy = bs.y[i], bs.gdy[i]
            h, w = bs.hw[i]
            st = bs.stats(eng, i, training, g0)
            kw = dict(tap_mask_k=MASK_FWD, tap_mask_k_shift=1) if (L.strided and MASK_FWD) else {}
            vi, vo, Bc, hc, wc = conv_io(ut, gdy, B, h, w, b0)
            A.conv3x3(L.fwd, vi, Bc, hc, wc, L.cout, out=vo, use_bias=False, reverse=False, k_split_ws=bs.ksw, **kw)
            if not st.const:
                _emit_bn(rec, _lib.OP_BN_REDUCE, _desc(L, B, y, st, bs.s2d[i], dz=bs.dz[i], u=gdy, groups=G, b0=b0), 2, st.sums3)
                if i in g_gammas:
                    rec.emit(_lib.OP_BN_PARAM_GRADS, _lib.CmdBnParamGrads(st.sums2, st.sums3, st.rstd, G, L.cout, Bg * h * w, None, None, g_gammas[i].data_ptr()),
                             ('g_gamma',))
            _emit_bn(rec, _lib.OP_BN_APPLY, _desc(L, B, y, st, bs.s2d[i], dz=bs.dz[i], u=gdy, out0=bs.g_dz[i], out1=bs.g_y[i], groups=G, b0=b0), 2)
            ut = bs.g_dz[i]
        rec.emit(_lib.OP_UNPACK_NCHW, _lib.CmdUnpackNchw(view_of(ut, b0=b0), B, bs.feat_shape[1], g_dfeat.data_ptr()), ('dst',))
    _replay(bs, ('bwd2', training, gg_all is not None, group), ext, build)
    S.inj_rows = (b0, B)                # only these images' g_y are defined: the backward pass that takes them in adds them there
    g_ys = [bs.g_y[i].detach() if (eng.layers[i].bn is not None and training) else None for i in range(n)]
    conv2 = {}
    if want_params:
        wk = ('bwd2', group)
        if wk not in bs.wg:
            bs.wg[wk] = _WgradSet(eng, bs, [(L, bs.dy[i], bs.ut0 if i == 0 else bs.g_dz[i - 1]) for i, L in enumerate(eng.layers)], b0, B)
        conv2 = {k: v[0] for k, v in bs.wg[wk].run().items()}
    return g_dfeat_full, g_ys, conv2, g_gammas


def _param_list(eng):
    """[(conv.weight, conv.bias, bn.weight or None, bn.bias or None)] flattened, None entries kept (positions are fixed: 4 per block)."""
    out = []
    for L in eng.layers:
        out += [L.conv.weight, L.conv.bias, L.bn.weight if L.bn is not None else None, L.bn.bias if L.bn is not None else None]
    return out


class _CriticFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, training, groups, x, *params):
        ctx.set_materialize_grads(False)
        feat, S = _fwd_pass(eng, x.detach(), training, groups)
        ctx.eng, ctx.S, ctx.np = eng, S, len(params)
        ys = tuple(y.detach() for y in S.bs.y)        # fresh tensor objects over the set's storage: autograd attaches its history to these
        ctx.save_for_backward(*[p for p in params if p is not None], *ys)
        ctx.pmask = [p is not None for p in params]
        return (feat,) + ys

    @staticmethod
    def backward(ctx, d_feat, *g_ys):
        saved = list(ctx.saved_tensors)
        nreal = sum(ctx.pmask)
        it = iter(saved[:nreal])
        params = [next(it) if m else None for m in ctx.pmask]
        ys = saved[nreal:]
        want_dx = ctx.needs_input_grad[3]
        igo, hint_group = _hints_for(ctx.S)
        want_params = any(ctx.needs_input_grad[4:]) and not igo
        group = hint_group if ctx.S.bs.groups > 1 else None
        outs = _CriticBwd.apply(ctx.eng, ctx.S, want_dx, want_params, group, d_feat, len(ys), *g_ys, *ys, *params)
        dx, pgrads = outs[0], outs[1:]
        return (None, None, None, dx if want_dx else None) + tuple(g if (g is not None and ctx.needs_input_grad[4 + k]) else None for k, g in enumerate(pgrads))


class _CriticBwd(torch.autograd.Function):
    """(d input, parameter gradients) = backward pass of the critic, as a function of (d features, cotangents injected at the y_l, the y_l
    themselves, the parameters): differentiable once more (WGAN-GP)."""
    NFIXED = 7            # eng, S, want_dx, want_params, group, d_feat, n

    @staticmethod
    def forward(ctx, eng, S, want_dx, want_params, group, d_feat, n, *rest):
        ctx.set_materialize_grads(False)
        g_ys, params = rest[:n], rest[2 * n:]
        dx, conv_grads, bn_grads = _bwd_pass(eng, S, d_feat, g_ys if any(g is not None for g in g_ys) else None, want_dx, want_params, group)
        ctx.eng, ctx.S, ctx.n, ctx.nparams, ctx.group = eng, S, n, len(params), group
        ctx.had_dfeat = d_feat is not None
        pg = []
        for i, L in enumerate(eng.layers):
            cw, cb = conv_grads.get(i, (None, None))
            bg, bb = bn_grads.get(i, (None, None))
            pg += [cw, cb, bg if L.bn is not None and L.bn.weight is not None else None, bb if L.bn is not None and L.bn.bias is not None else None]
        ctx.mark_non_differentiable(*[g for g in pg if g is not None])
        if dx is None:
            dx = torch.zeros((), device=S.bs.dev)             # placeholder output (never used: the input asked for no gradient)
            ctx.mark_non_differentiable(dx)
        return (dx,) + tuple(pg)

    @staticmethod
    def backward(ctx, u, *u_params):
        n, nf = ctx.n, _CriticBwd.NFIXED
        if u is None:
            return (None,) * (nf + 2 * n + ctx.nparams)
        want_params = any(ctx.needs_input_grad[nf + 2 * n:]) and not _hints_for(ctx.S)[0]
        g_dfeat, g_ys, conv2, g_gammas = _bwd2_pass(ctx.eng, ctx.S, u, want_params, ctx.group)
        pg = []
        for i, L in enumerate(ctx.eng.layers):
            pg += [conv2.get(i), None, g_gammas.get(i), None]
        return (None,) * (nf - 2) + (g_dfeat if ctx.had_dfeat else None, None) + (None,) * n + tuple(g_ys) + tuple(pg)


FUSED = True        # False: one autograd node and several FFI calls per layer (the readable definition; cross-checked by the tests)
_critic_forward_per_layer = critic_forward


def critic_forward(eng, x):
    """Logits [B, 1] of the critic for fp32 NCHW images `x`, differentiable to the order the WGAN-GP step needs.  Three launch lists per
    forward / backward / double backward (FUSED = False: one autograd node and several FFI calls per layer — the same kernels)."""
    if not FUSED:
        return _critic_forward_per_layer(eng, x)
    return critic_forward_group(eng, [x])[0]


GROUPED = True      # False: a grouped call runs its batches one by one


def critic_forward_group(eng, xs):
    """[critic(x) for x in xs] for equally shaped batches — the same values as separate calls in this order (each batch is normalised with
    its own batch statistics, the running statistics see the batches one after the other) — executed as ONE pass over the concatenated
    images: the critic's 512-channel layers on 8x8 / 4x4 maps cost the same for 32 or 96 images, the weight gradients of all batches
    become one launch, and autograd has one graph to walk.  The WGAN-GP step calls it with [real, fake, interpolated]
    (models/SRRaGAN_model.py); `input_grad_only(group=2)` around the penalty's autograd.grad keeps that pass on the interpolated images."""
    xs = list(xs)
    if not FUSED or not GROUPED or any(x.shape != xs[0].shape for x in xs):
        return [critic_forward(eng, x) for x in xs]
    for x in xs:
        A.require_gpu(x, 'critic input')
    net = eng.net
    eng.refresh()
    if xs[0].shape[1] != eng.layers[0].cin:
        raise EsrError('critic input: %d channels expected' % eng.layers[0].cin)
    x = xs[0] if len(xs) == 1 else torch.cat(xs)
    outs = _CriticFwd.apply(eng, net.training, len(xs), x, *_param_list(eng))
    feat = outs[0]
    logits = net.classifier(feat.reshape(feat.size(0), -1))
    res = [logits] if len(xs) == 1 else list(logits.chunk(len(xs)))
    S = getattr(feat.grad_fn, 'S', None)              # the call's state (ctx of _CriticFwd): lets input_grad_only(of=logits) address this graph alone
    for t in res:
        t._esr_critic_state = S
    return res
