"""Launch planner for the whole RRDB generator (reference: RRDBNet.forward, codes/models/modules/architecture.py:278-302
over block.py's RRDB / ResidualDenseBlock_5C / ShortcutBlock / upconv blocks).

What the reference does with ~10 torch ops per conv layer (cat, conv, leaky_relu_, mul, add, interpolate) is planned here
as ONE esr_conv3x3 launch per conv layer:
  * dense-block concatenation is zero-copy: each RDB owns one 24-group buffer [x | x1 | x2 | x3 | x4]; conv i reads the
    first 8+4i groups and writes groups 8+4i..; the latent Z is a separate 1-group buffer passed as segment 0
  * bias, LeakyReLU, the RDB residual (0.2*y + x), the RRDB residual (0.2*(.) + x_rrdb) and the trunk shortcut are
    epilogue terms of the producing conv; conv5 writes straight into the next RDB's buffer
  * nearest-neighbour upsampling is folded into the input read of the following conv
  * the CEM eval-mode replicate padding and the latent's bilinear /sf are folded into the input packing kernel
Buffers are allocated once per (batch, height, width) and reused; for inference three RDB buffers rotate.
"""
import torch

from . import act as A
from ._lib import EsrError


class RRDBEngine:
    def __init__(self, net):
        self.net = net
        self.split = True
        self._packed = None
        self._bufs = {}
        self._ev = None      # optional (start, end) torch.cuda.Event pair bracketing the conv launches of one forward (bench.py)

    def set_precision(self, precision):
        split = precision == 'split'
        if split != self.split:
            self.split = split
            self._packed = None
            self._bufs = {}

    # ------------------------------------------------------------------ weights
    def _convs(self):
        """(name, conv module, n_latent) in execution order, following the reference's module tree."""
        net = self.net
        m = net.model
        lat_first, lat = net.num_latent_channels if net.latent_input is not None else 0, net._lat_all_layers
        out = [('fea', m[0], lat_first)]
        sub = m[1].sub
        for r in range(net.nb):
            for k, rdb in enumerate((sub[r].RDB1, sub[r].RDB2, sub[r].RDB3)):
                for i in range(5):
                    out.append(('rrdb%d.rdb%d.conv%d' % (r, k, i), rdb.convs[i][0], lat))
        out.append(('lr_conv', sub[net.nb], lat))
        idx = 2
        n_up = 1 if net.upscale == 3 else len([1 for mod in m if isinstance(mod, torch.nn.Sequential)])
        for j in range(n_up):
            out.append(('up%d' % j, m[idx][1] if net.upsample_mode == 'upconv' else m[idx][0], 0))
            idx += 1
        lat_hr = lat
        out.append(('hr0', m[idx], lat_hr))
        out.append(('hr1', m[idx + 2], lat_hr))
        self.n_up = n_up
        return out

    def packed(self):
        if self._packed is None:
            self._packed = {name: A.PackedConv(c.weight, c.bias, lat, split=self.split) for name, c, lat in self._convs()}
        for p in self._packed.values():
            p.get()
        return self._packed

    # ------------------------------------------------------------------ buffers
    def _buffers(self, B, h, w, dev):
        key = (B, h, w, str(dev))
        if key in self._bufs:
            return self._bufs[key]
        if len(self._bufs) > 2:
            self._bufs.clear()
        net, sp = self.net, self.split
        sf = net.upscale
        d = {}
        has_lat = net.latent_input is not None and net.num_latent_channels > 0
        if has_lat:
            d['zlr'] = A.ActBuf(B, 1, h, w, dev, sp)
            if net._lat_all_layers:
                d['zhr'] = A.ActBuf(B, 1, sf * h, sf * w, dev, sp)
        d['xin'] = A.ActBuf(B, 1, h, w, dev, sp)
        d['fea'] = A.ActBuf(B, 8, h, w, dev, sp)
        d['rdb'] = [A.ActBuf(B, 24, h, w, dev, sp) for _ in range(3)]
        d['trunk'] = A.ActBuf(B, 8, h, w, dev, sp)
        ups = []
        s = 1
        for j in range(self.n_up):
            s *= 3 if sf == 3 else 2
            ups.append(A.ActBuf(B, 8, s * h, s * w, dev, sp))
        d['ups'] = ups
        d['hr0'] = A.ActBuf(B, 8, sf * h, sf * w, dev, sp)
        self._bufs[key] = d
        return d

    # ------------------------------------------------------------------ forward
    def forward(self, x, pad=0):
        A.require_gpu(x, 'generator input')
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.net.parameters())):
            from . import autograd as AG
            return AG.rrdb_forward_with_grad(self, x, pad)
        return self.forward_nograd(x, pad)

    def forward_nograd(self, x, pad=0):
        net = self.net
        if net.upsample_mode != 'upconv':
            raise NotImplementedError("upsample_mode='pixelshuffle' is constructible (state_dict parity) but only 'upconv' — the mode the "
                                      "reference hard-wires for RRDB_net (networks.py:99) — is executed by the HIP engine")
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        sf = net.upscale
        has_lat = net.latent_input is not None and net.num_latent_channels > 0
        lat1 = net.num_latent_channels if has_lat else 0
        B, Ct, h0, w0 = x.shape
        if Ct != lat1 * sf * sf + 3:
            raise EsrError('expected %d input channels (latent %d x sf^2 + 3), got %d' % (lat1 * sf * sf + 3, lat1, Ct))
        pk = self.packed()
        h, w = h0 + 2 * pad, w0 + 2 * pad
        H, W = sf * h, sf * w
        bufs = self._buffers(B, h, w, x.device)
        conv = A.conv3x3

        # ---- input packing (+ replicate pad, + latent bilinear /sf)
        A.pack_nchw(x, bufs['xin'].view(), c0=Ct - 3, nc=3, pad=pad)
        zlr = zhr = None
        if has_lat:
            # raw view of the first lat*sf^2 channels as [lat][sf*h0][sf*w0] (SRRaGAN_model.py:233, architecture.py:283)
            kw = dict(hw=(sf * h0, sf * w0), batch_stride=Ct * h0 * w0, channels=lat1)
            A.pack_nchw(x, bufs['zlr'].view(), 0, lat1, pad=sf * pad, down=sf, **kw)
            zlr = bufs['zlr'].view()
            if net._lat_all_layers:
                A.pack_nchw(x, bufs['zhr'].view(), 0, lat1, pad=sf * pad, **kw)
                zhr = bufs['zhr'].view()
        zall = zlr if net._lat_all_layers else None

        if self._ev:
            self._ev[0].record()
        # ---- fea_conv -> fea (shortcut source) and the first RDB buffer
        rdb = bufs['rdb']
        conv(pk['fea'], bufs['xin'].view(), B, h, w, 64, in0=zlr, out=bufs['fea'].view(), out2=rdb[0].view(0, 8) if net.nb else None)
        cur = 0
        for r in range(net.nb):
            rrdb_in = rdb[cur]
            for k in range(3):
                buf = rdb[(cur + k) % 3]
                nxt = rdb[(cur + k + 1) % 3]
                for i in range(4):
                    conv(pk['rrdb%d.rdb%d.conv%d' % (r, k, i)], buf.view(0, 8 + 4 * i), B, h, w, 32, in0=zall, act_slope=0.2,
                         out=buf.view(8 + 4 * i, 4))
                name = 'rrdb%d.rdb%d.conv4' % (r, k)
                if k < 2:     # RDB output: 0.2*conv5 + x            (block.py:235)
                    conv(pk[name], buf.view(0, 24), B, h, w, 64, in0=zall, alpha=0.2, res1=buf.view(0, 8), beta1=1.0, out=nxt.view(0, 8))
                else:         # RRDB output: 0.2*(0.2*conv5 + x) + x_rrdb   (block.py:270); lands in the next RRDB's first buffer
                    conv(pk[name], buf.view(0, 24), B, h, w, 64, in0=zall, alpha=0.04, res1=buf.view(0, 8), beta1=0.2,
                         res2=rrdb_in.view(0, 8), beta2=1.0, out=nxt.view(0, 8))
            cur = (cur + 3) % 3
        last = rdb[cur].view(0, 8) if net.nb else bufs['fea'].view()
        # LR_conv + trunk shortcut (block.py:96)
        conv(pk['lr_conv'], last, B, h, w, 64, in0=zall, res1=bufs['fea'].view(), beta1=1.0, out=bufs['trunk'].view())
        # upsamplers: nearest xs folded into the conv's input read
        src, s = bufs['trunk'], 1
        for j in range(self.n_up):
            f = 3 if sf == 3 else 2
            s *= f
            conv(pk['up%d' % j], src.view(), B, s * h, s * w, 64, upsample=f, act_slope=0.2, out=bufs['ups'][j].view())
            src = bufs['ups'][j]
        conv(pk['hr0'], src.view(), B, H, W, 64, in0=zhr, act_slope=0.2, out=bufs['hr0'].view())
        g = torch.empty(B, net.out_nc, H, W, dtype=torch.float32, device=x.device)
        conv(pk['hr1'], bufs['hr0'].view(), B, H, W, net.out_nc, in0=zhr, out_nchw=g)
        if self._ev:
            self._ev[1].record()
        return g
