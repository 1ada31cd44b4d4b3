"""Launch planner for the whole RRDB generator (reference: RRDBNet.forward, codes/models/modules/architecture.py:278-302
over block.py's RRDB / ResidualDenseBlock_5C / ShortcutBlock / upconv blocks) and for its backward pass (autograd in the
reference).

What the reference does with ~10 torch ops per conv layer (cat, conv, leaky_relu_, mul, add, interpolate) is planned here
as ONE esr_conv3x3 launch per conv layer:
  * dense-block concatenation is zero-copy: each RDB owns one 24-group buffer [x | x1 | x2 | x3 | x4]; conv i reads the
    first 8+4i groups and writes groups 8+4i..; the latent Z is a separate 1-group buffer passed as segment 0
  * bias, LeakyReLU, the RDB residual (0.2*y + x), the RRDB residual (0.2*(.) + x_rrdb) and the trunk shortcut are
    epilogue terms of the producing conv; conv5 writes straight into the next RDB's buffer
  * nearest-neighbour upsampling is folded into the input read of the following conv
  * the CEM eval-mode replicate padding and the latent's bilinear /sf are folded into the input packing kernel
Inference rotates three RDB buffers; a forward that must be differentiated keeps every RDB buffer (they ARE the saved
activations) and the backward re-uses the same conv kernel with transposed/flipped weight packs (data gradient), a
pixel-contraction MFMA kernel (weight gradient) and the gradient buffers laid out exactly like the activations.
"""
import weakref

import torch

import ctypes as C

from . import _lib
from . import act as A
from ._lib import EsrError


class _Lease:
    pass


class RRDBEngine:
    def __init__(self, net):
        self.net = net
        self.split = True
        self._convs_cache = None
        self._params_cache, self._params_probe = None, None
        self._packed = None
        self._packed_t = None
        self._packed_rdb_t = None
        self._bufs = {}
        self._gpool, self._gpool_key = {}, None
        self._pack_batch = A.PackBatch()
        self._packs_fp = None
        self._pack_sets = 0  # bumped whenever a pack dict is created or dropped: part of the fingerprint (ids of dicts get recycled)
        self._pack_gen = [0, 0, 0]   # the value of _pack_sets when the forward / data-gradient / dense-block data-gradient dict was made
        self._wgb = {}       # this engine's uploaded weight-gradient descriptor tables (A.conv3x3_wgrad_batch)
        # launch lists (esr_run): a pass over a cached buffer set is recorded once and replayed with one C call per segment afterwards;
        # use_plans = False issues every launch from Python instead (the recording path itself, used by the tests as the reference)
        self.use_plans = True
        # what a differentiable forward keeps for its backward when NO parameter wants a gradient (the Z search: frozen generator): 'masks' —
        # three rotating dense-block buffers as in inference plus, per RDB, a ONE-plane copy of its four intermediate activations, of which the
        # data gradient reads nothing but the signs (LeakyReLU'); 'full' — every RDB's whole 24-group hi+lo buffer, as training needs (the
        # weight gradient contracts the activations themselves).  Same gradients bit for bit; 1/3 of the memory per RDB.
        self.stash = 'masks'
        # 'mixed' only: format of its backward pass, 'f16' (as the forward) or 'bf16' (hi+lo gradients, 3 MFMAs per product); see _bwd_split
        self.mixed_bwd = 'f16'
        # 'mixed' only: which dense-block convs multiply the residual stream's lo plane as well: 'none' (default), 'conv4' or 'all' 
        self.mixed_xlo = 'none'
        # 'mixed' backward: renormalise the fp16 gradient of every RRDB's input (False: one scale for the whole pass)
        self.grad_renorm = True
        self._ptr_fp, self._ptr_epoch = None, 0   # parameter storages the recorded descriptors point into; epoch moves when any changes
        self.generation = 0  # bumped by invalidate(): consumers that cache derived state (GraphedForward) compare it
        self._ev = None      # optional (start, end) torch.cuda.Event pair bracketing the conv launches of one forward (bench.py)
        # fp16 range watch (precisions 'f16', 'f16x2', 'mixed'): the forward's conv launches report the first layer whose stored activations
        # reach 2^15 or are not finite into one device word (A.RangeWatch); it is copied to pinned host memory behind every forward and looked at
        # without synchronising at the start of the next one — or, synchronising, by check_range().  The reference's fp32 path has no such cliff
        # (codes/models/modules/architecture.py:278-302): leaving the range raises EsrError naming the layer instead of returning inf / NaN images.
        self._watch, self._watch_host, self._watch_ev = None, None, None
        # Data parallelism (esr_hip/dist.py: EarlyBucketReducer): when set, the backward's batched weight-gradient launch is cut into one launch per
        # gradient bucket (contiguous runs of layers of ~bucket_bytes in the flat dW buffer), and `wgrad_exchange.start(i, flat[o0:o1])` is called
        # behind launch i — the bucket's all-reduce then runs on RCCL's stream under the launches of the buckets that follow; `.finish()` is called
        # behind the last one (it makes the compute stream wait for the collectives and scales the sums).  None (the default, and always for a
        # one-process job): ONE launch, nothing exchanged here.
        self.wgrad_exchange = None
        # Weight gradients UNDER the data-gradient chain (recorded backward passes, one-plane gradient formats): the layers are cut, in the order
        # their dy become final, into this many groups; every group but the last is launched on a second stream as soon as the main stream has
        # produced its last dy (esr_conv3x3_wgrad_batch_run_side: one workgroup per CU, so the chain's small launches keep finding room on every
        # CU), the last one behind the chain on the main stream, which then waits for the second.  Same slicing as the one launch: bit-identical
        # gradients.  0: one launch behind the chain.
        self.wgrad_overlap = WGRAD_OVERLAP
        self._side, self._xs = None, None
        self.n_up = 1 if net.upscale == 3 else len([1 for mod in net.model if isinstance(mod, torch.nn.Sequential)])
        # channel groups (8 channels each) of the residual stream (nf) and of a dense block's buffer [x (nf) | x1 | x2 | x3 | x4 (gc = 32 each)]:
        # 8 and 24 for the reference's nf = 64 (every shipped options file); nf = 16, 32, 48 run the same plan
        self.ng = net.nf // 8
        self.nd = self.ng + 16

    def set_precision(self, precision):
        # 'mixed': fp16; residual stream stored hi+lo; hi+lo weights x hi+lo activations (3 MFMAs) in the few layers outside the dense blocks,
        # whose rounding would dominate the output error; one-plane weights x hi planes (1 MFMA) in the dense-block convs
        split = {'split': True, 'bf16': False, 'f16': 'f16', 'f16x2': 'f16x2', 'mixed': 'mixed'}[precision]
        if split != self.split:
            self.split = split
            if self._watch is not None:        # a pending range verdict belongs to the precision that is being left
                self._watch.flag.fill_(-1)
                self._watch_host.fill_(-1)
                self._watch_ev = None
            self._packed = None
            self._packed_t = None
            self._packed_rdb_t = None
            self._bufs = {}
            self._packs_fp = None
            self._pack_sets += 1

    # ------------------------------------------------------------------ fp16 range watch
    def _watching(self, device):
        if self.split not in ('f16', 'f16x2', 'mixed'):
            return None
        if self._watch is None or self._watch.flag.device != device:
            self._watch = A.RangeWatch(device)
            self._watch_host = torch.full((1,), -1, dtype=torch.int32).pin_memory()
            self._watch_ev = None
        return self._watch

    def _post_range(self):
        """Behind a forward's launches: the flag word on its way to the host (asynchronous; nothing waits for it here)."""
        if self._watch is None or A._rec() is not None or torch.cuda.is_current_stream_capturing():
            return
        self._watch_host.copy_(self._watch.flag, non_blocking=True)
        self._watch_ev = torch.cuda.Event()
        self._watch_ev.record()

    def check_range(self, wait=True):
        """Raise EsrError if a forward since the last check stored fp16 activations of magnitude >= 2^15 (or inf / NaN).  wait=False looks only
        at what has already arrived on the host (no synchronisation: what the next forward does by itself); wait=True waits for the last
        forward's flag first — call it where a result is consumed (RRDBNet.forward callers that hand images on, bench.py after its timed loop)."""
        ev = self._watch_ev
        if self._watch is None or torch.cuda.is_current_stream_capturing():
            return
        if ev is None:
            # no copy is on its way: nothing ran since the last check — or what ran was a HIP-graph replay / a pass recorded into somebody else's
            # launch list (_post_range does not post from inside those).  wait=True reads the device word itself (one synchronising 4-byte copy)
            if not wait:
                return
            self._watch_host.copy_(self._watch.flag)
        elif wait:
            ev.synchronize()
        elif not ev.query():
            return
        self._watch_ev = None
        tag = int(self._watch_host[0])
        if tag == -1:
            return
        names = self._watch.names
        name = names[tag] if 0 <= tag < len(names) else 'launch %d' % tag
        self._watch.flag.fill_(-1)
        self._watch_host.fill_(-1)
        raise EsrError("precision %r: the activations stored by layer '%s' reached fp16's last binade (|x| >= 32768) or are not finite — the "
                       "fp16 modes cannot hold this network's range; use set_precision('split') (fp32-class) or 'bf16'" % (
                           self.split if isinstance(self.split, str) else 'split', name))

    # ------------------------------------------------------------------ weights
    def invalidate(self):
        """Force every weight pack to be rebuilt at the next forward / backward.  Packs are refreshed automatically when a parameter's
        (storage, torch version counter) changes — optimizer steps, load_state_dict, in-place ops on the parameter all bump the counter —
        but writes through `.data` (p.data.mul_(2), dist.broadcast(p.data), the reference's m.bias.data.zero_()) do NOT: after such an
        edit call this (RRDBNet.invalidate_packs(); esr_hip.dist.broadcast_parameters does)."""
        for d in (self._packed, self._packed_t, self._packed_rdb_t):
            for p in (d or {}).values():
                p._key = None
        self.generation += 1

    def _convs(self):
        """(name, conv module, n_latent) in execution order, following the reference's module tree (walked once: the tree is static)."""
        if self._convs_cache is None:
            self._convs_cache = self._walk_convs()
        return self._convs_cache

    def parameters(self):
        """The generator's parameters in execution order, without walking the module tree (700 of them, every forward).  The list is
        rebuilt only when a module's parameter object was replaced (rare: nn.Module.to() keeps the objects, assigning a new nn.Parameter
        does not) — checked through the conv modules' identity-stable (weight, bias) attributes of the first and last layer."""
        convs = self._convs()
        probe = (convs[0][1].weight, convs[-1][1].weight)
        if self._params_cache is None or self._params_probe[0] is not probe[0] or self._params_probe[1] is not probe[1]:
            self._params_cache = [p for _, c, _ in convs for p in (c.weight, c.bias) if p is not None]
            self._params_probe = probe
        return self._params_cache

    def _walk_convs(self):
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
        for j in range(self.n_up):
            out.append(('up%d' % j, m[idx][1] if net.upsample_mode == 'upconv' else m[idx][0], 0))
            idx += 1
        out.append(('hr0', m[idx], lat))
        out.append(('hr1', m[idx + 2], lat))
        return out

    def _refresh_packs(self):
        """Re-pack every existing pack (forward and data-gradient ones) in one launch if any parameter changed since the last time.
        Steady state is one tuple comparison: the (storage, version) fingerprint of the parameters and the identity of the pack set; the
        per-pack staleness keys are only consulted when that fingerprint moved."""
        dicts = (self._packed, self._packed_t, self._packed_rdb_t)
        fp = (tuple((p.data_ptr(), p._version) for p in self.parameters()), tuple(d is not None for d in dicts), self._pack_sets, self.generation)
        if fp == self._packs_fp:
            return
        ptrs = tuple(a for a, _ in fp[0])
        if ptrs != self._ptr_fp:              # a parameter moved (load into new storage, .to()): biases are read in place by the kernels
            self._ptr_fp, self._ptr_epoch = ptrs, self._ptr_epoch + 1
        # something changed (a training step changes every parameter): ONE launch re-packs everything.  No per-pack staleness test — with
        # ~800 packs that bookkeeping cost more host time than the launch does on the device.
        packs = [p for d in dicts if d for p in d.values()]
        if packs:
            self._pack_batch.run(packs, epoch=(self._ptr_epoch, self._pack_sets))
        self._packs_fp = fp

    @property
    def _bwd_split(self):
        """Buffer format of the backward pass.  'mixed' back-propagates (to the input only) the way it runs forward: fp16 planes, the
        gradient of the residual stream stored hi+lo, one-plane operands inside the dense blocks, with the incoming gradient scaled by a
        power of two into fp16's range (run_backward).  mixed_bwd = 'bf16' selects the earlier variant: bf16 hi+lo gradients (3 MFMAs
        per product), the saved fp16 activations serving as LeakyReLU' masks only."""
        if self.split == 'mixed':
            return True if self.mixed_bwd == 'bf16' else 'mixed'
        return self.split

    def _bwd_wfmt(self, rdb):
        """Weight format of a data-gradient pack (rdb: a dense-block pack)."""
        if self._bwd_split != 'mixed':
            return self._bwd_split
        return 'f16x2' if rdb else 'f16x3'

    def _wfmt(self, name):
        """Weight format of one layer's forward pack."""
        if self.split != 'mixed':
            return self.split
        return 'f16x2' if name.startswith('rrdb') else 'f16x3'

    @property
    def _pshuf(self):
        """Pixel-shuffle upsamplers (block.py:278-291): the shuffle factor, or 0 for the reference's default nearest + conv ('upconv')."""
        return 0 if self.net.upsample_mode == 'upconv' else (3 if self.net.upscale == 3 else 2)

    def _ps_rows(self, q=None):
        """Conv output channels of a pixel-shuffle upsampler in the kernel's row-group order (esr_conv3x3_desc.pixel_shuffle): row group
        g = (output group g // r^2, sub-position g % r^2); launch q covers row groups 8q .. 8q+7, q = None all of them."""
        r2 = self._pshuf ** 2
        groups = range(8 * r2) if q is None else range(8 * q, min(8 * q + 8, 8 * r2))
        return [((g // r2) * 8 + e) * r2 + g % r2 for g in groups for e in range(8)]

    def packed(self):
        if self._packed is None:
            d = {}
            for name, c, lat in self._convs():
                if name.startswith('up') and self._pshuf:
                    for q in range(self._pshuf ** 2):         # 64 * r^2 conv channels = r^2 launches of 64 rows
                        d[name, q] = A.PackedConv(c.weight, c.bias, lat, split=self._wfmt(name), rows=self._ps_rows(q))
                elif c.weight.shape[0] > 64:                  # nf = 128, 192, ...: output slices of one launch
                    d[name] = A.PackedConvSlices(c.weight, c.bias, lat, split=self._wfmt(name))
                else:
                    d[name] = A.PackedConv(c.weight, c.bias, lat, split=self._wfmt(name))
            self._packed = d
            self._pack_sets += 1
            self._pack_gen[0] = self._pack_sets
        self._refresh_packs()
        return self._packed

    def packed_t(self):
        """Data-gradient packs: weights transposed + flipped, one pack per 64-channel slice of the conv's input channels
        ('m0', 'm1', 'm2') and one for the latent group ('z')."""
        if self._packed_t is None:
            d = {}
            for name, c, lat in self._convs():
                if name.startswith('rrdb'):
                    continue                  # dense blocks: packed_rdb_t()
                main = c.weight.shape[1] - lat
                rows = self._ps_rows() if (name.startswith('up') and self._pshuf) else None      # K axis in esr_pixel_unshuffle's order
                for j in range((main + 63) // 64):
                    d[name, 'm%d' % j] = A.PackedConv(c.weight, None, lat, split=self._bwd_wfmt(False), transposed=True, m_slice=(64 * j, min(main, 64 * j + 64)), rows=rows)
                if lat:
                    d[name, 'z'] = A.PackedConv(c.weight, None, lat, split=self._bwd_wfmt(False), transposed=True, m_slice='latent')
            self._packed_t = d
            self._pack_sets += 1
            self._pack_gen[1] = self._pack_sets
        self._refresh_packs()
        return self._packed_t

    def packed_rdb_t(self):
        """Data-gradient packs of the dense blocks in "mirrored" form.  With the gradients of an RDB's five conv outputs stored as
        G' = [dy conv4 (8 groups) | dy conv3 (4) | dy conv2 | dy conv1 | dy conv0], the gradient of block c+1 (the output of conv c) is
        ONE conv over the first 8 + 4(3-c) groups of G' — sum_{i>c} W_i^T[rows of block c+1] * dy_i — masked by LeakyReLU', written
        right behind them; the gradient of the RDB input is one conv over all 24 groups.  Per RDB: 'g3'..'g0' (32 rows), 'gx0' [, 'gx1' ...] (64 rows each),
        'gz' (latent rows).  conv4's 0.2 (and the RRDB's 0.2 for the third RDB) is folded into its piece of every pack."""
        if self._packed_rdb_t is None:
            d = {}
            mods = {name: (c, lat) for name, c, lat in self._convs()}
            nf = self.net.nf
            for r in range(self.net.nb):
                for k in range(3):
                    name = 'rrdb%d.rdb%d' % (r, k)
                    ws = [mods['%s.conv%d' % (name, i)][0].weight for i in range(5)]
                    lat = mods[name + '.conv0'][1]
                    s4 = 0.2 * (0.2 if k == 2 else 1.0)
                    pieces = [(ws[4], s4), (ws[3], 1.0), (ws[2], 1.0), (ws[1], 1.0), (ws[0], 1.0)]
                    for c in (3, 2, 1, 0):
                        rows = list(range(lat + nf + 32 * c, lat + nf + 32 + 32 * c))
                        d[name, 'g%d' % c] = A.PackedSum(pieces[:4 - c], [rows] * (4 - c), split=self._bwd_wfmt(True))
                    for j in range((nf + 63) // 64):       # the gradient of the block input: 64 rows per launch
                        d[name, 'gx%d' % j] = A.PackedSum(pieces, [list(range(lat + 64 * j, lat + min(nf, 64 * j + 64)))] * 5, split=self._bwd_wfmt(True))
                    if lat:
                        d[name, 'gz'] = A.PackedSum(pieces, [list(range(lat))] * 5, split=self._bwd_wfmt(True))
            self._packed_rdb_t = d
            self._pack_sets += 1
            self._pack_gen[2] = self._pack_sets
        self._refresh_packs()
        return self._packed_rdb_t

    # ------------------------------------------------------------------ buffers
    def _buffers(self, B, h, w, dev, keep):
        """Activation buffers, cached per shape (their zero borders are written once).  A differentiable forward (keep=True) LEASES
        its set until the autograd node that saved it is gone: a second differentiable forward in the meantime (two generator
        calls before one backward) gets a fresh, uncached set instead of overwriting saved activations."""
        key = (B, h, w, str(dev), keep)
        cached = self._bufs.get(key)
        if cached is not None:
            busy = cached.get('_busy')
            if not keep or busy is None or busy() is None:
                return self._lease(cached) if keep else cached
            fresh = self._new_buffers(B, h, w, dev, keep)
            fresh['_ephemeral'] = True        # lives for one pass: not worth recording a launch list for
            return self._lease(fresh)
        if len(self._bufs) > 2:
            self._bufs.clear()
        d = self._new_buffers(B, h, w, dev, keep)
        self._bufs[key] = d
        return self._lease(d) if keep else d

    @staticmethod
    def _lease(d):
        lease = _Lease()
        d['_busy'] = weakref.ref(lease)
        out = dict(d)
        out['_lease'] = lease          # lives exactly as long as the caller's copy (the autograd context)
        return out

    def _new_buffers(self, B, h, w, dev, keep):
        net, sp = self.net, self.split
        sf, ng = net.upscale, self.ng
        d = {}
        has_lat = net.latent_input is not None and net.num_latent_channels > 0
        if has_lat:
            d['zlr'] = A.ActBuf(B, 1, h, w, dev, sp)
            if net._lat_all_layers:
                d['zhr'] = A.ActBuf(B, 1, sf * h, sf * w, dev, sp)
        d['xin'] = A.ActBuf(B, 1, h, w, dev, sp)
        d['fea'] = A.ActBuf(B, ng, h, w, dev, sp)
        # inference: three rotating RDB buffers; differentiable forward: one per RDB (saved activations)
        d['rdb'] = [A.ActBuf(B, self.nd, h, w, dev, sp) for _ in range(3 * net.nb if keep is True else 3)]
        if keep == 'masks':
            one_plane = 'f16' if sp in ('mixed', 'f16', 'f16x2', 'f16x3') else False
            d['stash'] = [A.ActBuf(B, 16, h, w, dev, one_plane) for _ in range(3 * net.nb)]
        d['last'] = A.ActBuf(B, ng, h, w, dev, sp)
        d['trunk'] = A.ActBuf(B, ng, h, w, dev, sp)
        ups = []
        s = 1
        for j in range(self.n_up):
            s *= 3 if sf == 3 else 2
            ups.append(A.ActBuf(B, ng, s * h, s * w, dev, sp))
        d['ups'] = ups
        d['hr0'] = A.ActBuf(B, ng, sf * h, sf * w, dev, sp)
        d['_plans'] = {}                      # recorded launch lists over this buffer set (shared by its leased copies)
        return d

    # ------------------------------------------------------------------ forward
    def forward(self, x, pad=0):
        A.require_gpu(x, 'generator input')
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from . import autograd as AG
            return AG.rrdb_forward_with_grad(self, x, pad)
        return self.run_forward(x, pad, keep=False)[0]

    def _check(self, x):
        net = self.net
        sf = net.upscale
        has_lat = net.latent_input is not None and net.num_latent_channels > 0
        lat1 = net.num_latent_channels if has_lat else 0
        if x.shape[1] != lat1 * sf * sf + 3:
            raise EsrError('expected %d input channels (latent %d x sf^2 + 3), got %d' % (lat1 * sf * sf + 3, lat1, x.shape[1]))
        return sf, has_lat, lat1

    def keep_mode(self):
        """What a differentiable forward saves: everything (True) when a parameter wants a gradient, else per self.stash."""
        if self.stash == 'masks' and not any(p.requires_grad for p in self.parameters()):
            return 'masks'
        return True

    def _plan_key(self, kind, *what):
        # a forward list points into the forward packs only: creating the data-gradient packs later (first backward) leaves it valid
        packs = self._pack_gen[0] if kind == 'fwd' else tuple(self._pack_gen)
        return (kind,) + what + (self.split, self._ptr_epoch, packs, A.LDS_STAGES)      # (the scheduling hint is part of every recorded descriptor)

    @A.one_stream
    def run_forward(self, x, pad=0, keep=False):
        """Returns (g, bufs).  keep=True keeps one buffer per RDB so that `bufs` holds every activation the backward needs."""
        net = self.net
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        sf, has_lat, lat1 = self._check(x)
        B, Ct, h0, w0 = x.shape
        self.packed()                         # weight packs refreshed here, outside any recording: replays assume current packs
        self.check_range(wait=False)          # an earlier forward's verdict, if it has arrived
        watch = self._watching(x.device)
        bufs = self._buffers(B, h0 + 2 * pad, w0 + 2 * pad, x.device, keep)
        g = torch.empty(B, net.out_nc, sf * (h0 + 2 * pad), sf * (w0 + 2 * pad), dtype=torch.float32, device=x.device)
        if not self.use_plans or bufs.get('_ephemeral') or A._rec() is not None:
            if watch is not None:
                watch.names = []
            with A.watching(watch):
                self._forward_launches(x, pad, keep, bufs, g)
            self._post_range()
            return g, bufs
        key = self._plan_key('fwd', tuple(x.shape), pad, keep)
        plan = bufs['_plans'].get(key)
        if plan is None:
            rec = A.Recorder({'x': x, 'g': g})
            if watch is not None:
                watch.names = []              # (every forward of this network issues the same launches in the same order: one name list serves all plans)
            with A.recording(rec), A.watching(watch):
                self._forward_launches(x, pad, keep, bufs, g)
            plan = bufs['_plans'][key] = rec.finish()
        plan.run({'x': x, 'g': g})
        self._post_range()
        return g, bufs

    def _forward_launches(self, x, pad, keep, bufs, g):
        """The launch sequence of one forward pass over `bufs` into `g` (issued directly, or collected by an active A.Recorder)."""
        net = self.net
        sf, has_lat, lat1 = self._check(x)
        B, Ct, h0, w0 = x.shape
        pk = self.packed()
        h, w = h0 + 2 * pad, w0 + 2 * pad
        H, W = sf * h, sf * w
        conv = A.conv3x3
        nf, ng, nd = net.nf, self.ng, self.nd
        A.reset_launch_parity()

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

        A.host_op(lambda ctx: self._ev and self._ev[0].record())
        rdb = bufs['rdb']
        nrdb = 3 * net.nb

        def buf_of(j):       # buffer that holds RDB j's dense block (j = 3*r + k); j == nrdb: the trunk's last feature map
            if j == nrdb:
                return bufs['last']
            return rdb[j] if keep is True else rdb[j % 3]

        # ---- fea_conv -> fea (shortcut source) and the first RDB buffer
        conv(pk['fea'], bufs['xin'].view(), B, h, w, nf, in0=zlr, out=bufs['fea'].view(), out2=buf_of(0).view(0, ng) if net.nb else None, name='fea_conv')
        # 'mixed': the dense blocks' intermediate activations (outputs of convs 0-3, read only inside their RDB) are ONE fp16 plane; only
        # the RDB input (groups 0:8, the residual stream) keeps hi+lo.  Their lo planes are never written (they stay zero).
        mixed = self.split == 'mixed'
        lo8 = dict(in1_lo_groups=ng) if mixed else {}
        xlo_mode = self.mixed_xlo
        lo_in = lo8 if xlo_mode == 'all' else dict(in1_lo_groups=-1)
        lo_c4 = lo8 if xlo_mode in ('all', 'conv4') else dict(in1_lo_groups=-1)
        if not mixed:
            lo_in = lo_c4 = {}
        for r in range(net.nb):
            rrdb_in = buf_of(3 * r)
            for k in range(3):
                buf, nxt = buf_of(3 * r + k), buf_of(3 * r + k + 1)
                for i in range(4):
                    o2 = dict(out2=bufs['stash'][3 * r + k].view(4 * i, 4)) if keep == 'masks' else {}      # the one-plane copy the backward's masks read
                    conv(pk['rrdb%d.rdb%d.conv%d' % (r, k, i)], buf.view(0, ng + 4 * i), B, h, w, 32, in0=zall, act_slope=0.2,
                         out=buf.view(ng + 4 * i, 4, with_lo=not mixed), name='rrdb%d.rdb%d.conv%d' % (r, k, i), **(lo_in if (i > 0 or xlo_mode != 'all') else {}), **o2)
                name = 'rrdb%d.rdb%d.conv4' % (r, k)
                if k < 2:     # RDB output: 0.2*conv5 + x            (block.py:235)
                    conv(pk[name], buf.view(0, nd), B, h, w, nf, in0=zall, alpha=0.2, res1=buf.view(0, ng), beta1=1.0, out=nxt.view(0, ng), name=name, **lo_c4)
                else:         # RRDB output: 0.2*(0.2*conv5 + x) + x_rrdb   (block.py:270); lands in the next RRDB's first buffer
                    # (inference: that is rrdb_in's own buffer when the three buffers rotate; the kernel's in-place residual is safe)
                    conv(pk[name], buf.view(0, nd), B, h, w, nf, in0=zall, alpha=0.04, res1=buf.view(0, ng), beta1=0.2,
                         res2=rrdb_in.view(0, ng), beta2=1.0, out=nxt.view(0, ng), name=name, **lo_c4)
        last = buf_of(nrdb).view(0, ng) if net.nb else bufs['fea'].view()
        # LR_conv + trunk shortcut (block.py:96)
        conv(pk['lr_conv'], last, B, h, w, nf, in0=zall, res1=bufs['fea'].view(), beta1=1.0, out=bufs['trunk'].view(), name='LR_conv')
        # upsamplers: nearest xs folded into the conv's input read
        src, s = bufs['trunk'], 1
        for j in range(self.n_up):
            f = 3 if sf == 3 else 2
            s *= f
            if self._pshuf:     # conv 64 -> 64*f^2 at the INPUT resolution, pixel shuffle folded into the store, LeakyReLU before it (it commutes)
                for q in range(f * f):
                    conv(pk['up%d' % j, q], src.view(), B, s // f * h, s // f * w, 64, act_slope=0.2, out=bufs['ups'][j].view(), pixel_shuffle=f, ps_rowgroup0=8 * q, name='upconv%d' % j)
            else:
                conv(pk['up%d' % j], src.view(), B, s * h, s * w, nf, upsample=f, act_slope=0.2, out=bufs['ups'][j].view(), name='upconv%d' % j)
            src = bufs['ups'][j]
        conv(pk['hr0'], src.view(), B, H, W, nf, in0=zhr, act_slope=0.2, out=bufs['hr0'].view(), name='HR_conv0')
        conv(pk['hr1'], bufs['hr0'].view(), B, H, W, net.out_nc, in0=zhr, out_nchw=g)
        A.host_op(lambda ctx: self._ev and self._ev[1].record())

    # ------------------------------------------------------------------ backward
    @A.one_stream
    def run_backward(self, x_shape, pad, bufs, dg, need_dx=True, need_dw=False):
        """Gradients of sum(g * dg): returns (dx or None, {param: grad} or None).  `bufs` = run_forward(..., keep=True)[1]."""
        if self.split == 'mixed' and need_dw and self._bwd_split != 'mixed':
            raise NotImplementedError("precision 'mixed' with mixed_bwd = 'bf16' back-propagates to the INPUT only (Z optimisation): its saved "
                                      "activations are fp16, the gradients bf16 — the weight-gradient kernel contracts one element format")
        if self.split in ('f16', 'f16x2'):
            raise NotImplementedError("the fp16 precisions are inference modes: fp16 gradients underflow without loss scaling; "
                                      "use 'split' (fp32-class) or 'bf16' for training / Z optimisation")
        if need_dw and 'stash' in bufs:
            raise EsrError('this forward kept LeakyReLU masks only (no parameter asked for a gradient when it ran): weight gradients need the activations')
        dg = dg.detach()
        dg = (dg if dg.dtype == torch.float32 else dg.float()).contiguous()
        self.packed_t()                       # weight packs refreshed outside any recording
        if self.net.nb:
            self.packed_rdb_t()
        # the launch-list path covers the passes whose only non-library work is allocating dx and the flat dW buffer; 'mixed' (device-side
        # gradient scales read by torch ops in between) and pixel-shuffle networks (index_copy_ of permuted rows) launch directly
        planned = self.use_plans and self._bwd_split != 'mixed' and not self._pshuf and not bufs.get('_ephemeral') and A._rec() is None \
            and '_plans' in bufs
        if not planned:
            return self._backward_launches(x_shape, pad, bufs, dg, need_dx, need_dw)
        B, Ct, h0, w0 = x_shape
        dx = torch.zeros(B, Ct, h0, w0, dtype=torch.float32, device=dg.device) if need_dx else None
        ext = {'dg': dg}
        if dx is not None:
            ext['dx'] = dx
        key = self._plan_key('bwd', tuple(x_shape), pad, bool(need_dx), bool(need_dw), None if self.wgrad_exchange is None else int(self.wgrad_exchange.bucket_bytes), str(self.wgrad_overlap or 0))
        entry = bufs['_plans'].get(key)
        if entry is None:
            wg = WGrad(self, need_dw, B)
            rec = A.Recorder(ext)
            with A.recording(rec):
                _, grads = self._backward_launches(x_shape, pad, bufs, dg, need_dx, need_dw, dx=dx, wg=wg, pool={}, keep=rec.keep)
            entry = bufs['_plans'][key] = (rec.finish(), wg)
        else:
            grads = entry[1].rebind()
        entry[0].run(ext)
        return dx, grads

    def _backward_launches(self, x_shape, pad, bufs, dg, need_dx, need_dw, dx=None, wg=None, pool=None, keep=None):
        """The launch sequence of one backward pass (issued directly, or collected by an active A.Recorder — then `dx`, the weight-gradient
        collector `wg`, a private gradient-buffer `pool` and the list `keep` that receives every buffer the recorded commands point to are
        supplied by the caller)."""
        net, sp = self.net, self._bwd_split
        sf = net.upscale
        has_lat = net.latent_input is not None and net.num_latent_channels > 0
        lat1 = net.num_latent_channels if has_lat else 0
        lat = net._lat_all_layers
        B, Ct, h0, w0 = x_shape
        h, w = h0 + 2 * pad, w0 + 2 * pad
        H, W = sf * h, sf * w
        dev = dg.device
        pt = self.packed_t()
        conv = A.conv3x3
        nf, ng, nd = net.nf, self.ng, self.nd
        A.reset_launch_parity()
        f16_bwd = sp == 'mixed'
        gscale = None                 # device scalar: the power of two the gradients in flight are currently scaled by
        if f16_bwd:
            # fp16 gradients need their magnitude managed, on the device (no host synchronisation), by powers of two (exact):
            #   * the incoming gradient is scaled so that its largest element lies in [8, 16): 12 binades of head room for the HR stages,
            #     whose gradients are hi+lo pairs (22 bits as long as they stay above fp16's subnormals);
            #   * the gradient of the trunk — which the HR stages of a freshly initialised generator attenuate ~1000x — and then the
            #     gradient of every RRDB's input are brought into [512, 1024): the dense blocks' one-plane intermediate gradients are
            #     ~60x smaller than the stream's and must stay in fp16's normal range, while a trained network may amplify the stream
            #     gradient by an order of magnitude per RRDB;
            #   * dx and every layer's dW are divided by the scale that was in force where they were produced.
            scaler = A.GradScaler(dev, _pow2_scale(dg, 4), max_rescales=net.nb + 2)
            gscale = scaler.current
            dg = dg * gscale
        hi_only = dict(in1_lo_groups=-1) if f16_bwd else {}       # dense-block convs multiply hi planes only (see run_forward)
        if wg is None:
            wg = WGrad(self, need_dw, B, hi_only=f16_bwd)
        wg.gscale = gscale
        # gradient buffers come from a per-engine pool and go back to it when this pass is over: their zero borders (which the conv
        # kernels rely on and no producer ever writes) survive, so a steady-state step does no buffer memsets at all.  Everything
        # runs on one stream, so the next pass may reuse them as soon as its kernels are enqueued behind this one's.
        pool_key = (B, h, w, str(dev), sp)
        if pool is None:
            if self._gpool_key != pool_key:
                self._gpool, self._gpool_key = {}, pool_key
            pool = self._gpool
        taken = []

        def galloc(Bb, ncg, Hh, Ww):
            free = pool.setdefault((ncg, Hh, Ww), [])
            buf = free.pop() if free else A.ActBuf(Bb, ncg, Hh, Ww, dev, sp)
            taken.append(buf)
            return buf

        def zview(bufname):
            return bufs[bufname].view() if bufname in bufs else None

        def dgrad(name, dy, out, cg_lo, cg_hi, Hh, Ww, alpha=1.0, accumulate=False, extra=None, extra_beta=1.0, mask=None, upsample=1):
            """out[cg_lo:cg_hi] (+)= alpha * conv_T(dy) [+ extra_beta*extra], then * act'(mask[cg_lo:cg_hi]) where mask is given as
            (buffer, first masked group, end masked group) in `out` group numbering."""
            for j in range(cg_lo // 8, (cg_hi + 7) // 8):
                lo, hi = max(cg_lo, 8 * j), min(cg_hi, 8 * j + 8)
                if lo >= hi:
                    continue
                kw = {}
                ov = out.view(lo, hi - lo)
                if accumulate:
                    kw.update(res1=ov, beta1=1.0)
                    if extra is not None and lo < 8:
                        kw.update(res2=extra, beta2=extra_beta)
                elif extra is not None and lo < 8:
                    kw.update(res1=extra, beta1=extra_beta)
                if mask is not None:
                    mb, m0, m1 = mask
                    a0, a1 = max(m0, lo), min(m1, hi)
                    if a0 < a1:
                        kw.update(mask_src=mb.view(a0, a1 - a0), mask_cg=(a0 - lo, a1 - lo), mask_slope=0.2)
                conv(pt[name, 'm%d' % j], dy, B, Hh, Ww, (hi - lo) * 8, alpha=alpha, out=ov, use_bias=False, **kw)

        def dgrad_z(name, dy, gz, Hh, Ww, alpha, first):
            kw = {} if first else dict(res1=gz.view(), beta1=1.0)
            conv(pt[name, 'z'], dy, B, Hh, Ww, lat1, alpha=alpha, out=gz.view(), use_bias=False, **kw)

        # ---- HR part
        G_g = galloc(B, 1, H, W)
        A.pack_nchw(dg, G_g.view(), 0, net.out_nc)
        G_hr0 = galloc(B, ng, H, W)
        # the latent's own gradient is only computed when the input asks for one (Z search); a training step — Z is a noise input — skips
        # those launches (one 192 -> lat data-gradient conv per RDB plus the HR ones: ~1.5 ms of the configs[2] step)
        zgrad = need_dx and lat
        GZ_hr = galloc(B, 1, H, W) if (has_lat and zgrad) else None
        GZ_lr = galloc(B, 1, h, w) if (has_lat and need_dx) else None
        wg.conv('hr1', G_g.view(), bufs['hr0'].view(), zview('zhr') if lat else None, H, W, keep=(G_g,))
        dgrad('hr1', G_g.view(), G_hr0, 0, ng, H, W, mask=(bufs['hr0'], 0, ng))
        if GZ_hr is not None:
            dgrad_z('hr1', G_g.view(), GZ_hr, H, W, 1.0, first=True)
        G_up = galloc(B, ng, H, W)
        src_act = bufs['ups'][-1] if self.n_up else bufs['trunk']
        wg.conv('hr0', G_hr0.view(), src_act.view(), zview('zhr') if lat else None, H, W, keep=(G_hr0,))
        dgrad('hr0', G_hr0.view(), G_up, 0, ng, H, W, mask=(src_act, 0, ng) if self.n_up else None)
        if GZ_hr is not None:
            dgrad_z('hr0', G_hr0.view(), GZ_hr, H, W, 1.0, first=False)
        del G_hr0, G_g
        # ---- upsamplers (reverse): conv data-gradient at the upsampled size, then sum-pool = adjoint of nearest upsample
        cur_g, s = G_up, sf if sf != 3 else 3
        for j in reversed(range(self.n_up)):
            f = 3 if sf == 3 else 2
            Hj, Wj = s * h, s * w
            below = bufs['ups'][j - 1] if j > 0 else bufs['trunk']
            if self._pshuf:
                # cur_g = d(shuffled, pre-activation conv output); back to the conv's own channel layout (row-group order), then the plain
                # weight / data gradients of a 64 -> 64*f^2 conv at the lower resolution
                s //= f
                Gc = galloc(B, 8 * f * f, s * h, s * w)
                A.pixel_unshuffle(cur_g.view(), f, Gc.view(), B)
                wg.conv('up%d' % j, Gc.view(), below.view(), None, s * h, s * w, keep=(Gc, cur_g), rows=self._ps_rows())
                nxt_g = galloc(B, 8, s * h, s * w)
                dgrad('up%d' % j, Gc.view(), nxt_g, 0, 8, s * h, s * w, mask=(below, 0, 8) if j > 0 else None)
                cur_g = nxt_g
                continue
            wg.conv('up%d' % j, cur_g.view(), below.view(), None, Hj, Wj, upsample=f, keep=(cur_g,))
            tmp = galloc(B, ng, Hj, Wj)
            dgrad('up%d' % j, cur_g.view(), tmp, 0, ng, Hj, Wj)
            s //= f
            nxt_g = galloc(B, ng, s * h, s * w)
            A.act_combine(nxt_g.view(), B, Bv=tmp.view(), beta=1.0, s=f, mask=below.view() if j > 0 else None)
            cur_g = nxt_g
            del tmp
        G_trunk = cur_g
        if f16_bwd:
            scaler.rescale(B, [G_trunk.view()], 10)
            gscale_hr, gscale = gscale, scaler.current   # the HR-resolution latent gradient (GZ_hr) stays at the first scale
            gscale_trunk = gscale                        # ... and G_trunk, needed again for the trunk's shortcut, at this one
            wg.gscale = gscale
        # ---- trunk: trunk = fea + LR_conv(last)
        last_act = bufs['last'] if net.nb else bufs['fea']
        pr = self.packed_rdb_t() if net.nb else {}
        # RDB gradient buffers G' (see packed_rdb_t): 4 rotating ones — unless weight gradients are wanted: then every RDB keeps its
        # own, because its dy slices feed the deferred batched launch
        ring = [galloc(B, nd, h, w) for _ in range(4)] if (net.nb and not need_dw) else []
        nseq = 3 * net.nb

        def gbuf(n):                          # n-th RDB in backward order (n = 0: last RDB of the last RRDB)
            return ring[n % 4] if ring else galloc(B, nd, h, w)

        G_cur = gbuf(0) if net.nb else None
        G_first = G_cur if net.nb else galloc(B, ng, h, w)      # receives d(output of the last RRDB) = d(LR_conv input)
        wg.conv('lr_conv', G_trunk.view(), last_act.view(), zview('zlr') if lat else None, h, w, keep=(G_trunk,))
        dgrad('lr_conv', G_trunk.view(), G_first, 0, ng, h, w)
        zfirst = True
        if zgrad:
            dgrad_z('lr_conv', G_trunk.view(), GZ_lr, h, w, 1.0, first=True)
            zfirst = False
        dout = G_first                        # holds d(input of RRDB 0) in groups 0:8 when the loop is done
        n = 0
        for r in reversed(range(net.nb)):
            G_rrdb = G_cur                    # its groups 0:8 = d(output of RRDB r), needed again for the RRDB's skip connection
            for k in reversed(range(3)):
                stash = bufs['stash'][3 * r + k] if 'stash' in bufs else None       # masks-only forward: no full buffer per RDB
                X = bufs['rdb'][3 * r + k] if stash is None else None
                G = G_cur
                name = 'rrdb%d.rdb%d' % (r, k)
                s_out = 0.2 if k == 2 else 1.0          # RDB3's output enters the RRDB sum scaled by 0.2
                if need_dw:
                    wg.conv(name + '.conv4', G.view(0, ng), X.view(0, nd), zview('zlr') if lat else None, h, w, alpha=0.2 * s_out, keep=(G, X))
                for c in (3, 2, 1, 0):
                    g0 = ng + 4 * (3 - c)                # dy of conv c goes right behind the gradients it is computed from
                    conv(pr[name, 'g%d' % c], G.view(0, g0), B, h, w, 32, out=G.view(g0, 4, with_lo=not f16_bwd), use_bias=False,
                         mask_src=X.view(ng + 4 * c, 4) if stash is None else stash.view(4 * c, 4), mask_cg=(0, 4), mask_slope=0.2, **hi_only)
                    if need_dw:
                        wg.conv('%s.conv%d' % (name, c), G.view(g0, 4), X.view(0, ng + 4 * c), zview('zlr') if lat else None, h, w, keep=(G, X))
                if zgrad:
                    kwz = {} if zfirst else dict(res1=GZ_lr.view(), beta1=1.0)
                    conv(pr[name, 'gz'], G.view(0, nd), B, h, w, lat1, out=GZ_lr.view(), use_bias=False, **kwz, **hi_only)
                    zfirst = False
                # d(RDB input) = s_out*dy_out + sum_i W_i^T dy_i  (+ d(out of the RRDB) for its first RDB: the RRDB skip connection),
                # written where the next RDB in backward order expects its dy_out
                n += 1
                G_next = gbuf(n) if n < nseq else galloc(B, ng, h, w)
                for j in range((nf + 63) // 64):
                    g_lo, g_n = 8 * j, min(ng - 8 * j, 8)
                    kw = dict(res2=G_rrdb.view(g_lo, g_n), beta2=1.0) if k == 0 else {}
                    conv(pr[name, 'gx%d' % j], G.view(0, nd), B, h, w, 8 * g_n, out=G_next.view(g_lo, g_n), use_bias=False, res1=G.view(g_lo, g_n), beta1=s_out, **kw, **hi_only)
                G_cur = G_next
            if f16_bwd and self.grad_renorm:
                # d(input of RRDB r) is complete and not yet recorded anywhere: renormalise it (and the latent gradient accumulated so far)
                scaler.rescale(B, [G_cur.view(0, ng)] + ([GZ_lr.view()] if zgrad and not zfirst else []), 10)
                gscale = scaler.current
                wg.gscale = gscale
            dout = G_cur
        # d fea = d trunk (shortcut) + d(first RRDB input)
        G_fea = galloc(B, ng, h, w)
        G_short = G_trunk
        if f16_bwd and net.nb:        # the shortcut's gradient at the current scale (a copy: the weight-gradient launch still reads G_trunk)
            G_short = galloc(B, ng, h, w)
            scaler.rescaled_copy(B, G_trunk.view(), G_short.view(), gscale_trunk)
        A.act_combine(G_fea.view(), B, A_=dout.view(0, ng), alpha=1.0, Bv=G_short.view(), beta=1.0, s=1)
        wg.conv('fea', G_fea.view(), bufs['xin'].view(), zview('zlr'), h, w, keep=(G_fea,))
        if need_dx:
            if dx is None:
                dx = torch.zeros(B, Ct, h0, w0, dtype=torch.float32, device=dev)
            G_x = galloc(B, 1, h, w)
            conv(pt['fea', 'm0'], G_fea.view(), B, h, w, 3, out=G_x.view(), use_bias=False)
            A.unpack_grad_nchw(G_x.view(), dx, Ct, h0, w0, c0=Ct - 3, nc=3, pad=pad)
            if has_lat:
                dgrad_z('fea', G_fea.view(), GZ_lr, h, w, 1.0, first=zfirst)
                # the latent's gradient goes back through the raw [lat][sf*h0][sf*w0] view of the first lat*sf^2 channels
                kw = dict(batch_stride=Ct * h0 * w0)
                A.unpack_grad_nchw(GZ_lr.view(), dx, lat1, sf * h0, sf * w0, c0=0, nc=lat1, pad=sf * pad, down=sf, **kw)
                if GZ_hr is not None and not f16_bwd:
                    A.unpack_grad_nchw(GZ_hr.view(), dx, lat1, sf * h0, sf * w0, c0=0, nc=lat1, pad=sf * pad, accumulate=True, **kw)
        if dx is not None and gscale is not None:
            dx.div_(gscale)
            if GZ_hr is not None:     # scaled differently from the trunk's gradients: unpacked on its own, added in fp32
                dz = torch.zeros(B, lat1 * sf * sf, h0, w0, dtype=torch.float32, device=dev)
                A.unpack_grad_nchw(GZ_hr.view(), dz, lat1, sf * h0, sf * w0, c0=0, nc=lat1, pad=sf * pad)
                dx[:, :lat1 * sf * sf].add_(dz.div_(gscale_hr))
        grads = wg.result()                   # the batched weight-gradient launch is enqueued here, before the buffers are recycled
        if keep is not None:                  # recorded: the launch list owns its gradient buffers
            keep.extend(taken)
        else:
            for buf in reversed(taken):       # reversed: the next pass pops them in the same order, i.e. builds the same descriptors
                pool[buf.ncg, buf.H, buf.W].append(buf)
        return dx, grads


def _pow2_scale(t, exp):
    """Device scalar 2^k such that max|t| * 2^k lies in [2^(exp-1), 2^exp)  (k = exp for an all-zero t)."""
    _, e = torch.frexp(torch.linalg.vector_norm(t, ord=float('inf')).float())          # max|t| = m * 2^e, m in [0.5, 1)
    return torch.ldexp(torch.ones((), dtype=torch.float32, device=t.device), exp - e)


# RRDBEngine.wgrad_overlap of new engines (see there); 0 = one weight-gradient launch behind the data-gradient chain
WGRAD_OVERLAP = 3

_SIDE_CAPPED = {}


def _side_kernel_capped(f16):
    """Does the weight-gradient instantiation of the second stream hold its CU alone (esr_conv3x3_wgrad_side_occupancy == 1)?  That is what lets the
    data-gradient chain's small launches always find room next to it; it rests on the register allocator honouring a clobber, i.e. on the
    toolchain that built the library — asked once per process, and a build where it does not hold keeps the one-stream backward (with a warning:
    the two-stream form would then SLOW the chain down)."""
    if f16 not in _SIDE_CAPPED:
        occ = _lib.lib.esr_conv3x3_wgrad_side_occupancy(1 if f16 else 0)
        _SIDE_CAPPED[f16] = occ == 1
        if occ != 1:
            import warnings
            warnings.warn('libesr_hip: the side-stream weight-gradient kernel reports %d resident workgroups per CU (expected 1): '
                          'wgrad_overlap is ignored, the backward runs on one stream' % occ)
    return _SIDE_CAPPED[f16]


class WGrad:
    """Weight / bias gradient collection.  Layers are only RECORDED while the data-gradient pass walks the network; result() runs
    them all in one batched launch (esr_conv3x3_wgrad_batch), which is why every gradient / activation buffer a record refers to
    is kept alive here until then."""

    def __init__(self, engine, enabled, B, hi_only=False):
        self.engine, self.enabled, self.B = engine, enabled, B
        # hi_only ('mixed'): gradients and activations are fp16 planes and the contraction uses their hi planes (one MFMA per product, fp32
        # accumulate).  gscale: device scalar (or None), the power of two the dy recorded from now on are scaled by; result() divides
        # each layer's gradients by the value that was current when the layer was recorded.
        self.hi_only, self.gscale = hi_only, None
        self.scaled = []              # (flat offset, length, gscale)
        self.grads = {} if enabled else None
        self.mods = {name: c for name, c, _ in engine._convs()} if enabled else None
        self.lats = {name: lat for name, _, lat in engine._convs()} if enabled else None
        self.descs, self.keep, self.permuted, self.desc_off, self.ready = [], [], [], [], []
        if enabled:
            # one zeroed flat buffer per backward pass, handed out as views (a fresh one every time: the views become .grad tensors)
            self.offsets, n = {}, 0
            for name, c in self.mods.items():
                self.offsets[name] = n
                n += c.weight.numel() + c.weight.shape[0]
            self.flat = torch.zeros(n, dtype=torch.float32, device=next(iter(self.mods.values())).weight.device)
            self._sizes = [k for c in self.mods.values() for k in (c.weight.numel(), c.weight.shape[0])]
            self._params = [(c.weight, c.bias) for c in self.mods.values()]

    def conv(self, name, dy, x_main, x_lat, H, W, alpha=1.0, upsample=1, keep=(), rows=None):
        """rows: dy's channels are a permutation of the layer's output channels (pixel-shuffle convs): dy channel i is output channel rows[i]."""
        if not self.enabled:
            return
        c = self.mods[name]
        if self.hi_only:
            dy, x_main, x_lat = A.hi_plane(dy), A.hi_plane(x_main), A.hi_plane(x_lat)
        o, nw = self.offsets[name], c.weight.numel()
        out = (self.flat[o:o + nw].view(c.weight.shape), self.flat[o + nw:o + nw + c.weight.shape[0]])
        if rows is not None:       # accumulate in dy's channel order, un-permute after the launch (result())
            tmp = (torch.zeros_like(out[0]), torch.zeros_like(out[1]))
            self.permuted.append((tmp, out, torch.tensor(rows, dtype=torch.long, device=out[0].device)))
            final, out = out, tmp
        if self.gscale is not None:
            self.scaled.append((o, nw + c.weight.shape[0], self.gscale))
        d, dw, db = A.wgrad_desc(dy, x_main, x_lat, self.lats[name], c.weight.shape, self.B, H, W, alpha, upsample, c.weight.device, out=out)
        self.descs.append(d)
        self.desc_off.append(o)
        rec = A._rec()
        self.ready.append(rec.position() if rec is not None else None)      # (the launch that wrote dy is already in the list)
        self.keep.extend(keep)
        if rows is not None:
            dw, db = final
        self.grads[c.weight] = dw
        if c.bias is not None:
            self.grads[c.bias] = db
        else:
            self.keep.append(db)

    def _groups(self):
        """[(flat start, flat end, [indices into self.descs])]: ONE group normally; with an exchange attached (engine.wgrad_exchange) one per
        gradient bucket — contiguous runs of layers, in the flat buffer's order, of about exchange.bucket_bytes each."""
        ex = self.engine.wgrad_exchange
        n = self.flat.numel() if self.flat is not None else self._n
        if ex is None or self.permuted or self.scaled:
            return [(0, n, list(range(len(self.descs))))]
        names = list(self.mods)
        bounds, start = [], 0
        for k, name in enumerate(names):                  # bucket boundaries at layer boundaries of the flat layout
            end = self.offsets[names[k + 1]] if k + 1 < len(names) else n
            if (end - start) * 4 >= ex.bucket_bytes or k + 1 == len(names):
                bounds.append((start, end))
                start = end
        groups = []
        for o0, o1 in bounds:
            idx = [i for i, o in enumerate(self.desc_off) if o0 <= o < o1]
            if idx:
                groups.append((o0, o1, idx))
        # every element of the flat buffer belongs to exactly one bucket: stretch the groups over the gaps of layers without a recorded gradient
        for k in range(len(groups)):
            lo = 0 if k == 0 else groups[k - 1][1]
            hi = n if k + 1 == len(groups) else groups[k][1]
            groups[k] = (lo, hi, groups[k][2])
        return groups or [(0, n, [])]

    def result(self):
        if self.enabled and self.descs:
            rec = A._rec()
            dev = self.flat.device
            groups = self._groups()
            # several launches slice every layer's pixel sum exactly as the one launch over all layers would (bit-identical gradients)
            unit = A.wgrad_batch_unit(self.descs) if len(groups) > 1 else 0
            # (pixel-shuffle layers' gradients are un-permuted after the launch, 'mixed' gradients are still multiplied by this RANK's power-of-two
            # scales until the loop at the end of this function: both are exchanged by the caller, after the backward — summing scaled buffers
            # across ranks and dividing by the local scale would give every rank different, wrong gradients)
            ex = None if (self.permuted or self.scaled) else self.engine.wgrad_exchange
            if rec is not None:
                # recorded pass: the descriptor tables go to the device now, their launches into the list; rebind() serves the replays
                self._flat_ptr = self.flat.data_ptr()
                self._cur_flat = self.flat if ex is not None else None      # (only an exchange slices it; without one nothing here outlives the step)
                self._tables = []                 # [(descriptor array, workspace, plan)] one per group
                if self._overlap_groups(rec, dev):
                    rec.keep.extend(self.keep)
                    self._n, self._dev = self.flat.numel(), dev
                    grads, self.descs, self.keep, self.flat, self.grads, self.ready = self.grads, [], [], None, None, []
                    return grads
                for gi, (o0, o1, idx) in enumerate(groups):
                    if idx:
                        arr = (_lib.WgradDesc * len(idx))(*[self.descs[i] for i in idx])
                        ws, plan = A.wgrad_batch_upload(arr, dev, unit)
                        self._tables.append((arr, ws, plan))
                        rec.emit(_lib.OP_WGRAD_BATCH_RUN, _lib.CmdWgradBatchRun(ws.data_ptr(), plan))
                        rec.keep.append(ws)
                    if ex is not None:            # replayed between two segments of the list: the bucket of the launch just enqueued
                        A.host_op(lambda ctx, wg=self, gi=gi, o0=o0, o1=o1, last=(gi + 1 == len(groups)): wg._exchange(gi, o0, o1, last))
                rec.keep.extend(self.keep)
                assert not self.permuted and not self.scaled
                self._n, self._dev = self.flat.numel(), dev
                grads, self.descs, self.keep, self.flat, self.grads = self.grads, [], [], None, None      # hold no reference to a step's gradients
                return grads
            self._cur_flat = self.flat if ex is not None else None
            for gi, (o0, o1, idx) in enumerate(groups):
                if idx:
                    A.conv3x3_wgrad_batch([self.descs[i] for i in idx], dev, cache=self.engine._wgb, unit=unit)
                if ex is not None:
                    self._exchange(gi, o0, o1, gi + 1 == len(groups))
            for (tdw, tdb), (dw, db), rows in self.permuted:
                dw.index_copy_(0, rows, tdw)
                db.index_copy_(0, rows, tdb)
            # undo the gradient scaling: consecutive layers recorded under the same scale are contiguous in `flat` more often than not
            runs = []
            for o, n, g in sorted(self.scaled, key=lambda t: t[0]):
                if runs and runs[-1][2] is g and runs[-1][0] + runs[-1][1] == o:
                    runs[-1][1] += n
                else:
                    runs.append([o, n, g])
            for o, n, g in runs:
                self.flat[o:o + n].div_(g)
            self.descs, self.keep = [], []
        return self.grads

    def _overlap_groups(self, rec, dev):
        """engine.wgrad_overlap: the recorded launches as groups in readiness order — all but the last on the engine's second stream, each
        enqueued (a host step of the list) right behind the main-stream launch that completed its last dy; the last group and the join at the
        end of the list.  With an exchange attached (engine.wgrad_exchange) the groups are its buckets: each group's stretch of the flat buffer
        is handed to exchange.start() behind an event recorded after its own launch — the collective waits for that launch only and runs under
        everything that follows — and exchange.finish() comes behind the join.  False: not applicable here (permuted / rescaled gradients, too few layers, hi+lo
        gradients)."""
        eng = self.engine
        fr = eng.wgrad_overlap if isinstance(eng.wgrad_overlap, (tuple, list)) else None      # (experiments: the groups' shares of the layers)
        G = len(fr) if fr else int(eng.wgrad_overlap or 0)
        if G < 2 or self.permuted or self.scaled or len(self.descs) < 4 * G or eng._bwd_split is True:
            return False
        if not _side_kernel_capped(eng._bwd_split in ('mixed', 'f16')):
            return False                               # (this build's side instantiation is not one workgroup per CU: the one-stream backward)
        if eng._side is None:
            eng._side = torch.cuda.Stream(device=dev)
        side, ex = eng._side, eng.wgrad_exchange
        unit = A.wgrad_batch_unit(self.descs)          # every group slices its layers' pixel sums as the one launch would
        n = len(self.descs)
        bounds = [n * g // G for g in range(G + 1)] if not fr else [0] + [min(n, int(round(n * sum(fr[:g + 1]) / sum(fr)))) for g in range(G)]
        # the groups' stretches of the flat buffer: the backward meets the layers in (roughly) the reverse of the buffer's order, so a group is
        # a contiguous run of it; the runs are stretched over layers without a recorded gradient so that every element belongs to one group
        names = list(self.mods)
        ends = {self.offsets[nm]: (self.offsets[names[k + 1]] if k + 1 < len(names) else self.flat.numel()) for k, nm in enumerate(names)}
        spans = [(min(self.desc_off[i] for i in range(bounds[g], bounds[g + 1])), max(ends[self.desc_off[i]] for i in range(bounds[g], bounds[g + 1]))) for g in range(G)]
        order = sorted(range(G), key=lambda g: spans[g][0])
        disjoint = all(spans[order[k]][1] <= spans[order[k + 1]][0] for k in range(G - 1))
        if ex is not None and not disjoint:
            return False                               # (an exotic layer order: the bucketed launches of result() serve the exchange)
        cover = {}
        for k, g in enumerate(order):
            cover[g] = (0 if k == 0 else spans[order[k - 1]][1], self.flat.numel() if k + 1 == G else spans[g][1])

        def exchange(g, last=False):
            if ex is not None and self._cur_flat is not None:
                ex.start(g, self._cur_flat[cover[g][0]:cover[g][1]])
                if last:
                    ex.finish()
                    self._cur_flat = None
        hooks, done = [], {}
        for g in range(G):
            idx = range(bounds[g], bounds[g + 1])
            arr = (_lib.WgradDesc * len(idx))(*[self.descs[i] for i in idx])
            ws, plan = A.wgrad_batch_upload(arr, dev, unit)
            self._tables.append((arr, ws, plan))
            rec.keep.append(ws)
            if g + 1 < G:
                def launch(ctx, ws=ws, plan=plan, g=g):
                    ev = torch.cuda.Event()
                    ev.record()
                    side.wait_event(ev)
                    _lib.check(_lib.lib.esr_conv3x3_wgrad_batch_run_side(ws.data_ptr(), C.byref(plan), side.cuda_stream), 'esr_conv3x3_wgrad_batch_run_side')
                    if ex is not None:
                        done[g] = torch.cuda.Event()
                        done[g].record(side)
                # behind the launch that wrote the group's last dy = where the NEXT layer was recorded
                hooks.append((self.ready[bounds[g + 1]], launch))
            else:
                rec.emit(_lib.OP_WGRAD_BATCH_RUN, _lib.CmdWgradBatchRun(ws.data_ptr(), plan))

        def join(ctx):
            # The collectives are issued HERE, when the host has enqueued the whole pass (it runs milliseconds ahead of the GPU), not between the
            # segments of the chain: a collective call may block the host until the stream it waits for has caught up (a one-rank RCCL group does:
            # 2.0-2.7 ms per call with the group's launch still running — profiles/r05_c3_exchange_ab.log), and the chain's 12 us launches must
            # not wait for the host.  Each one still waits only for its own group's launch (the event recorded behind it on the second stream).
            if ex is not None:
                if eng._xs is None:
                    eng._xs = torch.cuda.Stream(device=dev)
                for g in range(G - 1):
                    eng._xs.wait_event(done.pop(g))
                    with torch.cuda.stream(eng._xs):
                        exchange(g)
            torch.cuda.current_stream().wait_stream(side)
            exchange(G - 1, last=True)
        rec.host(join)
        for pos, fn in reversed(hooks):                # (positions grow with the layer order: last first)
            rec.insert_host(pos, fn)
        return True

    def _exchange(self, gi, o0, o1, last):
        ex = self.engine.wgrad_exchange
        if ex is None or self._cur_flat is None:
            return
        ex.start(gi, self._cur_flat[o0:o1])
        if last:
            ex.finish()
            self._cur_flat = None

    def rebind(self):
        """Replay of a recorded pass: a fresh zeroed flat buffer (its views become the parameters' .grad), the descriptor table re-pointed
        (and re-uploaded) only if the allocator did not hand back the same storage.  Returns {param: grad}."""
        if not self.enabled:
            return None
        flat = torch.zeros(self._n, dtype=torch.float32, device=self._dev)
        # what the recorded exchange hooks slice (WGrad._exchange) — and only then: a plan that kept the previous step's 67 MB buffer alive would
        # keep the allocator from handing the same block back, i.e. re-base every table every step
        self._cur_flat = flat if self.engine.wgrad_exchange is not None else None
        delta = flat.data_ptr() - self._flat_ptr
        if delta:             # the tables' dW / db pointers move with the buffer: patched on the device (no host copy, stream-ordered)
            for arr, ws, plan in self._tables:
                _lib.check(_lib.lib.esr_conv3x3_wgrad_batch_rebase(ws.data_ptr(), C.byref(plan), delta, A.stream_ptr()), 'esr_conv3x3_wgrad_batch_rebase')
            self._flat_ptr += delta
        parts = flat.split(self._sizes)
        grads = {}
        for i, (w, b) in enumerate(self._params):
            grads[w] = parts[2 * i].view(w.shape)
            if b is not None:
                grads[b] = parts[2 * i + 1]
        return grads
