"""RRDBNet — the ESRGAN-style generator of the reference (codes/models/modules/architecture.py:228-302) with the
same constructor, attributes, module tree and state_dict keys, executed by the gfx950 engine (esr_hip/engine.py).

Discriminator_VGG_128 (architecture.py:446-508), the other half of the configs[2] training step, is a stock-PyTorch module (MIOpen
convolutions, BatchNorm, Linear): the WGAN-GP penalty differentiates it twice, which autograd does for stock ops.  The other
discriminators / feature extractors of the reference's architecture.py are outside this build.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import block as B
from esr_hip.engine import RRDBEngine


class RRDBNet(nn.Module):
    def __init__(self, in_nc, out_nc, nf, nb, gc=32, upscale=4, norm_type=None, act_type='leakyrelu', mode='CNA',
                 upsample_mode='upconv', latent_input=None, num_latent_channels=None):
        super(RRDBNet, self).__init__()
        self.latent_input = None
        num_latent_channels = 0 if num_latent_channels is None else num_latent_channels   # (the reference raises TypeError on None)
        num_latent_channels_HR = 0
        if num_latent_channels > 0:
            self.latent_input = latent_input
            num_latent_channels_HR = 1 * num_latent_channels
            if 'HR_rearranged' in latent_input:
                num_latent_channels *= upscale ** 2
        self.num_latent_channels = 1 * num_latent_channels
        self.upscale = upscale
        n_upscale = int(math.log(upscale, 2))
        if upscale == 3:
            n_upscale = 1
        if latent_input is not None:
            in_nc += num_latent_channels
        if latent_input is None or 'all_layers' not in latent_input:
            num_latent_channels, num_latent_channels_HR = 0, 0
        if norm_type is not None:
            raise NotImplementedError('RRDBNet with normalisation layers is not on the RRDB+CEM path (reference default: norm_type=null)')
        if act_type != 'leakyrelu' or mode != 'CNA':
            # the fused kernels implement conv -> LeakyReLU(0.2) (what define_G hard-wires, networks.py:97-99) and nothing else
            raise NotImplementedError("RRDBNet(act_type=%r, mode=%r): the HIP engine implements act_type='leakyrelu', mode='CNA' only" % (act_type, mode))
        if nf < 16 or nf % 16 or (nf > 64 and nf % 64) or out_nc < 1 or in_nc < 1 or (upsample_mode == 'pixelshuffle' and nf != 64):
            # The residual stream is nf / 8 channel groups of the activation layout, a dense block ONE (nf / 8 + 16)-group buffer; nf = 16, 32, 48, 64
            # and 128, 192, ... run the same launch plan (round 6; reference architecture.py:228-230 takes any nf, every options file it ships uses
            # 64).  Multiples of 16: a K chunk of the MFMA contraction is two channel groups, and the mirrored data gradient concatenates the layers'
            # gradients chunk by chunk.  Above 64: multiples of 64 — a launch produces its output in 64-channel slices (esr_conv3x3_desc.cout > 64).
            # The pixel-shuffle store orders its rows in blocks of 64 channels: nf = 64 only.
            raise NotImplementedError("RRDBNet(nf=%r, upsample_mode=%r): the HIP engine runs nf = 16, 32, 48, 64, 128, 192, ... (the reference's options files use 64; "
                                      "gc is 32 — the reference ignores the argument too); pixel-shuffle upsamplers nf = 64 only" % (nf, upsample_mode))

        fea_conv = B.conv_block(in_nc, nf, kernel_size=3, norm_type=None, act_type=None, return_module_list=True)
        # NB: like the reference (architecture.py:250) the `gc` argument is ignored: growth channels are 32.
        rb_blocks = [B.RRDB(nf, kernel_size=3, gc=32, stride=1, bias=True, pad_type='zero', norm_type=norm_type, act_type=act_type,
                            mode='CNA', latent_input_channels=num_latent_channels) for _ in range(nb)]
        LR_conv = B.conv_block(nf + num_latent_channels, nf, kernel_size=3, norm_type=norm_type, act_type=None, mode=mode,
                               return_module_list=True)
        if upsample_mode == 'upconv':
            upsample_block = B.upconv_blcok
        elif upsample_mode == 'pixelshuffle':
            upsample_block = B.pixelshuffle_block
        else:
            raise NotImplementedError('upsample mode [{:s}] is not found'.format(upsample_mode))
        if upscale == 3:
            # the reference builds a bare Sequential here and then fails to concatenate it (architecture.py:260-261,271);
            # wrapping it in a list is the evident intent
            upsampler = [upsample_block(nf, nf, 3, act_type=act_type)]
        else:
            upsampler = [upsample_block(nf, nf, act_type=act_type) for _ in range(n_upscale)]
        HR_conv0 = B.conv_block(nf + num_latent_channels_HR, nf, kernel_size=3, norm_type=None, act_type=act_type, return_module_list=True)
        HR_conv1 = B.conv_block(nf + num_latent_channels_HR, out_nc, kernel_size=3, norm_type=None, act_type=None, return_module_list=True)
        self.model = nn.ModuleList(fea_conv + [B.ShortcutBlock(B.sequential(*(rb_blocks + LR_conv), return_module_list=True),
                                                               latent_input_channels=num_latent_channels, use_module_list=True)]
                                   + upsampler + HR_conv0 + HR_conv1)
        self.upsample_mode = upsample_mode
        self.nb, self.nf, self.out_nc = nb, nf, out_nc
        self._lat_all_layers = num_latent_channels
        self._engine = None

    # ---- engine plumbing
    @property
    def engine(self):
        if self._engine is None:
            self._engine = RRDBEngine(self)
        return self._engine

    def set_precision(self, precision):
        """Operand scheme of the MFMA kernels (fp32 accumulate and fp32 I/O in all of them):
        'split'  default: bf16 hi+lo weights x bf16 hi+lo activations, 3 MFMAs per product, fp32-class forward AND gradients;
        'mixed'  fp16: residual stream stored hi+lo, hi+lo operands in the six convs outside the dense blocks, one-plane operands inside
                 them; forward within 3e-5..3e-4 of fp32 at 0.55x the time of split; back-propagates with power-of-two gradient scaling
                 (input gradient ~1e-5, weight gradients ~3e-3 of fp32): inference, Z search, optionally training;
        'f16x2'  fp16 weights x fp16 hi+lo activations, 2 MFMAs, ~5e-4 on RRDB-23, inference only;
        'f16' / 'bf16'  one plane, one MFMA; 'f16' (1.6e-3) is inference only, 'bf16' (1.3e-2) also trains."""
        assert precision in ('split', 'mixed', 'f16x2', 'f16', 'bf16')
        self.engine.set_precision(precision)

    def check_range(self, wait=True):
        """fp16 precisions ('mixed', 'f16x2', 'f16'): raise EsrError if a forward since the last check stored activations that reached fp16's last
        binade or are not finite, naming the first such layer (esr_hip/engine.py: RRDBEngine.check_range; the kernels report it without an
        extra pass).  A forward looks at what has arrived by itself, without synchronising; call this where a result is consumed."""
        if self._engine is not None:
            self._engine.check_range(wait=wait)

    def invalidate_packs(self):
        """Call after editing parameters through `.data` (such writes bypass torch's version counters, which is what the engine watches)."""
        if self._engine is not None:
            self._engine.invalidate()

    def forward(self, x, pad=0):
        """x: [B, num_latent_channels*upscale^2 + 3, h, w] (Z packed by the raw view of SRRaGAN_model.py:233, LR image last).
        `pad` > 0 evaluates the generator on the replicate-padded input (CEM eval mode, CEMnet.py:286-295) without
        materialising the padded tensors."""
        if self.latent_input is not None and 'HR_downscaled' not in self.latent_input:
            if 'HR_rearranged' in self.latent_input and 'all_layers' in self.latent_input:
                raise Exception('Unsupported yet')          # same behaviour as architecture.py:295
            raise NotImplementedError("latent_input '%s': only '<all_layers|first_layer>_HR_downscaled' is implemented" % self.latent_input)
        return self.engine.forward(x, pad=pad)


class Discriminator_VGG_128(nn.Module):
    """VGG-style critic (reference architecture.py:446-508): 3x3 stride-1 convs alternating with 4x4 stride-2 convs, widths
    nf, nf, 2nf, 2nf, 4nf, 4nf, 8nf x4, BatchNorm + LeakyReLU(0.2) after every conv but the first (no norm there); the first `nb` of the
    ten conv blocks are kept; `num_2_strides` of the five 4x4 convs stride (the rest run at stride 1).  With all five strides the
    classifier is Linear(8nf * s^2, 100) -> LeakyReLU -> Linear(100, 1) on the s x s feature map (s = 4 for 128x128 inputs); with fewer
    it is a "patch" critic: conv 8x8 (no padding) -> LeakyReLU -> conv 1x1.  `features` is ONE flat nn.Sequential and `classifier` a
    3-element one, so state_dict keys and their order are the reference's (features.0.weight, features.2.weight, features.3.weight ...)."""

    def __init__(self, in_nc, base_nf, norm_type='batch', act_type='leakyrelu', mode='CNA', input_patch_size=128, num_2_strides=5, nb=10):
        super(Discriminator_VGG_128, self).__init__()
        assert num_2_strides <= 5, 'Can be modified by adding more stridable layers, if needed.'
        if mode != 'CNA' or act_type != 'leakyrelu' or norm_type not in ('batch', 'instance', None):
            raise NotImplementedError('Discriminator_VGG_128(norm_type=%r, act_type=%r, mode=%r)' % (norm_type, act_type, mode))
        self.num_2_strides = 1 * num_2_strides
        widths = [base_nf, base_nf, 2 * base_nf, 2 * base_nf, 4 * base_nf, 4 * base_nf, 8 * base_nf, 8 * base_nf, 8 * base_nf, 8 * base_nf]
        layers, cin, size, strides_left = [], in_nc, float(input_patch_size), num_2_strides
        for i, cout in enumerate(widths):
            k, stride = 3, 1
            if i % 2 == 1:                     # the down-sampling slots: 4x4, stride 2 while strides remain
                k, stride = 4, (2 if strides_left > 0 else 1)
                size = np.ceil((size - 1) / stride)
                strides_left -= 1
            block = [nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=(k - 1) // 2)]
            if i > 0 and norm_type is not None:
                block.append(B.norm(norm_type, cout))
            block.append(nn.LeakyReLU(0.2, True))
            layers.append(block)
            cin = cout
        layers = layers[:nb]
        self.features = nn.Sequential(*[m for block in layers for m in block])
        self.last_FC_layers = self.num_2_strides == 5
        if self.last_FC_layers:
            self.classifier = nn.Sequential(nn.Linear(base_nf * 8 * int(size) ** 2, 100), nn.LeakyReLU(0.2, True), nn.Linear(100, 1))
        else:
            feat = widths[len(layers) - 1]
            hidden = min(100, feat)
            c0 = [nn.Conv2d(feat, hidden, kernel_size=8, stride=1, padding=0)] + ([B.norm(norm_type, hidden)] if norm_type else []) + [nn.LeakyReLU(0.2, True)]
            c1 = [nn.Conv2d(hidden, 1, kernel_size=1, stride=1, padding=0)] + ([B.norm(norm_type, 1)] if norm_type else []) + [nn.LeakyReLU(0.2, True)]
            self.classifier = nn.Sequential(nn.Sequential(*c0), nn.LeakyReLU(0.2, False), nn.Sequential(*c1))

    def forward(self, x):
        x = self.features(x)
        if self.last_FC_layers:
            x = x.reshape(x.size(0), -1)          # (NCHW order also when the features ran channels_last)
        return self.classifier(x)
