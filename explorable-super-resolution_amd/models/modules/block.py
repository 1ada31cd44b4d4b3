"""Building blocks of the RRDB generator — same names, signatures, module tree and state_dict keys as the reference's
codes/models/modules/block.py, re-implemented over the gfx950 kernels of libesr_hip.so.

The block classes here are parameter containers with the reference's structure: what executes them is
`architecture.RRDBNet.forward`, which plans the whole generator with zero-copy dense-block buffers and fused epilogues
(esr_hip/engine.py).  A block called on its own raises `EsrError` — there is no second, torch-op implementation of the
path to fall into; a single layer can be run through `HipConv2d` (one HIP conv launch, layout conversion at both ends).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from esr_hip import act as _act
from esr_hip._lib import EsrError


def _engine_only(self, *args, **kwargs):
    raise EsrError('%s is executed by RRDBNet.forward (esr_hip.engine.RRDBEngine) on the HIP kernels; it has no stand-alone forward'
                   % type(self).__name__)


class HipConv2d(nn.Conv2d):
    """nn.Conv2d(k=3, s=1, p=1) whose forward is the MFMA implicit-GEMM kernel (reference: block.py:141-142).
    Keeps nn.Conv2d's parameters/keys so checkpoints and `init_weights` (networks.py:29-42) work unchanged."""
    precision = 'split'      # 'split' (fp32-class, bf16x3) or 'bf16'

    def forward(self, x, act_slope=1.0):
        if self.kernel_size != (3, 3) or self.stride != (1, 1) or self.padding != (1, 1) or self.groups != 1 or self.dilation != (1, 1):
            raise NotImplementedError('HipConv2d implements the RRDB path only: 3x3, stride 1, zero pad 1')
        _act.require_gpu(x, 'conv input')
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            from esr_hip import autograd as _ag
            return _ag.conv3x3_function(x, self.weight, self.bias, act_slope, self.precision == 'split')
        return _act.conv3x3_nchw(x, self.weight, self.bias, act_slope, self.precision == 'split')


def act(act_type, inplace=True, neg_slope=0.2, n_prelu=1):
    """Activation module by name (reference block.py:10-23).  The generator's kernels fuse LeakyReLU(0.2); 'relu' is kept for callers
    that build plain blocks, 'prelu' (SRResNet only in the reference) is not part of this build."""
    act_type = act_type.lower()
    if act_type == 'relu':
        return nn.ReLU(inplace)
    if act_type == 'leakyrelu':
        return nn.LeakyReLU(neg_slope, inplace)
    raise NotImplementedError('activation layer [{:s}] is not found'.format(act_type))


def norm(norm_type, nc):
    """Normalisation module by name (reference block.py:26-36); used by the discriminator only — the generator has none."""
    norm_type = norm_type.lower()
    if norm_type == 'batch':
        return nn.BatchNorm2d(nc, affine=True)
    if norm_type == 'instance':
        return nn.InstanceNorm2d(nc, affine=False)
    raise NotImplementedError('normalization layer [{:s}] is not found'.format(norm_type))


def get_valid_padding(kernel_size, dilation):
    kernel_size = kernel_size + (kernel_size - 1) * (dilation - 1)
    return (kernel_size - 1) // 2


def sequential(*args, return_module_list=False):
    """Flatten Sequential (reference block.py:106-126)."""
    if len(args) == 1:
        if isinstance(args[0], OrderedDict):
            raise NotImplementedError('sequential does not support OrderedDict input.')
        return args[0]
    modules = []
    for module in args:
        if isinstance(module, nn.Sequential):
            modules.extend(module.children())
        elif isinstance(module, nn.Module):
            modules.append(module)
    return modules if return_module_list else nn.Sequential(*modules)


def conv_block(in_nc, out_nc, kernel_size, stride=1, dilation=1, groups=1, bias=True, pad_type='zero', norm_type=None,
               act_type='relu', mode='CNA', return_module_list=False):
    """Conv (+norm) (+act), reference block.py:129-155, in the one arrangement the RRDB generator uses: zero padding, 'CNA' order.  The
    reference's other arrangements (reflect / replicate padding modules, 'NAC' / 'CNAC' orders) serve SRResNet-style nets outside this build."""
    assert mode in ['CNA', 'NAC', 'CNAC'], 'Wong conv mode [{:s}]'.format(mode)
    if mode != 'CNA' or pad_type != 'zero':
        raise NotImplementedError("conv_block(mode=%r, pad_type=%r): the RRDB path uses mode='CNA' with zero padding" % (mode, pad_type))
    c = HipConv2d(in_nc, out_nc, kernel_size=kernel_size, stride=stride, padding=get_valid_padding(kernel_size, dilation), dilation=dilation, bias=bias,
                  groups=groups)
    n = norm(norm_type, out_nc) if norm_type else None
    a = act(act_type) if act_type else None
    return sequential(c, n, a, return_module_list=return_module_list)


class ShortcutBlock(nn.Module):
    """Elementwise sum of a sub-module chain's output and its input (reference block.py:76-103), with the latent
    channels re-prepended before every sub-module."""

    def __init__(self, submodule, latent_input_channels=0, use_module_list=False):
        super(ShortcutBlock, self).__init__()
        if use_module_list:
            submodule = nn.ModuleList(submodule)
        self.sub = submodule
        self.num_latent_channels = latent_input_channels

    forward = _engine_only

    def __repr__(self):
        return 'Identity + \n|' + self.sub.__repr__().replace('\n', '\n|')


class ResidualDenseBlock_5C(nn.Module):
    """Residual dense block, 5 convs (reference block.py:196-242)."""

    def __init__(self, nc, kernel_size=3, gc=32, stride=1, bias=True, pad_type='zero', norm_type=None, act_type='leakyrelu',
                 mode='CNA', latent_input_channels=0):
        super(ResidualDenseBlock_5C, self).__init__()
        self.USE_MODULE_LIST = True
        last_act = None if mode == 'CNA' else act_type
        self.convs = nn.ModuleList([
            conv_block(nc + i * gc + latent_input_channels, gc if i < 4 else nc, kernel_size if i < 4 else 3, stride, bias=bias,
                       pad_type=pad_type, norm_type=norm_type, act_type=act_type if i < 4 else last_act, mode=mode)
            for i in range(5)])

    forward = _engine_only


class RRDB(nn.Module):
    """Residual in residual dense block (reference block.py:245-270)."""

    def __init__(self, nc, kernel_size=3, gc=32, stride=1, bias=True, pad_type='zero', norm_type=None, act_type='leakyrelu',
                 mode='CNA', latent_input_channels=0):
        super(RRDB, self).__init__()
        self.num_latent_channels = latent_input_channels
        self.RDB1 = ResidualDenseBlock_5C(nc, kernel_size, gc, stride, bias, pad_type, norm_type, act_type, mode, latent_input_channels)
        self.RDB2 = ResidualDenseBlock_5C(nc, kernel_size, gc, stride, bias, pad_type, norm_type, act_type, mode, latent_input_channels)
        self.RDB3 = ResidualDenseBlock_5C(nc, kernel_size, gc, stride, bias, pad_type, norm_type, act_type, mode, latent_input_channels)

    forward = _engine_only


def pixelshuffle_block(in_nc, out_nc, upscale_factor=2, kernel_size=3, stride=1, bias=True, pad_type='zero', norm_type=None,
                       act_type='relu'):
    """Pixel-shuffle upsampler (reference block.py:278-291); reachable through RRDBNet(upsample_mode='pixelshuffle')."""
    conv = conv_block(in_nc, out_nc * (upscale_factor ** 2), kernel_size, stride, bias=bias, pad_type=pad_type, norm_type=None, act_type=None)
    n = norm(norm_type, out_nc) if norm_type else None
    a = act(act_type) if act_type else None
    return sequential(conv, nn.PixelShuffle(upscale_factor), n, a)


class Upsampler(nn.Module):
    def __init__(self, upscale_factor, mode):
        super(Upsampler, self).__init__()
        self.upscale_factor = upscale_factor
        self.mode = mode

    forward = _engine_only          # (nearest x2 / x3 is fused into the following conv's input fetch: esr_conv3x3_desc.upsample)


def upconv_blcok(in_nc, out_nc, upscale_factor=2, kernel_size=3, stride=1, bias=True, pad_type='zero', norm_type=None,
                 act_type='relu', mode='nearest'):
    """nearest upsample + conv (+act), reference block.py:302-309 (the reference's spelling of the name is kept)."""
    upsample = Upsampler(upscale_factor, mode)
    conv = conv_block(in_nc, out_nc, kernel_size, stride, bias=bias, pad_type=pad_type, norm_type=norm_type, act_type=act_type)
    return sequential(upsample, conv)
