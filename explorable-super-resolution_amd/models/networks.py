"""Network factory — define_G / init_weights with the reference's signatures (codes/models/networks.py:14-123).

Differences that follow from the MI355X design (one process per GPU, RCCL data parallelism):
  * define_G never wraps the generator in nn.DataParallel; callers that reach through `.module` (GUI.py:1687) still work
    because the CEM wrapper and RRDBNet expose `.module` as themselves.
  * define_D builds the reference's default critic (Discriminator_VGG_128) from stock PyTorch modules (MIOpen kernels): it is the other
    half of the configs[2] training step, not part of the RRDB+CEM kernel path.  The VGG feature extractor (define_F) needs
    torchvision and stays out.
"""
import functools

import torch
import torch.nn as nn

import models.modules.architecture as arch


_WEIGHTED = (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)


def _is_cem_filter(m):
    """CEM taps are constants and keep their values (reference networks.py:29-31 checks the same marker attribute)."""
    return bool(getattr(m, 'filter_layer', False))


def _init_module(m, kind, scale=1.0, std=0.02):
    """One module's initialisation; semantics of the reference's weights_init_{normal,kaiming,orthogonal} (networks.py:14-60):
    conv / linear weights drawn per `kind` (kaiming: fan_in normal, then * scale), biases zero, BatchNorm affine = identity
    (weight ~ N(1, std) for 'normal')."""
    if isinstance(m, _WEIGHTED):
        if kind == 'kaiming' and _is_cem_filter(m):
            return
        with torch.no_grad():
            if kind == 'normal':
                m.weight.normal_(0.0, std)
            elif kind == 'kaiming':
                nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in')
                m.weight.mul_(scale)
            else:
                nn.init.orthogonal_(m.weight, gain=1)
            if m.bias is not None:
                m.bias.zero_()
    elif isinstance(m, nn.BatchNorm2d):
        with torch.no_grad():
            if kind == 'normal':
                m.weight.normal_(1.0, std)
            else:
                m.weight.fill_(1.0)
            m.bias.zero_()


def weights_init_normal(m, std=0.02):
    _init_module(m, 'normal', std=std)


def weights_init_kaiming(m, scale=1):
    _init_module(m, 'kaiming', scale=scale)


def weights_init_orthogonal(m):
    _init_module(m, 'orthogonal')


def init_weights(net, init_type='kaiming', scale=1, std=0.02):
    if init_type not in ('normal', 'kaiming', 'orthogonal'):
        raise NotImplementedError('initialization method [{:s}] not implemented'.format(init_type))
    print('initialization method [{:s}]'.format(init_type))
    net.apply(functools.partial(_init_module, kind=init_type, scale=scale, std=std))


def define_G(opt, CEM=None, num_latent_channels=None, **kwargs):
    gpu_ids = opt['gpu_ids']
    opt_net = opt['network_G']
    which_model = opt_net['which_model_G']
    opt_net['latent_input'] = opt_net['latent_input'] if opt_net['latent_input'] != "None" else None
    if which_model == 'RRDB_net':
        netG = arch.RRDBNet(in_nc=opt_net['in_nc'], out_nc=opt_net['out_nc'], nf=opt_net['nf'], nb=opt_net['nb'], gc=opt_net['gc'],
                            upscale=opt_net['scale'], norm_type=opt_net['norm_type'], act_type='leakyrelu', mode=opt_net['mode'],
                            upsample_mode='upconv',
                            latent_input=(opt_net['latent_input'] + '_' + opt_net['latent_input_domain']) if opt_net['latent_input'] is not None else None,
                            num_latent_channels=num_latent_channels)
    elif which_model in ('sr_resnet', 'DnCNN', 'MSRResNet'):
        raise NotImplementedError('Generator model [{:s}] is outside the RRDB+CEM hot path of this build'.format(which_model))
    else:
        raise NotImplementedError('Generator model [{:s}] not recognized'.format(which_model))
    if opt_net['CEM_arch']:
        netG = CEM.WrapArchitecture_PyTorch(netG, opt['datasets']['train']['patch_size'] if opt['is_train'] else None)
    if opt['is_train']:
        init_weights(netG, init_type='kaiming', scale=0.1)
    if gpu_ids:
        assert torch.cuda.is_available()
        # no nn.DataParallel: one process per GPU; gradients are all-reduced over RCCL by esr_hip.dist
    return netG


def define_D(opt, CEM=None, **kwargs):
    """Critic factory (reference networks.py:126-182).  The critic sees the generator's output with the CEM's invalidity frame cropped
    (patch_size - 2*margins_HR, :133-135)."""
    opt_net = opt['network_D']
    which_model = opt_net['which_model_D']
    input_patch_size = opt['datasets']['train']['patch_size']
    assert not ((opt_net['pre_clipping'] or opt_net['decomposed_input']) and which_model != 'PatchGAN'), 'Unsupported yet'
    if CEM is not None:
        input_patch_size -= 2 * CEM.invalidity_margins_HR
    if which_model == 'discriminator_vgg_128':
        kw = {'num_2_strides': opt_net['num_2_strides']} if opt_net['num_2_strides'] is not None else {}
        netD = arch.Discriminator_VGG_128(in_nc=opt_net['in_nc'], base_nf=opt_net['nf'], nb=opt_net['n_layers'] or 10, norm_type=opt_net['norm_type'],
                                          mode=opt_net['mode'], act_type=opt_net['act_type'], input_patch_size=int(input_patch_size), **kw)
    elif which_model in ('discriminator_vgg_128_nonModified', 'dis_acd', 'PatchGAN', 'discriminator_vgg_96', 'discriminator_vgg_192',
                         'discriminator_vgg_128_SN') or 'DnCNN_D' in which_model:
        raise NotImplementedError('Discriminator model [{:s}] is outside this build (the explorable-SR configuration uses discriminator_vgg_128)'.format(which_model))
    else:
        raise NotImplementedError('Discriminator model [{:s}] not recognized'.format(which_model))
    init_weights(netD, init_type='kaiming', scale=1)
    return netD          # no nn.DataParallel: one process per GPU, D gradients all-reduced over RCCL (esr_hip.dist)


def define_F(opt, use_bn=False, **kwargs):
    raise NotImplementedError('define_F: the VGG feature extractor needs torchvision and is outside the RRDB+CEM hot path')
