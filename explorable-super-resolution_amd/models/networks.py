"""Network factory — define_G / init_weights with the reference's signatures (codes/models/networks.py:14-123).

Differences that follow from the MI355X design (one process per GPU, RCCL data parallelism):
  * define_G never wraps the generator in nn.DataParallel; callers that reach through `.module` (GUI.py:1687) still work
    because the CEM wrapper and RRDBNet expose `.module` as themselves.
  * discriminators / feature extractors (define_D / define_F) are not part of the RRDB+CEM path.
"""
import functools

import torch
import torch.nn as nn
from torch.nn import init

import models.modules.architecture as arch


def weights_init_normal(m, std=0.02):
    classname = m.__class__.__name__
    if classname.find('Conv') != -1 or classname.find('Linear') != -1:
        init.normal_(m.weight.data, 0.0, std)
        if m.bias is not None:
            m.bias.data.zero_()
    elif classname.find('BatchNorm2d') != -1:
        init.normal_(m.weight.data, 1.0, std)
        init.constant_(m.bias.data, 0.0)


def weights_init_kaiming(m, scale=1):
    if 'filter_layer' in m.__dict__ and m.__getattribute__('filter_layer'):
        return      # CEM taps are constants (reference networks.py:29-31)
    classname = m.__class__.__name__
    if classname.find('Conv') != -1 or classname.find('Linear') != -1:
        init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
        m.weight.data *= scale
        if m.bias is not None:
            m.bias.data.zero_()
    elif classname.find('BatchNorm2d') != -1:
        init.constant_(m.weight.data, 1.0)
        init.constant_(m.bias.data, 0.0)


def weights_init_orthogonal(m):
    classname = m.__class__.__name__
    if classname.find('Conv') != -1 or classname.find('Linear') != -1:
        init.orthogonal_(m.weight.data, gain=1)
        if m.bias is not None:
            m.bias.data.zero_()
    elif classname.find('BatchNorm2d') != -1:
        init.constant_(m.weight.data, 1.0)
        init.constant_(m.bias.data, 0.0)


def init_weights(net, init_type='kaiming', scale=1, std=0.02):
    print('initialization method [{:s}]'.format(init_type))
    if init_type == 'normal':
        net.apply(functools.partial(weights_init_normal, std=std))
    elif init_type == 'kaiming':
        net.apply(functools.partial(weights_init_kaiming, scale=scale))
    elif init_type == 'orthogonal':
        net.apply(weights_init_orthogonal)
    else:
        raise NotImplementedError('initialization method [{:s}] not implemented'.format(init_type))


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
    raise NotImplementedError('define_D: discriminators are outside the RRDB+CEM hot path (SURVEY.md §8(f) "next")')


def define_F(opt, use_bn=False, **kwargs):
    raise NotImplementedError('define_F: the VGG feature extractor needs torchvision and is outside the RRDB+CEM hot path')
