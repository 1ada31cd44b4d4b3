"""Losses around the generator (reference: codes/models/modules/loss.py): GANLoss (:212-246), GradientPenaltyLoss (:260-279),
CreateRangeLoss (:248-258), Latent_channels_desc_2_num_channels (:16-25).  Element-wise / reduction torch ops on the generator's and the
discriminator's outputs; device-agnostic (the reference hard-codes torch.cuda.FloatTensor).  FilterLoss — the structure-tensor
latent-control loss (:27-209) — needs OpenCV's Sobel taps for one of its variants and is not part of this build (it raises)."""
import re

import torch
import torch.nn as nn


def Latent_channels_desc_2_num_channels(latent_channels_desc):
    """Number of Z channels of a `network_G.latent_channels` option: an int is taken as is, 'STD_1dir' -> 2,
    'STD_directional' / '*structure_tensor*' -> the number in the name, else 3."""
    if latent_channels_desc is None:
        return 0
    if isinstance(latent_channels_desc, int):
        return latent_channels_desc
    if latent_channels_desc == 'STD_1dir':
        return 2
    if latent_channels_desc == 'STD_directional' or 'structure_tensor' in latent_channels_desc:
        m = re.search(r'(\d)+', latent_channels_desc)
        return int(m.group(0)) if m is not None else 3
    raise NotImplementedError('Unknown latent channel setting %s' % latent_channels_desc)


class GANLoss(nn.Module):
    """'vanilla' (BCE with logits against constant labels), 'lsgan' (MSE against constant labels) or 'wgan*' (-mean for real, +mean for
    fake).  hinge_threshold clamps the logits from above (real) / below (fake) first."""

    def __init__(self, gan_type, real_label_val=1.0, fake_label_val=0.0):
        super(GANLoss, self).__init__()
        self.gan_type = gan_type.lower()
        self.real_label_val, self.fake_label_val = real_label_val, fake_label_val
        if self.gan_type == 'vanilla':
            self.loss = nn.BCEWithLogitsLoss()
        elif self.gan_type == 'lsgan':
            self.loss = nn.MSELoss()
        elif 'wgan' in self.gan_type:
            self.loss = lambda logits, is_real: -logits.mean() if is_real else logits.mean()
        else:
            raise NotImplementedError('GAN type [{:s}] is not found'.format(self.gan_type))

    def get_target_label(self, input, target_is_real):
        if 'wgan' in self.gan_type:
            return target_is_real
        return torch.full_like(input, self.real_label_val if target_is_real else self.fake_label_val)

    def forward(self, input, target_is_real, hinge_threshold=None):
        if hinge_threshold is not None:
            input = torch.clamp_max(input, hinge_threshold) if target_is_real else torch.clamp_min(input, -hinge_threshold)
        return self.loss(input, self.get_target_label(input, target_is_real))


def CreateRangeLoss(legit_range, chroma_mode=False):
    """Mean distance of the values from the legit range (0 inside it)."""
    lo, hi = float(legit_range[0]), float(legit_range[1])

    def RangeLoss(x):
        if chroma_mode:
            x = x[:, 1:, ...]
        return torch.max(torch.max(x - hi, lo - x), torch.zeros_like(x)).mean()
    return RangeLoss


class GradientPenaltyLoss(nn.Module):
    """WGAN-GP: mean over the batch of (||d critic / d interp||_2 - 1)^2, differentiable w.r.t. the critic's parameters (double
    backward through the discriminator only: `interp` is a leaf)."""

    def __init__(self, device=torch.device('cpu')):
        super(GradientPenaltyLoss, self).__init__()
        self.device = device

    def forward(self, interp, interp_crit):
        grad_interp = torch.autograd.grad(outputs=interp_crit, inputs=interp, grad_outputs=torch.ones_like(interp_crit),
                                          create_graph=True, retain_graph=True, only_inputs=True)[0]
        norms = grad_interp.reshape(grad_interp.size(0), -1).norm(2, dim=1)
        return ((norms - 1) ** 2).mean()


class FilterLoss(nn.Module):
    def __init__(self, latent_channels, **kwargs):
        super(FilterLoss, self).__init__()
        self.latent_channels = latent_channels
        self.num_channels = Latent_channels_desc_2_num_channels(latent_channels)

    def forward(self, data):
        raise NotImplementedError('FilterLoss (structure-tensor latent-control loss, reference loss.py:27-209) is not part of this build: '
                                  'set train.latent_weight to null')
