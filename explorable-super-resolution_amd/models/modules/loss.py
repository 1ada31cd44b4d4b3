"""Losses around the generator (reference: codes/models/modules/loss.py): GANLoss (:212-246), GradientPenaltyLoss (:260-279),
CreateRangeLoss (:248-258), Latent_channels_desc_2_num_channels (:16-25), FilterLoss (:27-209, the structure-tensor codes).
Element-wise / reduction torch ops on the generator's and the discriminator's outputs; device-agnostic (the reference hard-codes
torch.cuda.FloatTensor)."""
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

    def forward(self, interp, interp_crit, critic_group=None):
        """critic_group: `interp_crit` is the critic_group-th output of a grouped critic call (esr_hip.critic.critic_forward_group): the
        penalty's first backward then runs on that batch's images only."""
        from esr_hip.critic import input_grad_only
        with input_grad_only(group=critic_group, of=interp_crit):       # (hints to the HIP critic's nodes of THIS call; stock modules ignore them)
            grad_interp = torch.autograd.grad(outputs=interp_crit, inputs=interp, grad_outputs=torch.ones_like(interp_crit),
                                              create_graph=True, retain_graph=True, only_inputs=True)[0]
        norms = grad_interp.reshape(grad_interp.size(0), -1).norm(2, dim=1)
        return ((norms - 1) ** 2).mean()


class FilterLoss(nn.Module):
    """L_struct of the explorable-SR training (reference loss.py:27-209), model-training form for the structure-tensor latent codes
    ('structure_tensor*' and 'SVDinNormedOut_structure_tensor*'): the image's mean structure tensor, from forward differences
        Ix = x[.., 1:] - x[.., :-1] (first H-1 rows),  Iy = x[1:, ..] - x[:-1, ..] (first W-1 columns)        (2x2 filters, :49-62)
        S = [mean Ix^2, mean Iy^2, mean Ix*Iy]  over channels and pixels, per image                            (:141-151)
    measured on the SR output and turned into ratios against the ground truth's tensor (:160-175):
        'SVDinNormedOut_*': S_SR / (sqrt(S_HR[0] * S_HR[1]) + 1/255);   'structure_tensor*': S_SR[i] / (S_HR[i] + sign*1/255) for i < 2, S_SR[2] as is
    The target for each ratio is the latent code mapped affinely onto the [5th, 95th] percentile range of the ratios seen so far
    (a 10000-deep history per channel, updated with the current batch first — :178-186): the loss is |measured - target|, shape [B, 3].
    The percentile bounds are host-side statistics (not differentiated), exactly as in the reference.  Other latent codes
    ('STD_1dir', 'STD_directional', 'SVD_structure_tensor') and the GUI's constant-Z form are not part of this build."""
    NOISE_STD = 1 / 255
    LOWER_PERCENTILE, HIGHER_PERCENTILE = 5, 95

    def __init__(self, latent_channels, constant_Z=None, reference_images=None, masks=None, task='SR', gray_scale=False):
        super(FilterLoss, self).__init__()
        from collections import deque
        self.latent_channels = latent_channels
        self.num_channels = Latent_channels_desc_2_num_channels(latent_channels)
        self.model_training = isinstance(latent_channels, str)
        self.supported = self.model_training and 'structure_tensor' in latent_channels and not latent_channels.startswith('SVD_') and \
            constant_Z is None and task == 'SR'
        if self.supported:
            self.collected_ratios = [deque(maxlen=10000) for _ in range(self.num_channels)]

    @staticmethod
    def structure_tensor(x):
        if x.is_cuda:                       # one reduction kernel per call, closed-form gradient (esr_img_stats kind 2)
            from esr_hip import zobj
            return zobj.structure_tensor(x)
        ix = (x[..., :, 1:] - x[..., :, :-1])[..., :-1, :]
        iy = (x[..., 1:, :] - x[..., :-1, :])[..., :, :-1]
        return torch.stack([(ix * ix).mean(dim=(1, 2, 3)), (iy * iy).mean(dim=(1, 2, 3)), (ix * iy).mean(dim=(1, 2, 3))], 0)      # [3, B]

    def forward(self, data):
        if not self.supported:
            raise NotImplementedError("FilterLoss(latent_channels=%r): only the model-training structure-tensor codes are part of this build" % (self.latent_channels,))
        import numpy as np
        cur_Z = data['Z'].mean(dim=(2, 3))
        s_sr, s_hr = self.structure_tensor(data['SR']), self.structure_tensor(data['HR'])
        if self.latent_channels.startswith('SVDinNormedOut'):
            norm = torch.sqrt(s_hr[0]) * torch.sqrt(s_hr[1])
            measured = [s_sr[i] / (norm + self.NOISE_STD) for i in range(3)]
        else:
            measured = [s_sr[i] / (s_hr[i] + torch.sign(s_sr[i]) * self.NOISE_STD) if i < 2 else s_sr[i] for i in range(3)]
        targets = []
        for i in range(3):
            self.collected_ratios[i] += [float(v) for v in measured[i].detach()]
            hi = np.percentile(self.collected_ratios[i], self.HIGHER_PERCENTILE)
            lo = np.percentile(self.collected_ratios[i], self.LOWER_PERCENTILE)
            targets.append(cur_Z[:, i] / 2 * (hi - lo) + np.mean([hi, lo]))
        return (torch.stack(measured, 1) - torch.stack(targets, 1)).abs()
