"""TEST INFRASTRUCTURE ONLY.  CPU restatement (plain torch ops + autograd) of the CALLERS of the hot path, pinned against fixture F7
(tests/golden/callers_f7.npz, produced by the reference's own classes):
  * the generator's loss graph of SRRaGANModel.optimize_parameters — codes/models/SRRaGAN_model.py:319-333 (HR_unpadder crop),
    :418-447 (pixel / range terms), :462-472 (GAN term)
  * Discriminator_VGG_128.forward as a function of its state_dict — codes/models/modules/architecture.py:446-508 — and the critic's
    WGAN-GP step — SRRaGAN_model.py:350-369, codes/models/modules/loss.py:212-246,260-279
  * Z_optimizer.optimize — codes/Z_optimization.py:273-319 (tanh-bounded Z), :647-797 (Adam loop, objectives STD / TV, min-loss keeper)
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import cem_oracle as co
from oracle import rrdb_oracle as ro


def generator_output(sd, lr, z, nb, lat, sf, taps, eval_mode):
    """CEM-wrapped generator on (LR, HR-resolution Z): Prepare_Input's raw view (SRRaGAN_model.py:230-236), eval-mode replicate padding
    and crop (CEMnet.py:286-295,311)."""
    x = lr
    if eval_mode:
        x = F.pad(x, (taps.margins_LR,) * 4, mode='replicate')
        z = F.pad(z, (taps.margins_HR,) * 4, mode='replicate') if z is not None else None
    xin = x if z is None else torch.cat([z.reshape(z.size(0), lat * sf * sf, x.shape[2], x.shape[3]), x], 1)
    gen = ro.rrdb_forward(sd, xin, nb, sf, lat, prefix='generated_image_model.model')
    return co.cem_combine(x, gen, taps, crop=eval_mode)


def range_loss(x, lo=0.0, hi=1.0):
    return torch.max(torch.max(x - hi, lo - x), torch.zeros_like(x)).mean()          # loss.py:248-258


def crop(t, m):
    return t[..., m:-m, m:-m]


# ------------------------------------------------------------------------------------------------ discriminator
def d_forward(sd, x, train=True):
    """Discriminator_VGG_128 from its state_dict: conv (3x3 s1 / 4x4 s2, zero pad (k-1)//2) [+ BatchNorm] + LeakyReLU(0.2) per feature
    block, then Linear -> LeakyReLU -> Linear.  train=True: BatchNorm uses batch statistics (the module's training mode)."""
    idx = sorted({int(k.split('.')[1]) for k in sd if k.startswith('features.')})
    i = 0
    while i < len(idx):
        n = idx[i]
        w = sd['features.%d.weight' % n]
        assert w.dim() == 4
        k = w.shape[-1]
        x = F.conv2d(x, w, sd['features.%d.bias' % n], stride=2 if k == 4 else 1, padding=(k - 1) // 2)
        i += 1
        if i < len(idx) and ('features.%d.running_mean' % idx[i]) in sd:
            m = idx[i]
            x = F.batch_norm(x, None if train else sd['features.%d.running_mean' % m], None if train else sd['features.%d.running_var' % m],
                             sd['features.%d.weight' % m], sd['features.%d.bias' % m], training=train, eps=1e-5)
            i += 1
        x = F.leaky_relu(x, 0.2)
    x = x.reshape(x.size(0), -1)
    x = F.leaky_relu(F.linear(x, sd['classifier.0.weight'], sd['classifier.0.bias']), 0.2)
    return F.linear(x, sd['classifier.2.weight'], sd['classifier.2.bias'])


def d_losses(sdD, real, fake, pt, gp_w):
    """Non-relativistic WGAN-GP critic losses (SRRaGAN_model.py:350-369): returns (l_d_real, l_d_fake, l_d_gp, total, pred_real, pred_fake)."""
    pred_real, pred_fake = d_forward(sdD, real), d_forward(sdD, fake)
    l_d_real, l_d_fake = 2 * (-pred_real.mean()), 2 * pred_fake.mean()
    interp = (pt * fake + (1 - pt) * real).detach().requires_grad_(True)
    crit = d_forward(sdD, interp)
    grad = torch.autograd.grad(crit, interp, torch.ones_like(crit), create_graph=True, retain_graph=True)[0]
    l_d_gp = gp_w * ((grad.reshape(grad.size(0), -1).norm(2, dim=1) - 1) ** 2).mean()
    return l_d_real, l_d_fake, l_d_gp, (l_d_real + l_d_fake) / 2 + l_d_gp, pred_real, pred_fake


# ------------------------------------------------------------------------------------------------ latent search
def tv_loss(im):
    return (im[:, :, :, :-1] - im[:, :, :, 1:]).abs().mean(dim=(1, 2, 3)) + (im[:, :, :-1, :] - im[:, :, 1:, :]).abs().mean(dim=(1, 2, 3))


def arctanh(t):
    eps = torch.finfo(t.dtype).eps
    return 0.5 * torch.log((1 + t + eps) / (1 - t + eps))


def z_search(sd, lr, z0, nb, lat, sf, taps, objective, iters, lr_adam, std_increment=None, z_range=1.0, image_mask=None, z_mask=None):
    """Returns (loss history, final Z).  lr: [B,3,h,w]; z0: [B,lat,sf*h,sf*w] in (-z_range, z_range).  image_mask [H,W]: the objective sees
    output * mask (Z_optimization.py:383-388,627,728); z_mask [H,W]: entries outside it are pinned to their initial value (:278-299)."""
    def output(z):
        img = torch.clamp(generator_output(sd, lr, z, nb, lat, sf, taps, eval_mode=True), 0, 1)
        return img if image_mask is None else img * image_mask

    def std_of(img):
        return torch.std(img, dim=(1, 2, 3)).view(1, -1)
    with torch.no_grad():
        # Masked_STD(first_image_only=True) honours its flag only for the 'local' objectives (Z_optimization.py:617-627): for the
        # whole-image objectives every sample is measured against its OWN initial STD
        initial_std = std_of(output(z0))
    desired = initial_std
    if 'increase' in objective or 'decrease' in objective:
        sign = 1 if 'increase' in objective else -1
        desired = initial_std * (1.05 ** sign) if std_increment is None else initial_std + sign * std_increment
    pre0 = arctanh(torch.clamp(z0 / z_range, -1 + torch.finfo(z0.dtype).eps, 1 - torch.finfo(z0.dtype).eps))
    pre = pre0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pre], lr=lr_adam)
    losses, history = [], []
    for _ in range(iters):
        opt.zero_grad()
        if z_mask is not None:
            pre.data = z_mask * pre.data + (1 - z_mask) * pre0
        history.append(pre.detach().clone())
        img = output(z_range * torch.tanh(pre))
        if 'TV' in objective:
            loss = (100 * (std_of(img) - initial_std) ** 2).mean(0) + tv_loss(img)      # STD_PRESERVING_WEIGHT = 100 (Z_optimization.py:508-509)
        else:
            loss = std_of(img)
            if 'increase' in objective or 'decrease' in objective:
                loss = (loss - desired) ** 2
            loss = loss.mean(0)
        if 'max' in objective:
            loss = -loss
        loss = loss.mean()
        loss.backward()
        losses.append(float(loss))
        opt.step()
    final = pre.detach() if z_mask is None else z_mask * pre.detach() + (1 - z_mask) * pre0
    if np.min(losses) != losses[-1]:                   # keep the iterate with the smallest loss (Z_optimization.py:755-762)
        k = int(np.argmin(losses))
        final, losses = history[k], losses[:k + 1]
    return losses, z_range * torch.tanh(final), initial_std


# ------------------------------------------------------------------------------------------------ soft histogram objective
def soft_hist(gray, bins, lo, hi, T, eps=1e-7):
    """Z_optimization.py:170-209 (ComputeSoftHistogram, non-KDE form): normalised soft histogram of the gray values [n] in float64."""
    c = torch.linspace(lo, hi, bins).double().view(1, -1)
    v = gray.double().view(-1, 1)
    d = (v - c).abs()
    d = torch.min(d, (v - c - hi).abs())
    d = torch.min(d, (v - c + hi).abs())
    h = torch.exp(-((d + eps) ** 2) / T).mean(0)
    return (h / h.sum()).float()


def soft_hist_loss(cur_images, desired_image, bins, lo, hi, T, mask=None):
    """SoftHistogramLoss.forward (:211-229) for gray scale / patch size 1: KLDivLoss(log p_cur stacked over the batch, p_desired)."""
    p_des = soft_hist(desired_image.mean(0).reshape(-1), bins, lo, hi, T).view(1, -1)
    logs = []
    for im in cur_images:
        g = im.mean(0).reshape(-1)
        if mask is not None:
            g = g[mask.reshape(-1).bool()]
        p = soft_hist(g, bins, lo, hi, T)
        logs.append(torch.log(p + torch.finfo(p.dtype).eps).view(1, -1))
    return F.kl_div(torch.cat(logs, 0), p_des, reduction='mean')
