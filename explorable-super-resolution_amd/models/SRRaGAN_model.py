"""SRRaGANModel — the generator-side counterpart of the reference's model wrapper (codes/models/SRRaGAN_model.py): the calls the
drivers (train.py / test.py / GUI.py) and Z_optimizer make around the RRDB+CEM hot path, with the same method names, attributes and
input packing.  What is reproduced:
  * Prepare_Input: [Z viewed as B x (lat*sf^2) x h x w | LR] (raw view, :230-236); GetLatent; feed_data's Z sampling (:244-278)
  * test(): eval mode (CEM pre-padding) with or without autograd (:523-531); Output_Batch
  * optimize_parameters(): the GENERATOR step — forward in train mode (no pre-pad), HR_unpadder crop of targets, pixel (L1/L2),
    range and latent-consistency-free losses, gradient accumulation, Adam — with the gradients all-reduced over RCCL when
    several ranks run (one process per GPU) instead of nn.DataParallel
  * save / load through BaseModel (positional checkpoint loading)
Out of scope for the hot path and therefore refused loudly: discriminator / GAN / VGG-feature losses (define_D / define_F),
the D-verification and LR-rollback heuristics, validation image dumps.
"""
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

import CEM.CEMnet as CEMnet
import models.networks as networks
from esr_hip import dist as esr_dist
from .base_model import BaseModel


def Latent_channels_desc_2_num_channels(latent_channels_desc):
    """Number of Z channels for a `latent_channels` option (reference models/modules/loss.py:16-25)."""
    if latent_channels_desc is None or latent_channels_desc == 0:
        return 0
    if isinstance(latent_channels_desc, int):
        return latent_channels_desc
    if 'structure_tensor' in latent_channels_desc:
        return 3
    if latent_channels_desc in ('STD_1dir', 'STD_directional'):
        return 2
    raise NotImplementedError('Unknown latent channel setting %s' % latent_channels_desc)


def CreateRangeLoss(legit_range):
    """Penalty on values outside the legit range (reference models/modules/loss.py:248-258)."""
    lo, hi = float(legit_range[0]), float(legit_range[1])

    def RangeLoss(x):
        return torch.max(torch.max(x - hi, lo - x), torch.zeros_like(x)).mean()
    return RangeLoss


class SRRaGANModel(BaseModel):
    def __init__(self, opt, accumulation_steps_per_batch=1, init_Fnet=None, init_Dnet=None, **kwargs):
        super(SRRaGANModel, self).__init__(opt)
        train_opt = opt['train']
        self.log_path = opt['path']['log']
        self.latent_input_domain = opt['network_G']['latent_input_domain']
        self.latent_input = opt['network_G']['latent_input'] if opt['network_G']['latent_input'] != 'None' else None
        if self.latent_input is not None:
            self.Z_size_factor = opt['scale'] if 'HR' in opt['network_G']['latent_input_domain'] else 1
        self.num_latent_channels = Latent_channels_desc_2_num_channels(opt['network_G']['latent_channels'])
        self.cri_latent = None
        self.step = 0
        self.CEM_net = None
        self.CEM_arch = opt['network_G']['CEM_arch']
        if self.CEM_arch or (opt['is_train'] and train_opt['CEM_exp']) or self.latent_input is not None:
            CEM_conf = CEMnet.Get_CEM_Conf(opt['scale'])
            CEM_conf.sigmoid_range_limit = bool(opt['network_G']['sigmoid_range_limit'])
            CEM_conf.input_range = np.array(opt['range'])
            if self.is_train:
                assert train_opt['pixel_domain'] == 'HR' or not self.CEM_arch, 'Why should I use CEM_arch AND penalize MSE in the LR domain?'
                CEM_conf.decomposed_output = bool(opt['network_D']['decomposed_input']) if opt['network_D'] else False
            if opt['test'] is not None and opt['test']['kernel'] == 'estimated':
                CEM_conf.lower_magnitude_bound = 0.1   # estimated kernels: keep the inversion of hTh stable
            kernel = kwargs['kernel'] if 'kernel' in kwargs else (None if opt['test'] is None else opt['test']['kernel'])
            self.CEM_net = CEMnet.CEMnet(CEM_conf, upscale_kernel=kernel)
            if not self.CEM_arch:
                self.CEM_net.WrapArchitecture_PyTorch(only_padders=True)
        self.netG = networks.define_G(opt, CEM=self.CEM_net, num_latent_channels=self.num_latent_channels)
        self.netG.to(self.device)
        esr_dist.broadcast_parameters(self.netG)          # every rank starts from rank 0's weights
        logs_2_keep = ['l_g_pix', 'l_g_range', 'psnr_val', 'LR_decrease']
        self._log_dict = OrderedDict(zip(logs_2_keep, [[] for _ in logs_2_keep]))
        self._pending_logs = []          # (key, step, [loss tensors of the accumulation steps]): read back lazily, see log_dict
        self.D_exists = False
        self.generator_changed = True
        if self.is_train:
            if train_opt['gan_weight'] is not None or train_opt['feature_weight'] is not None:
                raise NotImplementedError('GAN / VGG-feature losses need define_D / define_F, which are outside the RRDB+CEM hot path '
                                          '(SURVEY.md §8(f)); set train.gan_weight and train.feature_weight to null')
            self.grad_accumulation_steps_G = train_opt['grad_accumulation_steps_G'] or 1
            self.max_accumulation_steps = accumulation_steps_per_batch
            self.decomposed_output = False
            self.netG.train()
            self.cri_pix = None
            if train_opt['pixel_weight'] is not None:
                l_pix_type = train_opt['pixel_criterion']
                if l_pix_type == 'l1':
                    self.cri_pix = nn.L1Loss().to(self.device)
                elif l_pix_type == 'l2':
                    self.cri_pix = nn.MSELoss().to(self.device)
                else:
                    raise NotImplementedError('Loss type [{:s}] not recognized.'.format(l_pix_type))
                self.l_pix_w = train_opt['pixel_weight']
            self.cri_range = None
            if train_opt['range_weight'] is not None and train_opt['range_weight'] > 0:
                self.cri_range = CreateRangeLoss(opt['range'])
                self.l_range_w = train_opt['range_weight']
            wd_G = train_opt['weight_decay_G'] if train_opt['weight_decay_G'] else 0
            optim_params = [v for k, v in self.netG.named_parameters() if v.requires_grad]
            self.optimizer_G = torch.optim.Adam(optim_params, lr=train_opt['lr_G'], weight_decay=wd_G, betas=(train_opt['beta1_G'] or 0.9, 0.999))
            self.optimizers.append(self.optimizer_G)
            self.lr_G = train_opt['lr_G']
            if train_opt['lr_scheme'] == 'MultiStepLR':
                for optimizer in self.optimizers:
                    self.schedulers.append(torch.optim.lr_scheduler.MultiStepLR(optimizer, train_opt['lr_steps'], train_opt['lr_gamma']))
            self.grad_reducer = esr_dist.GradBucketAllReducer(optim_params)
            self.gradient_step_num = 0
        self.load()
        print('---------- Model initialized ------------------')

    # ------------------------------------------------------------------ input packing (reference :224-278)
    def Output_Batch(self, within_0_1):
        return torch.clamp(self.fake_H, 0, 1) if within_0_1 else self.fake_H

    def Prepare_Input(self, LR_image, latent_input, **kwargs):
        if latent_input is not None:
            if LR_image.size()[2:] != latent_input.size()[2:]:
                latent_input = latent_input.contiguous().view([latent_input.size(0)] + [latent_input.size(1) * self.opt['scale'] ** 2] + list(LR_image.size()[2:]))
            self.model_input = torch.cat([latent_input, LR_image], dim=1)
        else:
            self.model_input = 1 * LR_image

    def GetLatent(self):
        latent = 1 * self.model_input[:, :-3, ...]
        if latent.size(1) != self.num_latent_channels:
            latent = latent.view([latent.size(0)] + [self.num_latent_channels] + [self.opt['scale'] * val for val in list(latent.size()[2:])])
        return latent

    def feed_data(self, data, need_GT=True, **kwargs):
        self.var_L = data['LR'].to(self.device)
        if self.latent_input is not None:
            if 'Z' in data.keys():
                cur_Z = data['Z']
            else:
                cur_Z = 2 * torch.rand([self.var_L.size(0), self.num_latent_channels] + [self.Z_size_factor * v for v in list(self.var_L.size()[2:])]) - 1
            if isinstance(cur_Z, (int, float)) or (not torch.is_tensor(cur_Z) and np.ndim(cur_Z) < 4):
                cur_Z = cur_Z * np.ones([1, self.num_latent_channels] + [self.Z_size_factor * v for v in list(self.var_L.size()[2:])])
            elif torch.is_tensor(cur_Z) and cur_Z.dim() == 4 and cur_Z.size(2) == 1:
                cur_Z = cur_Z * torch.ones([1, 1] + [self.Z_size_factor * v for v in list(self.var_L.size()[2:])], device=cur_Z.device)
            if not torch.is_tensor(cur_Z):
                cur_Z = torch.from_numpy(np.asarray(cur_Z, dtype=np.float32))
            cur_Z = cur_Z.to(device=self.device, dtype=self.var_L.dtype)
            if cur_Z.size(0) == 1 and self.var_L.size(0) > 1:
                cur_Z = cur_Z.expand(self.var_L.size(0), -1, -1, -1)
        else:
            cur_Z = None
        self.Prepare_Input(LR_image=self.var_L, latent_input=cur_Z)
        if need_GT:
            self.var_H = data['HR'].to(self.device)
            input_ref = data['ref'] if 'ref' in data else data['HR']
            self.var_ref = input_ref.to(self.device)

    # ------------------------------------------------------------------ inference (reference :523-531)
    def test(self, prevent_grads_calc=True, **kwargs):
        self.netG.eval()
        if prevent_grads_calc:
            with torch.no_grad():
                self.fake_H = self.netG(self.model_input)
        else:
            self.fake_H = self.netG(self.model_input)
        self.output_image = 1 * self.fake_H
        self.netG.train()

    # ------------------------------------------------------------------ generator step (reference :280-333,418-499, G side)
    def optimize_parameters(self):
        self.gradient_step_num = self.step // max(self.max_accumulation_steps, 1)
        first_grad_accumulation_step_G = self.step % self.grad_accumulation_steps_G == 0
        last_grad_accumulation_step_G = self.step % self.grad_accumulation_steps_G == (self.grad_accumulation_steps_G - 1)
        if first_grad_accumulation_step_G:
            self.log_dict                                   # read back the previous step's loss values (already computed: no stall)
            self.optimizer_G.zero_grad()
            self.l_g_pix_grad_step, self.l_g_range_grad_step = [], []
        var_H = self.var_H
        if self.CEM_net is not None and self.CEM_arch and var_H.size(2) > 2 * int(self.CEM_net.invalidity_margins_HR):
            var_H = self.CEM_net.HR_unpadder(self.var_H)     # losses ignore the frame the CEM cannot constrain (reference :319-320)
        self.fake_H = self.netG(self.model_input)            # train mode: no pre-padding
        fake_H = self.fake_H
        if self.CEM_net is not None and self.CEM_arch and fake_H.size(2) > 2 * int(self.CEM_net.invalidity_margins_HR):
            fake_H = self.CEM_net.HR_unpadder(fake_H)
        l_g_total = 0
        if self.cri_pix is not None:
            l_g_pix = self.cri_pix(fake_H, var_H)
            l_g_total = l_g_total + self.l_pix_w * l_g_pix / self.grad_accumulation_steps_G
        if self.cri_range is not None:
            l_g_range = self.cri_range(fake_H)
            l_g_total = l_g_total + self.l_range_w * l_g_range / self.grad_accumulation_steps_G
        l_g_total.backward()
        # The reference reads the loss values back right here (.item(), :483-493): a host synchronisation per step, during which the GPU
        # drains and then idles while the host prepares the next step.  Here they stay on the device and are read when somebody looks at
        # the log (log_dict / get_current_log) or when the next step starts — by then the kernels that produced them finished long ago.
        if self.cri_pix is not None:
            self.l_g_pix_grad_step.append(l_g_pix.detach())
        if self.cri_range is not None:
            self.l_g_range_grad_step.append(l_g_range.detach())
        if last_grad_accumulation_step_G:
            self.grad_reducer()                               # RCCL all-reduce (mean) of the G gradients: the one exchange step
            self.optimizer_G.step()
            self.generator_changed = True
            if self.cri_pix is not None:
                self._pending_logs.append(('l_g_pix', self.gradient_step_num, self.l_g_pix_grad_step))
            if self.cri_range is not None:
                self._pending_logs.append(('l_g_range', self.gradient_step_num, self.l_g_range_grad_step))
        self.step += 1

    # ------------------------------------------------------------------ bookkeeping
    @property
    def log_dict(self):
        """{name: [(gradient step, value), ...]} as in the reference; pending device-side loss values are read back first."""
        pending, self._pending_logs = self._pending_logs, []
        for key, step, vals in pending:
            self._log_dict[key].append((step, float(np.mean([float(v) for v in vals]))))
        return self._log_dict

    def get_current_log(self):
        return OrderedDict((k, v[-1][1]) for k, v in self.log_dict.items() if len(v) > 0)

    def get_current_visuals(self, need_HR=True, entire_batch=False, to_cpu=True):
        out_dict = OrderedDict()
        sel = (lambda t: t) if entire_batch else (lambda t: t[0])
        mov = (lambda t: t.detach().float().cpu()) if to_cpu else (lambda t: t.detach().float())
        out_dict['LR'] = mov(sel(self.var_L))
        out_dict['SR'] = mov(sel(self.fake_H))
        if need_HR:
            out_dict['HR'] = mov(sel(self.var_H))
        return out_dict

    def print_network(self):
        s, n = self.get_network_description(self.netG)
        print('Number of parameters in G: {:,d}'.format(n))

    def load(self, max_step=None, resume_train=None):
        load_path_G = self.opt['path']['pretrain_model_G'] if self.opt['path'] else None
        if resume_train is None:
            resume_train = self.is_train and self.opt['train']['resume']
        if resume_train and os.path.isdir(self.save_dir):
            steps = [int(f.split('_')[0]) for f in os.listdir(self.save_dir) if f.endswith('_G.pth') and f.split('_')[0].isdigit()]
            if max_step is not None:
                steps = [s for s in steps if s <= max_step]
            if steps:
                load_path_G = os.path.join(self.save_dir, '%d_G.pth' % max(steps))
                self.step = max(steps) * max(getattr(self, 'max_accumulation_steps', 1), 1)
        if load_path_G is not None:
            print('loading model for G [{:s}] ...'.format(load_path_G))
            self.load_network(load_path_G, self.netG, optimizer=self.optimizer_G if (self.is_train and resume_train) else None)

    def save(self, iter_label):
        if esr_dist.rank() != 0:
            return None
        return self.save_network(self.save_dir, self.netG, 'G', iter_label, self.optimizer_G)
