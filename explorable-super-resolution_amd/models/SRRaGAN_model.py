"""SRRaGANModel — counterpart of the reference's live model wrapper (codes/models/SRRaGAN_model.py): the calls the drivers
(train.py / test.py / GUI.py) and Z_optimizer make around the RRDB+CEM hot path, with the same method names, attributes and input
packing.  What is reproduced:
  * Prepare_Input: [Z viewed as B x (lat*sf^2) x h x w | LR] (raw view, :230-236); GetLatent; feed_data's Z sampling (:244-278)
  * test(): eval mode (CEM pre-padding) with or without autograd (:523-531); Output_Batch
  * optimize_parameters() (:280-521): gradient accumulation, the D / G step gating (D_init_iters, D_update_ratio, D_verification
    'current' / 'past' / 'convergence'), the discriminator step (relativistic or plain; vanilla / lsgan / wgan-gp with the gradient penalty's double
    backward through D), the generator step (pixel, range, GAN and optimal-Z reference losses; the optimal-Z dual step runs
    Z_optimizer through the frozen generator), Adam for both — with G's and D's gradients all-reduced over RCCL when several ranks
    run (one process per GPU) instead of nn.DataParallel
  * perform_validation / save_log / save / load (G, D, optimizers, logs.npz) through BaseModel (positional checkpoint loading)
  * FilterLoss for train.latent_weight (structure-tensor statistics on the library's kernels, esr_hip/zobj.py)
Not reproduced, and refused loudly: the VGG-feature loss (define_F needs torchvision), the LR-rollback heuristics (:592-632), D_update_ratio
given as a controller range.
"""
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

import CEM.CEMnet as CEMnet
import models.networks as networks
from esr_hip import dist as esr_dist
from esr_hip import optim as esr_optim
from esr_hip.critic import CriticEngine, critic_forward, critic_forward_group
from esr_hip._lib import EsrError
from models.modules.loss import CreateRangeLoss, FilterLoss, GANLoss, GradientPenaltyLoss, Latent_channels_desc_2_num_channels
from .base_model import BaseModel


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
        logs_2_keep = ['l_g_pix', 'l_g_fea', 'l_g_range', 'l_g_gan', 'l_d_real', 'l_d_fake', 'D_loss_STD', 'l_d_real_fake', 'D_real', 'D_fake',
                       'D_logits_diff', 'psnr_val', 'D_update_ratio', 'LR_decrease', 'Correctly_distinguished', 'l_d_gp', 'per_pix_STD_val',
                       'l_g_optimalZ', 'Z_effect'] + ['l_g_latent_%d' % i for i in range(self.num_latent_channels)]
        self._log_dict = OrderedDict(zip(logs_2_keep, [[] for _ in logs_2_keep]))
        self._pending_logs = []          # (key, step, [device scalars of the accumulation steps]): read back lazily, see log_dict
        self.D_exists = False
        self.generator_changed = True
        self.generator_started_learning = False
        self.optimalZ_loss_type = None
        self.D_engine, self.D_engine_mode, self.D_engine_fallback = None, None, None
        self.timing = None               # set to {} to accumulate per-phase GPU milliseconds of optimize_parameters (bench.py --workload c3)
        if self.is_train:
            if train_opt['feature_weight'] is not None:
                raise NotImplementedError('The VGG-feature loss needs define_F (torchvision), which is outside this build; set train.feature_weight to null')
            if self.latent_input is not None and train_opt['latent_weight'] is not None:
                # L_struct: the structure tensor of G's output against the value the latent code asks for (reference :35-40)
                self.l_latent_w = train_opt['latent_weight']
                self.cri_latent = FilterLoss(latent_channels=opt['network_G']['latent_channels'])
                if not self.cri_latent.supported:
                    raise NotImplementedError("train.latent_weight with network_G.latent_channels = %r: only the structure-tensor codes are part of this build"
                                              % (opt['network_G']['latent_channels'],))
            if self.latent_input is not None and train_opt['optimalZ_loss_type'] is not None and train_opt['optimalZ_loss_weight'] is not None:
                self.optimalZ_loss_type = train_opt['optimalZ_loss_type']
            self.D_verification = train_opt['D_verification']
            if self.D_verification not in ['current', 'convergence', 'past', None]:
                raise ValueError("train.D_verification = %r: 'current', 'convergence', 'past' or null" % (self.D_verification,))
            self.D_verified = self.verified_D_saved = self.D_verification is None
            if self.D_verification == 'convergence':
                self.D_converged = False
            net_D = opt['network_D'] or {}
            self.relativistic_D = net_D.get('relativistic') is None or bool(net_D.get('relativistic'))
            self.add_quantization_noise = bool(net_D.get('add_quantization_noise'))
            self.grad_accumulation_steps_G = train_opt['grad_accumulation_steps_G'] or 1
            self.grad_accumulation_steps_D = train_opt['grad_accumulation_steps_D'] or self.grad_accumulation_steps_G
            self.min_accumulation_steps = min(self.grad_accumulation_steps_G, self.grad_accumulation_steps_D)
            self.max_accumulation_steps = accumulation_steps_per_batch
            self.decomposed_output = bool(self.CEM_arch and net_D.get('decomposed_input'))
            if self.decomposed_output:
                raise NotImplementedError('network_D.decomposed_input: the reference supports it for PatchGAN only, which is outside this build')
            self.netG.train()
            self.l_gan_w = train_opt['gan_weight']
            self.D_exists = self.l_gan_w is not None
            if self.D_exists:
                self.netD = networks.define_D(opt, CEM=self.CEM_net).to(self.device)
                esr_dist.broadcast_parameters(self.netD)
                self.netD.train()
                # network_D.precision = 'bf16': the critic's convolutions run under bf16 autocast (fp32 parameters, fp32 losses; configs[2] of
                # BASELINE.json names bf16); default: fp32 like the reference.  network_D.channels_last: NHWC activations for MIOpen.
                self.D_dtype = torch.bfloat16 if net_D.get('precision') == 'bf16' else None
                if net_D.get('channels_last'):
                    self.netD = self.netD.to(memory_format=torch.channels_last)
                # network_D.engine: 'hip' runs the critic, its backward and the penalty's double backward on the library's kernels (conv 3x3 s1 /
                # 4x4 s2 + BatchNorm + LeakyReLU blocks, Linear classifier: Discriminator_VGG_128); 'stock' keeps the nn.Module on MIOpen; 'auto'
                # (default) = 'hip' where the kernels cover the architecture and the input size, otherwise 'stock' — and it SAYS so: the reason is
                # printed, logged and kept in self.D_engine_fallback (None while the library's kernels are the ones running).  'hip' raises instead.
                want = net_D.get('engine') or 'auto'
                if want not in ('auto', 'hip', 'stock'):
                    raise NotImplementedError("network_D.engine = %r: 'hip', 'stock' or 'auto'" % (want,))
                self.D_engine_mode, self.D_engine_fallback = want, ('network_D.engine = stock' if want == 'stock' else None)
                if want != 'stock' and self.device.type == 'cuda':
                    try:
                        self.D_engine = CriticEngine(self.netD)
                    except EsrError as e:                # (the constructor checks the architecture only)
                        if want == 'hip':
                            raise
                        self._D_fall_back(str(e))
                # network_D.miopen_find (stock critic only): let MIOpen time its kernels once and keep the fastest (configs[2] shapes, bf16 critic:
                # 30.7 -> 20.0 ms per D step; costs tens of seconds of search the first time a shape is seen, so it is opt-in)
                if net_D.get('miopen_find'):
                    torch.backends.cudnn.benchmark = True
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
            self.cri_optimalZ = None
            if self.optimalZ_loss_type is not None:
                from Z_optimization import Z_optimizer
                if self.optimalZ_loss_type not in ('l1', 'l2'):
                    # ('hist' cannot be constructed in the reference either: its trainer builds Z_optimizer(objective='hist') without a desired image,
                    # and SoftHistogramLoss(desired_hist_image=None, patch_size=1) takes len() of it — codes/Z_optimization.py:72, :538-541)
                    raise NotImplementedError("train.optimalZ_loss_type = %r: 'l1' or 'l2' (the reference's 'hist' fails at construction: "
                                              "SoftHistogramLoss without a desired image)" % self.optimalZ_loss_type)
                self.l_g_optimalZ_w = train_opt['optimalZ_loss_weight']
                z_side = int(opt['datasets']['train']['patch_size'] / (opt['scale'] / self.Z_size_factor))
                self.Z_optimizer = Z_optimizer(objective=self.optimalZ_loss_type, Z_size=2 * [z_side], model=self, Z_range=1, max_iters=10, initial_LR=1,
                                               batch_size=opt['datasets']['train']['batch_size'], HR_unpadder=self.CEM_net.HR_unpadder)
                self.cri_optimalZ = (nn.MSELoss() if self.optimalZ_loss_type == 'l2' else nn.L1Loss()).to(self.device)
            self.cri_fea = None
            self.cri_range = None
            if train_opt['range_weight'] is not None:
                self.cri_range = CreateRangeLoss(opt['range'])
                self.l_range_w = train_opt['range_weight']
            self.GD_update_controller = None
            self.cri_gan, self.D_init_iters, self.global_D_update_ratio = None, 0, 1
            if self.D_exists:
                self.cri_gan = GANLoss(train_opt['gan_type'], 1.0, 0.0).to(self.device)
                self.global_D_update_ratio = train_opt['D_update_ratio'] if train_opt['D_update_ratio'] is not None else 1
                if isinstance(self.global_D_update_ratio, list):
                    raise NotImplementedError('train.D_update_ratio as a controller range (utils.util.G_D_updates_controller) is not part of this build; pass a number')
                self.D_init_iters = train_opt['D_init_iters'] if train_opt['D_init_iters'] else 0
                if train_opt['gan_type'] == 'wgan-gp':
                    self.cri_gp = GradientPenaltyLoss(device=self.device).to(self.device)
                    self.l_gp_w = train_opt['gp_weight']
            wd_G = train_opt['weight_decay_G'] if train_opt['weight_decay_G'] else 0
            optim_params = [v for k, v in self.netG.named_parameters() if v.requires_grad]
            self.lr_G, self.lr_D = train_opt['lr_G'], train_opt['lr_D']
            # (torch's fused Adam was measured on this part: 7.1 ms per step for the generator's 702 tensors against 1.4 ms for the default
            # foreach implementation — not used)
            fused = dict(fused=True) if train_opt['fused_adam'] else {}
            # esr_hip.optim.Adam: torch.optim.Adam's update for all 702 tensors as one kernel launch (train.torch_adam: true selects torch's)
            Adam = torch.optim.Adam if (train_opt['torch_adam'] or fused or self.device.type != 'cuda') else esr_optim.Adam
            self.optimizer_G = Adam(optim_params, lr=self.lr_G, weight_decay=wd_G,
                                    betas=(train_opt['beta1_G'] or 0.9, train_opt['beta2_G'] if train_opt['beta2_G'] is not None else 0.999), **fused)
            self.optimizers.append(self.optimizer_G)
            self.grad_reducer = esr_dist.EarlyBucketReducer(optim_params)      # train.early_gradient_exchange: exchanged from inside the backward (see _G_backward)
            if train_opt.get('early_gradient_exchange'):
                self.grad_reducer.ENABLED = True
            if self.D_exists:
                wd_D = train_opt['weight_decay_D'] if train_opt['weight_decay_D'] else 0
                # (the one-launch Adam steps contiguous fp32 tensors: a channels_last critic keeps torch's)
                plain = all(p.is_contiguous() and p.dtype == torch.float32 for p in self.netD.parameters())
                self.optimizer_D = (Adam if plain else torch.optim.Adam)(self.netD.parameters(), lr=self.lr_D, weight_decay=wd_D,
                                        betas=(train_opt['beta1_D'] or 0.9, train_opt['beta2_D'] if train_opt['beta2_D'] is not None else 0.999), **fused)
                self.optimizers.append(self.optimizer_D)
                self.grad_reducer_D = esr_dist.GradBucketAllReducer(list(self.netD.parameters()))
            if train_opt['lr_scheme'] == 'MultiStepLR':
                for optimizer in self.optimizers:
                    self.schedulers.append(torch.optim.lr_scheduler.MultiStepLR(optimizer, train_opt['lr_steps'], train_opt['lr_gamma']))
            elif train_opt['lr_scheme'] is not None:
                raise NotImplementedError('MultiStepLR learning rate scheme is enough.')
            self.generator_step = False
            self.gradient_step_num = 0
        elif init_Dnet:
            self.netD = networks.define_D(opt, CEM=self.CEM_net).to(self.device)
            self.netD.eval()
        if init_Fnet:
            raise NotImplementedError('init_Fnet: the VGG feature extractor needs torchvision')
        self.load()
        if self.is_train:
            # what the reference does after loading (:209-218): once the critic counts as verified, G trains at D's learning rate and the
            # optimal-Z search runs train.Num_Z_iterations[-1] iterations
            self.D_verified, self.verified_D_saved = bool(self.D_verified), bool(self.verified_D_saved)
            if self.D_exists:
                for group in self.optimizer_D.param_groups:
                    group['lr'] = self.lr_D
                if self.verified_D_saved:
                    self.lr_G = 1 * self.lr_D
                    if self.optimalZ_loss_type is not None:
                        self.Z_optimizer.max_iters = train_opt['Num_Z_iterations'][-1]
            for group in self.optimizer_G.param_groups:
                group['lr'] = self.lr_G
        if self.is_train and opt['gc_freeze'] is not False:
            # A training step allocates thousands of short-lived Python objects (launch descriptors, autograd nodes): CPython's cyclic collector
            # then runs several times per step, and its full passes walk every module / parameter / buffer object of the two networks — measured
            # 6-12 ms per step at the configs[2] shape (52.7 -> 47.0 ms for the G+D step, 37 -> 25 ms generator-only).  Everything alive now lives
            # as long as the model: move it to the permanent generation so that collections only look at what a step creates.
            import gc
            gc.collect()
            gc.freeze()
            self._gc_frozen = True          # close() undoes it
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
            elif self.cri_latent is not None:
                # training with L_struct: one latent code per image, constant over the image (reference :252-254); for the
                # 'SVD*structure_tensor' codes it is drawn as (lambda0, lambda1, theta) and converted (utils/util.py:285-291)
                cur_Z = torch.rand([self.var_L.size(0), self.num_latent_channels, 1, 1])
                if self.opt['network_G']['latent_channels'] in ['SVD_structure_tensor', 'SVDinNormedOut_structure_tensor']:
                    l0, l1, th = cur_Z[:, 0], cur_Z[:, 1], 2 * np.pi * cur_Z[:, -1]
                    self.SVD = {'theta': th, 'lambda0_ratio': 1 * l0, 'lambda1_ratio': 1 * l1}
                    cur_Z = torch.stack([2 * (l1 * torch.sin(th) ** 2 + l0 * torch.cos(th) ** 2) - 1, 2 * (l0 * torch.sin(th) ** 2 + l1 * torch.cos(th) ** 2) - 1,
                                         2 * (l0 - l1) * torch.sin(th) * torch.cos(th)], 1)
                else:
                    cur_Z = 2 * cur_Z - 1
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
        G = self.netG.module if hasattr(self.netG, 'module') else self.netG
        G = getattr(G, 'generated_image_model', G)
        if hasattr(G, 'check_range'):
            G.check_range()                    # fp16 precisions: an image built from saturated activations is an error, not a result
        self.netG.train()

    # ------------------------------------------------------------------ training step (reference :280-521)
    def _draw_interp_points(self, batch_size):
        """Interpolation points of the WGAN-GP penalty, one per image, U[0,1) (reference :364-367).  A method so that parity tests can
        replay the reference's draws."""
        return torch.rand(batch_size, 1, 1, 1, device=self.device)

    def close(self):
        """Undo the process-wide side effect of construction (gc.freeze(), see __init__): a process that builds a second model — validation,
        the GUI — calls this when it drops a trained one, so that its objects become collectable again."""
        if getattr(self, '_gc_frozen', False):
            import gc
            gc.unfreeze()
            self._gc_frozen = False

    def _G_backward(self, loss, early):
        """loss.backward() — with the gradient exchange of a multi-rank job started from inside the generator's backward pass when this is the
        step's only backward into the generator (no accumulation: the .grad tensors are the flat buffer's views themselves): the engine launches
        its weight gradients bucket by bucket and every bucket's all-reduce overlaps the launches behind it (esr_hip.dist.EarlyBucketReducer)."""
        G = self.netG.module if hasattr(self.netG, 'module') else self.netG
        G = getattr(G, 'generated_image_model', G)
        eng = getattr(G, '_engine', None)
        use = (early and eng is not None and self.grad_reducer.ENABLED and esr_dist.is_distributed() and
               all(p.grad is None for p in self.grad_reducer.params))
        if use:
            eng.wgrad_exchange = self.grad_reducer
        try:
            loss.backward()
        finally:
            if use:
                eng.wgrad_exchange = None

    def _D_fall_back(self, reason, permanent=True):
        """network_D.engine = 'auto' leaving the library's kernels for the stock nn.Module: never silently.  permanent=False: for this call only
        (an input size the kernels do not run — a ragged validation crop); the engine stays for the sizes it covers."""
        import logging
        if permanent:
            self.D_engine = None
        first = self.D_engine_fallback != reason
        self.D_engine_fallback = reason
        if first:                              # (a per-call fallback repeats: one line per distinct reason, not one per batch)
            msg = "network_D.engine = 'auto': the critic runs as the stock nn.Module on MIOpen, NOT on libesr_hip (%s)%s" % (
                reason, '' if permanent else ' — for inputs of this size only')
            logging.getLogger('base').warning(msg)
            print('WARNING: ' + msg)

    def _D_engine_for(self, xs):
        """The critic engine if it runs inputs of the sizes of `xs` (a tensor or a list of them; CriticEngine.unsupported_input), else the
        documented fallback for THIS call (the engine is kept: later batches of a supported size run on it again) / the error."""
        if self.D_engine is None:
            return None
        for x in (xs if isinstance(xs, (list, tuple)) else (xs,)):
            reason = self.D_engine.unsupported_input(x.shape[-2], x.shape[-1])
            if reason is not None:
                if self.D_engine_mode == 'hip':
                    raise EsrError("network_D.engine = 'hip': " + reason)
                self._D_fall_back(reason, permanent=False)
                return None
        if self.D_engine_fallback is not None and self.D_engine_mode != 'stock':
            self.D_engine_fallback = None      # back on the library's kernels
        self.D_engine.set_precision('bf16' if self.D_dtype is torch.bfloat16 else 'split')
        return self.D_engine

    def _D(self, x):
        """The critic's logits in fp32: on the library's kernels (esr_hip/critic.py; 'bf16' operands when network_D.precision = 'bf16', the
        fp32-class 'split' otherwise), or — network_D.engine = 'stock', or an architecture the kernels do not cover — the nn.Module on
        MIOpen (under bf16 autocast for 'bf16')."""
        eng = self._D_engine_for(x)
        if eng is not None:
            return critic_forward(eng, x)
        if self.D_dtype is None:
            return self.netD(x)
        with torch.autocast(device_type='cuda', dtype=self.D_dtype):
            return self.netD(x).float()

    def _D_group(self, xs):
        """[self._D(x) for x in xs] — on the library's kernels as ONE pass over the concatenated batches (same values: every batch keeps its own
        BatchNorm statistics, the running statistics see them in this order; esr_hip.critic.critic_forward_group)."""
        eng = self._D_engine_for(xs)
        if eng is not None:
            return critic_forward_group(eng, xs)
        return [self._D(x) for x in xs]

    def _tick(self, name):
        """Phase timer (only when self.timing is a dict): GPU time since the previous tick is charged to `name`."""
        if self.timing is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if self._last_ev is not None:
            self._phase_evs.append((name, self._last_ev, ev))
        self._last_ev = ev

    def optimize_parameters(self):
        train_opt = self.opt['train']
        acc_G, acc_D = self.grad_accumulation_steps_G, self.grad_accumulation_steps_D
        self.gradient_step_num = self.step // max(self.max_accumulation_steps, 1)
        first_acc_G, last_acc_G = self.step % acc_G == 0, self.step % acc_G == acc_G - 1
        first_acc_D, last_acc_D = self.step % acc_D == 0, self.step % acc_D == acc_D - 1
        self._last_ev, self._phase_evs = None, []
        self._tick('start')
        # ---- which networks step this time (:287-306)
        if first_acc_G:
            self.generator_step = self.gradient_step_num > self.D_init_iters
            if self.generator_step:
                self.generator_step = self.gradient_step_num % max(1, self.global_D_update_ratio) == 0
                # with a larger D batch, G steps on the last accumulation steps of D only
                self.generator_step = self.generator_step and self.step % acc_D >= acc_D - acc_G
        if self.D_exists and first_acc_D:
            self.discriminator_step = self.gradient_step_num >= -self.D_init_iters
            if self.discriminator_step and self.verified_D_saved:
                self.discriminator_step = self.gradient_step_num % max(1, np.ceil(1 / self.global_D_update_ratio)) == 0
        G_grads_retained = first_acc_D or self.generator_step
        self.Set_Require_Grad_Status(self.netG, bool(G_grads_retained))
        dual_steps = int(self.optimalZ_loss_type is not None and self.generator_started_learning) + 1
        for dual in range(dual_steps):
            optimized_Z_step = dual == dual_steps - 2     # first the optimised-Z pass (its Z search needs no G gradients), then the assigned Z
            first_dual, last_dual = dual == 0, dual == dual_steps - 1
            if self.CEM_net is not None and first_dual:
                # the losses and the critic ignore the frame the CEM cannot constrain (:319-320)
                self.var_H, self.var_ref = self.CEM_net.HR_unpadder(self.var_H), self.CEM_net.HR_unpadder(self.var_ref)
            if first_dual:
                static_Z = self.GetLatent() if self.latent_input is not None else None
            if optimized_Z_step:
                self.Z_optimizer.feed_data({'LR': self.var_L, 'desired': self.var_H})
                self.Z_optimizer.optimize()               # leaves self.fake_H = G(optimal Z) with G's graph
            else:
                self.Prepare_Input(LR_image=self.var_L, latent_input=static_Z)
                self.fake_H = None                            # drop the previous step's graph first: its saved activation buffers are then free for reuse
                self.fake_H = self.netG(self.model_input)     # train mode: no pre-padding
            if self.CEM_net is not None:
                self.fake_H = self.CEM_net.HR_unpadder(self.fake_H)
            self._tick('G_forward')
            # ---- D step (:340-414)
            if not self.D_exists:
                self.generator_step = self.gradient_step_num > 0       # one idle first iteration, as the reference (initial validation)
            elif self.discriminator_step:
                self.Set_Require_Grad_Status(self.netD, True)
                self.Set_Require_Grad_Status(self.netG, False)
                if first_acc_D and first_dual:
                    self.optimizer_D.zero_grad()
                    self._d_acc = {k: [] for k in ('l_d_real', 'l_d_fake', 'D_real', 'D_fake', 'D_logits_diff')}
                # the critic's calls of this step — real (first dual pass only), fake, and the penalty's interpolated batch — as one grouped pass
                # (the reference calls netD three times, :345-368; its only random draw in between is the interpolation points)
                d_inputs = ([self.var_ref] if first_dual else []) + [self.fake_H.detach()]
                if train_opt['gan_type'] == 'wgan-gp':
                    pt = self._draw_interp_points(self.var_ref.size(0))
                    interp = (pt * self.fake_H.detach() + (1 - pt) * self.var_ref).requires_grad_(True)
                    d_inputs.append(interp)
                d_preds = self._D_group(d_inputs)
                if first_dual:
                    pred_d_real = d_preds[0]
                pred_d_fake = d_preds[1 if first_dual else 0]
                if self.relativistic_D:
                    assert train_opt['hinge_threshold'] is None, 'Unsupported yet, should think whether it reuires special adaptation of hinge loss'
                    l_d_real = self.cri_gan(pred_d_real - torch.mean(pred_d_fake), True)
                    l_d_fake = self.cri_gan(pred_d_fake - torch.mean(pred_d_real), False)
                else:
                    if first_dual:
                        l_d_real = 2 * self.cri_gan(pred_d_real, True, train_opt['hinge_threshold'])
                    l_d_fake = 2 * self.cri_gan(pred_d_fake, False, train_opt['hinge_threshold'])
                l_d_total = (l_d_real + l_d_fake) / 2
                if train_opt['gan_type'] == 'wgan-gp':
                    l_d_gp = self.l_gp_w * self.cri_gp(interp, d_preds[-1], critic_group=len(d_inputs) - 1)
                    l_d_total = l_d_total + l_d_gp
                logits_diff = torch.mean(pred_d_real.detach() - pred_d_fake.detach(), dim=list(range(1, pred_d_real.dim())))
                self._d_acc['l_d_real'].append(l_d_real.detach()); self._d_acc['l_d_fake'].append(l_d_fake.detach())
                self._d_acc['D_real'].append(pred_d_real.detach().mean()); self._d_acc['D_fake'].append(pred_d_fake.detach().mean())
                self._d_acc['D_logits_diff'].append(logits_diff)
                if first_acc_D and first_dual and self.generator_step and self.D_verification == 'past' and (train_opt['D_valid_Steps_4_G_update'] or 0) > 0:
                    n = train_opt['D_valid_Steps_4_G_update']
                    log = self.log_dict
                    self.generator_step = len(log['D_logits_diff']) >= n and \
                        all(v[1] > np.log(train_opt['min_D_prob_ratio_4_G']) for v in log['D_logits_diff'][-n:]) and \
                        all(v[1] > train_opt['min_mean_D_correct'] for v in log['Correctly_distinguished'][-n:])
                if first_acc_D and first_dual and self.generator_step and self.D_verification == 'convergence':
                    # G waits until the critic's losses have levelled out (reference :383-393): a line is fitted to the logged l_d_real / l_d_fake of the
                    # last steps_4_loss_std steps; converged once lr_change_ratio * |slope| (at least 1e-5) is below the residual's standard
                    # deviation.  A host read of the log, as in the reference; under data parallelism the ranks' logs hold their own shards'
                    # losses, so the decision is made global (all ranks converged) before anybody acts on it.
                    if not self.D_converged and self.gradient_step_num >= train_opt['steps_4_D_convergence']:
                        log = self.log_dict
                        std = slope = 0.0
                        fit_ok = True
                        for key in ('l_d_real', 'l_d_fake'):
                            vals = [v[1] for v in log[key] if v[0] >= self.gradient_step_num - train_opt['steps_4_loss_std']]
                            if len(vals) < 4:                   # numpy needs more points than parameters + 2 for the covariance
                                fit_ok = False
                                break
                            (cur_slope, _), ((cur_var, _), _) = np.polyfit(np.arange(len(vals)), vals, 1, cov=True)
                            std += 0.5 * np.sqrt(cur_var)
                            slope += 0.5 * cur_slope
                        conv = bool(fit_ok and -train_opt['lr_change_ratio'] * np.minimum(-1e-5, slope) < std)
                        flags = esr_dist.gather_scalars(torch.tensor([1.0 if conv else 0.0], device=self.device))
                        self.D_converged = bool(flags.min().item() > 0.5)
                    self.generator_step = bool(self.D_converged)
                if self.D_verification == 'current' and self.generator_step:
                    # a host read, as in the reference: this mode gates G on the current batch — the GLOBAL batch under data parallelism (all
                    # ranks must take the same decision: the generator's all-reduce is only entered by ranks that do a G step)
                    stats = esr_dist.gather_scalars(torch.stack([logits_diff.min(), logits_diff.mean()])).cpu().numpy()
                    self.generator_step = bool(stats[:, 0].min() > 0 and stats[:, 1].mean() > np.log(train_opt['min_D_prob_ratio_4_G']))
                if G_grads_retained and not self.generator_step:
                    self.fake_H = self.fake_H.detach()     # nobody will back-propagate into G this time
                # pred_d_real / l_d_real are computed on the first dual pass only and re-used by the second one: their graph has to survive the
                # first backward (the reference retains it whenever gan_type is 'wgan-gp' or a G step follows, :400; retaining it only while
                # another dual pass will back-propagate through it is the same computation without holding the buffers longer)
                (l_d_total / (acc_D * dual_steps)).backward(retain_graph=not last_dual)
                if last_acc_D and last_dual:
                    self.grad_reducer_D()                  # RCCL all-reduce (mean) of the D gradients
                    self.optimizer_D.step()
                    st = self.gradient_step_num
                    for k in ('l_d_real', 'l_d_fake', 'D_real', 'D_fake'):
                        self._pending_logs.append((k, st, self._d_acc[k]))
                    self._pending_logs.append(('l_d_real_fake', st, [torch.stack(self._d_acc['l_d_real']).mean() + torch.stack(self._d_acc['l_d_fake']).mean()]))
                    if train_opt['gan_type'] == 'wgan-gp':
                        self._pending_logs.append(('l_d_gp', st, [l_d_gp.detach()]))
                    diffs = torch.cat([d.reshape(-1) for d in self._d_acc['D_logits_diff']])
                    # these two gate later generator steps (D_verification = 'past'): every rank logs the mean over ALL ranks' shards
                    gate = esr_dist.gather_scalars(torch.stack([diffs.mean(), (diffs > 0).float().mean()])).mean(0)
                    self._pending_logs.append(('D_logits_diff', st, [gate[0]]))
                    self._pending_logs.append(('Correctly_distinguished', st, [gate[1]]))
                self._tick('D_step')
            # ---- G step (:418-499)
            if self.generator_step:
                self.generator_started_learning = True
                if self.D_exists:
                    self.Set_Require_Grad_Status(self.netD, False)
                self.Set_Require_Grad_Status(self.netG, True)
                if first_acc_G and first_dual:
                    self.optimizer_G.zero_grad()
                    self._g_acc = {k: [] for k in ['l_g_pix', 'l_g_range', 'l_g_gan', 'l_g_optimalZ'] + ['l_g_latent_%d' % i for i in range(self.num_latent_channels)]}
                scale = acc_G * dual_steps
                l_g_total = 0
                if self.cri_pix is not None:
                    if train_opt['pixel_domain'] == 'LR':
                        raise NotImplementedError("train.pixel_domain = 'LR'")
                    l_g_pix = self.cri_pix(self.fake_H, self.var_H)
                    l_g_total = l_g_total + self.l_pix_w * l_g_pix / scale
                    self._g_acc['l_g_pix'].append(l_g_pix.detach())
                if self.cri_range is not None:
                    l_g_range = self.cri_range(self.fake_H)
                    l_g_total = l_g_total + self.l_range_w * l_g_range / scale
                    self._g_acc['l_g_range'].append(l_g_range.detach())
                if self.cri_latent is not None and last_dual:
                    l_g_latent = self.cri_latent({'SR': self.fake_H, 'HR': self.var_H, 'Z': static_Z}).mean(0)      # [num_latent_channels]
                    l_g_total = l_g_total + self.l_latent_w * l_g_latent.mean() / acc_G
                    for i in range(self.num_latent_channels):
                        self._g_acc['l_g_latent_%d' % i].append(l_g_latent[i].detach())
                if self.cri_optimalZ is not None and first_dual:
                    l_g_optimalZ = self.cri_optimalZ(self.fake_H, self.var_H)
                    l_g_total = l_g_total + self.l_g_optimalZ_w * l_g_optimalZ / acc_G
                    self._g_acc['l_g_optimalZ'].append(l_g_optimalZ.detach())
                if self.D_exists:
                    pred_g_fake = self._D(self.fake_H)
                    if self.relativistic_D:
                        pred_d_real = self._D(self.var_ref).detach()
                        l_g_gan = self.l_gan_w * (self.cri_gan(pred_d_real - torch.mean(pred_g_fake), False) +
                                                  self.cri_gan(pred_g_fake - torch.mean(pred_d_real), True)) / 2 / scale
                    else:
                        l_g_gan = self.l_gan_w * self.cri_gan(pred_g_fake, True) / scale
                    l_g_total = l_g_total + l_g_gan
                    self._g_acc['l_g_gan'].append(l_g_gan.detach())
                self._G_backward(l_g_total, early=(last_acc_G and last_dual and scale == 1))
                self._tick('G_losses_and_backward')
                # The reference reads every loss value back right here (.item(), :483-493): a host synchronisation per step.  Here they
                # stay on the device and are read when somebody looks at the log (log_dict / get_current_log).
                if last_acc_G and last_dual:
                    self.grad_reducer()                    # RCCL all-reduce (mean) of the G gradients: the one exchange step
                    self.optimizer_G.step()
                    self.generator_changed = True
                    for k, v in self._g_acc.items():
                        if v:
                            self._pending_logs.append((k, self.gradient_step_num, v))
                    self._tick('G_allreduce_and_Adam')
        self.step += 1
        if self.timing is not None:
            torch.cuda.synchronize()
            for name, e0, e1 in self._phase_evs:
                self.timing[name] = self.timing.get(name, 0.0) + e0.elapsed_time(e1)

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
        if self.is_train and self.D_exists:
            s, n = self.get_network_description(self.netD)
            print('Number of parameters in D: {:,d}'.format(n))

    # ------------------------------------------------------------------ validation (reference :533-590)
    def perform_validation(self, data_loader, cur_Z, print_rlt, first_eval, save_images):
        """PSNR of the validation set for the latent value `cur_Z` (reference :533-590); print_rlt['psnr'] is incremented by the
        average.  Returns the SR images as float32 HWC arrays in [0, 255].  The reference also writes image collages through OpenCV
        when save_images is set; here the collage is kept in self.im_collages and written as .npy (no image codec in this build)."""
        psnrs, sr_images, rows = [], [], []
        for val_data in data_loader:
            val_data = dict(val_data)
            val_data['Z'] = cur_Z
            self.feed_data(val_data)
            self.test()
            visuals = self.get_current_visuals()
            sr = 255 * np.clip(visuals['SR'].numpy(), 0, 1).transpose(1, 2, 0).astype(np.float32)
            gt = 255 * np.clip(visuals['HR'].numpy(), 0, 1).transpose(1, 2, 0).astype(np.float32)
            sr_images.append(sr)
            mse = float(np.mean((sr.astype(np.float64) - gt.astype(np.float64)) ** 2))
            psnrs.append(float('inf') if mse == 0 else 20 * np.log10(255.0 / np.sqrt(mse)))
            if save_images:
                rows.append(np.clip(sr, 0, 255).astype(np.uint8))
        avg_psnr = float(np.mean(psnrs))
        if save_images:
            self.generator_changed = False
            side = min(min(r.shape[:2]) for r in rows)
            collage = np.concatenate([r[:side, :side] for r in rows], 1)
            if not hasattr(self, 'im_collages'):
                self.im_collages = []
            self.im_collages.append(collage)
            val_dir = self.opt['path']['val_images']
            if val_dir and esr_dist.rank() == 0:
                os.makedirs(val_dir, exist_ok=True)
                np.save(os.path.join(val_dir, '{:d}_{}PSNR{:.3f}.npy'.format(self.gradient_step_num, ('Z' + str(cur_Z)) if self.latent_input else '', avg_psnr)), collage)
        print_rlt['psnr'] += avg_psnr
        return sr_images

    # ------------------------------------------------------------------ logs.npz (reference :644-675)
    ADDITIONALLY_SAVED_ATTRIBUTES = ['D_verified', 'verified_D_saved', 'lr_G', 'lr_D']

    def save_log(self):
        if esr_dist.rank() != 0:
            return
        to_save = {k: np.array(v, dtype=np.float64).reshape(-1, 2) for k, v in self.log_dict.items()}
        for attr in self.ADDITIONALLY_SAVED_ATTRIBUTES:
            if attr in self.__dict__ and getattr(self, attr) is not None:
                to_save[attr] = getattr(self, attr)
        os.makedirs(self.log_path, exist_ok=True)
        np.savez(os.path.join(self.log_path, 'logs.npz'), **to_save)

    def load_log(self, max_step=None):
        loaded = np.load(os.path.join(self.log_path, 'logs.npz'))
        self.log_dict                                     # flush pending values before replacing the history
        for key in list(self._log_dict):
            self._log_dict[key] = []
        for key in loaded.files:
            if key in self.ADDITIONALLY_SAVED_ATTRIBUTES:
                v = loaded[key]
                setattr(self, key, v.item() if v.ndim == 0 else v)
                continue
            pairs = [(int(p[0]), float(p[1])) for p in loaded[key]]
            if max_step is not None:
                pairs = [p for p in pairs if p[0] <= max_step]
            self._log_dict[key] = pairs

    # ------------------------------------------------------------------ checkpoints (reference :732-776)
    def load(self, max_step=None, resume_train=None):
        path_opt = self.opt['path'] or {}
        resume = resume_train if resume_train is not None else (self.is_train and bool(self.opt['train']['resume']))
        own = sorted(int(f.split('_')[0]) for f in (os.listdir(self.save_dir) if os.path.isdir(self.save_dir) else [])
                     if f.endswith('_G.pth') and f.split('_')[0].isdigit())
        if max_step is not None:
            own = [st for st in own if st <= max_step]
        load_own = bool(own) and (max_step is not None or resume or not self.is_train)
        if load_own:
            st = own[-1]
            path_G = os.path.join(self.save_dir, '%d_G.pth' % st)
            path_D = os.path.join(self.save_dir, '%d_D.pth' % st)
            if self.is_train:
                self.step = (st + 1) * max(self.max_accumulation_steps, 1)
                print('Resuming training with model for G [{:s}] ...'.format(path_G))
                self.load_network(path_G, self.netG, optimizer=self.optimizer_G)
                if os.path.exists(os.path.join(self.log_path or '', 'logs.npz')):
                    self.load_log(max_step=st)
                if self.D_exists:
                    print('Resuming training with model for D [{:s}] ...'.format(path_D))
                    self.load_network(path_D, self.netD, optimizer=self.optimizer_D)
            else:
                print('Testing model for G [{:s}] ...'.format(path_G))
                self.load_network(path_G, self.netG)
                if 'netD' in self.__dict__ and os.path.exists(path_D):
                    self.load_network(path_D, self.netD)
                self.gradient_step_num = st
            return
        load_path_G = path_opt.get('pretrained_model_G') or path_opt.get('pretrain_model_G')      # the reference reads 'pretrained_model_G'
        if load_path_G is not None:
            print('loading model for G [{:s}] ...'.format(load_path_G))
            self.load_network(load_path_G, self.netG)
        load_path_D = path_opt.get('pretrained_model_D') or path_opt.get('pretrain_model_D')
        if self.is_train and self.D_exists and load_path_D is not None:
            print('loading model for D [{:s}] ...'.format(load_path_D))
            self.load_network(load_path_D, self.netD, optimizer=self.optimizer_D)

    def save(self, iter_label):
        if esr_dist.rank() != 0:
            return None
        path = self.save_network(self.save_dir, self.netG, 'G', iter_label, self.optimizer_G)
        if self.D_exists:
            self.save_network(self.save_dir, self.netD, 'D', iter_label, self.optimizer_D)
        return path
