"""Latent-space (Z) optimisation through the frozen generator — the loop of the reference's codes/Z_optimization.py
(Optimizable_Z :273-319, Z_optimizer.optimize :647-797): tanh-bounded Z parameter, Adam, per iteration
    Z = Z_range*tanh(P)  ->  model.feed_data({'LR', 'Z'})  ->  model.test(prevent_grads_calc=False)   (G + CEM forward WITH graph,
    weights frozen)  ->  clamp(0,1)  ->  objective  ->  loss.mean().backward()  (data-gradient kernels only)  ->  Adam step,
keeping the iterate with the smallest loss.  The forward/backward are the HIP kernels; the objectives below are element-wise /
reduction torch ops on the SR output.  Implemented objectives: 'max_STD', 'min_STD', 'STD_increase', 'STD_decrease', 'TV', 'l1',
whole-image or restricted to a user-marked region (image_mask: where the objective looks; Z_mask: which latent entries may move — the
GUI's region tools, GUI.py:1925-2057).  The GUI's scribble / histogram / periodicity / dictionary / adversarial / 'local_*' patch objectives are
not part of this build and raise NotImplementedError.

Multi-GPU: the Z batch is sharded over ranks (independent samples, no data-path collective).  Like the reference, the loss is the
mean over the WHOLE batch, so each shard scales its local sum by 1/B_global; the loss history that picks the best iterate is
all-reduced (one scalar per iteration).
"""
import numpy as np
import torch

from esr_hip import dist as esr_dist


def ArcTanH(input_tensor):
    eps = torch.finfo(input_tensor.dtype).eps
    return 0.5 * torch.log((1 + input_tensor + eps) / (1 - input_tensor + eps))


def TV_Loss(image, mask=None, clamp01=False):
    """reference :324-326.  GPU tensors: one reduction kernel (esr_hip.zobj.tv_loss; `mask` / `clamp01` fold the image mask and the clamp of
    Output_Batch(within_0_1=True) into its read); CPU tensors: the defining torch expression."""
    if image.is_cuda:
        from esr_hip import zobj
        return zobj.tv_loss(image, mask, clamp01)
    if clamp01:
        image = torch.clamp(image, 0, 1)
    if mask is not None:
        image = image * mask
    return (image[:, :, :, :-1] - image[:, :, :, 1:]).abs().mean(dim=(1, 2, 3)) + (image[:, :, :-1, :] - image[:, :, 1:, :]).abs().mean(dim=(1, 2, 3))


class SoftHistogramLoss(torch.nn.Module):
    """KL divergence between the soft gray-level histogram of the produced image(s) and that of a desired image — the 'hist' Z objective
    (reference Z_optimization.py:24-230), in the form the whole-image tool uses: gray scale, patch size 1, fixed temperature.
        h[k] = mean_i exp(-(d(v_i, c_k) + 1e-7)^2 / T) over the (masked) pixels, c = linspace(min, max, bins), d wrapped with period max;
        p = h / sum(h);   loss = KLDivLoss()(log(p_current + eps) stacked over the batch, p_desired)      (:170-209, :211-229)
    The O(pixels x bins) histogram and its gradient are the HIP kernels of csrc/esr_zobj.hip (the reference materialises that matrix in
    float64).  Patch (KDE) histograms, dictionaries and the automatic temperature search are not part of this build."""

    def __init__(self, bins, min, max, desired_hist_image_mask=None, desired_hist_image=None, gray_scale=True, input_im_HR_mask=None, patch_size=1,
                 automatic_temperature=False, image_Z=None, temperature=0.05, dictionary_not_histogram=False, no_patch_DC=False, no_patch_STD=False):
        super(SoftHistogramLoss, self).__init__()
        if not gray_scale or patch_size != 1 or automatic_temperature or dictionary_not_histogram:
            raise NotImplementedError('SoftHistogramLoss: gray-scale, patch_size 1, fixed temperature histograms only')
        self.bins_n, self.min, self.max, self.temperature = int(bins), float(min), float(max), float(temperature)
        self.SQRT_EPSILON = 1e-7
        self.image_mask = None if input_im_HR_mask is None else input_im_HR_mask.reshape(-1).bool()
        self.loss = torch.nn.KLDivLoss()
        self.desired_hists_list = []
        if desired_hist_image is not None:
            # (the reference ignores desired_hist_image_mask for non-patch gray histograms: it is applied in its KDE branch only, :74-76)
            self.Feed_Desired_Hist_Im([im[0] if im.dim() == 4 else im for im in desired_hist_image][:1])

    def _hist(self, gray_values, log):
        from esr_hip import zobj
        h = zobj.soft_histogram(gray_values, self.bins_n, self.min, self.max, self.temperature, self.SQRT_EPSILON)
        h = (h / h.sum()).float()
        return torch.log(h + torch.finfo(h.dtype).eps).view(1, -1) if log else h.view(1, -1)

    def Feed_Desired_Hist_Im(self, desired_hist_image):
        self.desired_hists_list = []
        with torch.no_grad():
            for im in desired_hist_image:
                self.desired_hists_list.append(self._hist(im.mean(0).reshape(-1), log=False).detach())

    def forward(self, cur_images):
        logs = []
        for im in cur_images:
            gray = im.mean(0).reshape(-1)
            if self.image_mask is not None:
                gray = gray[self.image_mask.to(gray.device)]
            logs.append(self._hist(gray, log=True))
        return self.loss(torch.cat(logs, 0), torch.cat(self.desired_hists_list, 0).to(logs[0].device)).float()


class Optimizable_Z(torch.nn.Module):
    def __init__(self, Z_shape, Z_range=None, initial_pre_tanh_Z=None, Z_mask=None, random_perturbations=False, device=None):
        super(Optimizable_Z, self).__init__()
        device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
        self.Z = torch.nn.Parameter(data=torch.zeros(Z_shape, dtype=torch.float32, device=device))
        self.mask = None
        if Z_mask is not None and not np.all(Z_mask):
            self.mask = torch.from_numpy(np.asarray(Z_mask, dtype=np.float32)).to(device)
            self.initial_pre_tanh_Z = (1 * initial_pre_tanh_Z).float().to(device)
        if initial_pre_tanh_Z is not None:
            assert initial_pre_tanh_Z.size()[1:] == self.Z.data.size()[1:] and (initial_pre_tanh_Z.size(0) in [1, self.Z.data.size(0)]), \
                'Initilizer size does not match desired Z size'
            if random_perturbations:
                initial_pre_tanh_Z = initial_pre_tanh_Z + 0.001 * torch.randn_like(initial_pre_tanh_Z)
            self.Z.data[:initial_pre_tanh_Z.size(0), ...] = initial_pre_tanh_Z.to(device)
        self.Z_range = Z_range

    def forward(self):
        if self.Z_range is not None:
            fmax = torch.finfo(self.Z.dtype).max
            self.Z.data = torch.clamp(self.Z.data, -fmax, fmax)
        if self.mask is not None:
            self.Z.data = self.mask * self.Z.data + (1 - self.mask) * self.initial_pre_tanh_Z
        return self.Z_range * torch.tanh(self.Z) if self.Z_range is not None else self.Z

    def PreTanhZ(self):
        return self.mask * self.Z.data + (1 - self.mask) * self.initial_pre_tanh_Z if self.mask is not None else self.Z.data

    def Randomize_Z(self, what_2_shuffle):
        assert what_2_shuffle in ['all', 'allButFirst']
        torch.nn.init.xavier_uniform_(self.Z.data if what_2_shuffle == 'all' else self.Z.data[1:], gain=100)

    def Return_Detached_Z(self):
        return self.forward().detach()

    def Assign_Z(self, Z):
        self.Z.data = 1 * Z


class Z_optimizer():
    MIN_LR = 1e-5
    SUPPORTED = ['max_STD', 'min_STD', 'STD_increase', 'STD_decrease', 'TV', 'l1', 'hist']

    def __init__(self, objective, Z_size, model, Z_range, max_iters, data=None, loggers=None, image_mask=None, Z_mask=None, initial_Z=None,
                 initial_LR=None, existing_optimizer=None, batch_size=1, HR_unpadder=None, random_Z_inits=False, **unsupported):
        if objective not in self.SUPPORTED or ((image_mask is not None or Z_mask is not None) and 'l1' in objective):
            raise NotImplementedError("Z objective '%s': implemented are %s (optionally with image_mask / Z_mask, except 'l1'); the GUI's other "
                                      "editing objectives are not part of this build" % (objective, self.SUPPORTED))
        assert (image_mask is None) == (Z_mask is None), 'Should either supply both masks or niether'        # (reference :384)
        self.objective, self.model, self.data, self.loggers = objective, model, data, loggers
        self.device = model.device
        initial_pre_tanh_Z = None
        if initial_Z is not None:
            initial_pre_tanh_Z = initial_Z / Z_range
            eps = torch.finfo(initial_pre_tanh_Z.dtype).eps
            initial_pre_tanh_Z = ArcTanH(torch.clamp(initial_pre_tanh_Z, min=-1 + eps, max=1. - eps))
        self.model_training = HR_unpadder is not None
        # this rank's shard of the Z batch (all of it when not distributed)
        self.global_batch = batch_size
        self.shard = esr_dist.shard_range(batch_size)
        local_bs = self.shard[1] - self.shard[0]
        if initial_pre_tanh_Z is not None and initial_pre_tanh_Z.size(0) == batch_size and batch_size > 1:
            initial_pre_tanh_Z = initial_pre_tanh_Z[self.shard[0]:self.shard[1]]
        if Z_mask is not None and initial_pre_tanh_Z is None:       # a masked search keeps the unmasked entries at the model's current latent
            z_now = model.GetLatent() / Z_range
            eps = torch.finfo(z_now.dtype).eps
            initial_pre_tanh_Z = ArcTanH(torch.clamp(z_now, min=-1 + eps, max=1. - eps))
        self.Z_model = Optimizable_Z(Z_shape=[local_bs, model.num_latent_channels] + list(Z_size), Z_range=Z_range,
                                     initial_pre_tanh_Z=initial_pre_tanh_Z, Z_mask=Z_mask, random_perturbations=random_Z_inits, device=self.device)
        assert (initial_LR is not None) or (existing_optimizer is not None), 'Should either supply optimizer from previous iterations or initial LR for new optimizer'
        self.image_mask = None if image_mask is None else torch.from_numpy(np.asarray(image_mask, dtype=np.float32)).to(self.device)
        if not self.model_training and 'fake_H' in model.__dict__:
            self.initial_output = model.Output_Batch(within_0_1=True).detach()
            # every sample's own initial STD (the reference's first_image_only flag is honoured by its 'local' objectives only,
            # Z_optimization.py:617-627): per-sample reference points, so sharding the batch over ranks needs no exchange
            self.initial_STD = self.Masked_STD(first_image_only=True).detach()
        if 'STD' in objective and any(p in objective for p in ['increase', 'decrease']):
            STD_CHANGE_FACTOR = 1.05
            self.desired_STD = 1 * self.initial_STD
            inc = data.get('STD_increment') if data is not None else None
            if inc is None:
                self.desired_STD = self.desired_STD * (STD_CHANGE_FACTOR if 'increase' in objective else 1 / STD_CHANGE_FACTOR)
            else:
                self.desired_STD = self.desired_STD + (inc if 'increase' in objective else -inc)
        if 'l1' in objective and data is not None and 'desired' in data:
            self.desired_im = data['desired'].to(self.device)
        if objective == 'hist':          # reference :536-541: 256 bins on [0, 1], temperature 5e-4
            self.loss = SoftHistogramLoss(bins=256, min=0, max=1, desired_hist_image=[d.to(self.device) for d in data['desired']] if data is not None else None,
                                          desired_hist_image_mask=data.get('Desired_Im_Mask') if data is not None else None, input_im_HR_mask=self.image_mask,
                                          gray_scale=True, patch_size=1, temperature=5e-4)
        self.optimizer = torch.optim.Adam(self.Z_model.parameters(), lr=initial_LR) if existing_optimizer is None else existing_optimizer
        self.LR = initial_LR
        self.cur_iter = 0
        self.max_iters = max_iters
        self.random_Z_inits = 'all' if (random_Z_inits or self.model_training) else False
        self.HR_unpadder = HR_unpadder
        self.STD_PRESERVING_WEIGHT = 100 if 'TV' in objective else 20      # reference Z_optimization.py:508-509 (TV), :471 (others)

    def Masked_STD(self, first_image_only=False):
        # whole-image objectives: the STD of EVERY sample, [1, B], whatever the flag says (as the reference, see __init__)
        if self.model.fake_H.is_cuda:        # clamp, mask and the two moments in one pass over the batch (esr_img_stats)
            from esr_hip import zobj
            return zobj.image_std(self.model.fake_H, self.image_mask, clamp01=True).view(1, -1)
        out = self.model.Output_Batch(within_0_1=True)
        return torch.std(out if self.image_mask is None else out * self.image_mask, dim=(1, 2, 3)).view(1, -1)

    def feed_data(self, data):
        self.data = data
        self.cur_iter = 0
        if 'l1' in self.objective:
            self.desired_im = data['desired'].to(self.device)

    def Manage_Model_Grad_Requirements(self, verify_disabled):
        if verify_disabled:
            self.original_requires_grad_status = []
            for p in self.model.netG.parameters():
                self.original_requires_grad_status.append(p.requires_grad)
                p.requires_grad = False
        else:
            for i, p in enumerate(self.model.netG.parameters()):
                p.requires_grad = self.original_requires_grad_status[i]

    def _local_data(self):
        d = dict(self.data)
        lr = d['LR']
        if lr.size(0) == self.global_batch and self.global_batch > 1:
            d['LR'] = lr[self.shard[0]:self.shard[1]]
        elif lr.size(0) == 1:
            d['LR'] = lr.expand(self.shard[1] - self.shard[0], -1, -1, -1)
        return d

    def optimize(self):
        USE_MIN_LOSS_Z = not self.model_training
        self.Manage_Model_Grad_Requirements(verify_disabled=True)
        self.loss_values, per_iter_pre_tanh_Z = [], []
        if self.random_Z_inits and self.cur_iter == 0:
            self.Z_model.Randomize_Z(what_2_shuffle=self.random_Z_inits)
        z_iter = self.cur_iter
        data = self._local_data()
        while True:
            if self.max_iters > 0:
                if z_iter == (self.cur_iter + self.max_iters):
                    break
            elif len(self.loss_values) >= -self.max_iters:   # stop when the loss stops decreasing, or after 5*(-max_iters)
                if z_iter == (self.cur_iter - 5 * self.max_iters):
                    break
                if (self.loss_values[self.max_iters] - self.loss_values[-1]) / np.abs(self.loss_values[self.max_iters]) < 1e-2 * self.LR:
                    break
            self.optimizer.zero_grad()
            data['Z'] = self.Z_model()
            if USE_MIN_LOSS_Z:
                per_iter_pre_tanh_Z.append(1 * self.Z_model.PreTanhZ())
            self.model.feed_data(data, need_GT=False)
            # drop the previous iteration's output first: its graph holds the generator's saved-activation buffers, and a forward that finds
            # them busy allocates a second full set (2 x 90 GB at the configs[3] shape)
            self.output_image = Z_loss = loss = None
            self.model.fake_H = self.model.output_image = None
            self.model.test(prevent_grads_calc=False)
            self.output_image = self.model.Output_Batch(within_0_1=True)
            if self.model_training:
                self.output_image = self.HR_unpadder(self.output_image)
            if self.objective == 'hist':
                Z_loss = self.loss(self.output_image).reshape(1)
            elif 'l1' in self.objective:
                Z_loss = (self.output_image - self.desired_im).abs().mean(dim=(1, 2, 3))
            elif 'TV' in self.objective:
                Z_loss = (self.STD_PRESERVING_WEIGHT * (self.Masked_STD() - self.initial_STD) ** 2).mean(0) + \
                    (TV_Loss(self.model.fake_H, self.image_mask, clamp01=True) if not self.model_training else
                     TV_Loss(self.output_image if self.image_mask is None else self.output_image * self.image_mask))
            else:
                Z_loss = self.Masked_STD()
                if any(p in self.objective for p in ['increase', 'decrease']):
                    Z_loss = (Z_loss - self.desired_STD) ** 2
                Z_loss = Z_loss.mean(0)
            if 'max' in self.objective:
                Z_loss = -1 * Z_loss
            self.latest_Z_loss_values = [v.item() for v in Z_loss.reshape(-1)]
            # mean over the GLOBAL batch (reference :742): this shard contributes sum/B_global
            loss = Z_loss.reshape(-1).sum() / self.global_batch if Z_loss.numel() > 1 else Z_loss.mean() * (self.shard[1] - self.shard[0]) / self.global_batch
            loss.backward()
            self.loss_values.append(esr_dist.all_reduce_mean_scalar(loss.item(), self.device) * esr_dist.world_size())
            self.optimizer.step()
            z_iter += 1
        if USE_MIN_LOSS_Z and len(self.loss_values) > 0 and np.min(self.loss_values) != self.loss_values[-1]:
            min_loss_iter = int(np.argmin(self.loss_values))
            print('Minimum loss observed in %d/%d iteration, discarding subsequent iterations.' % (min_loss_iter + 1, len(self.loss_values)))
            self.Z_model.Z.data = 1 * per_iter_pre_tanh_Z[min_loss_iter]
            self.loss_values = self.loss_values[:min_loss_iter + 1]
        self.cur_iter = z_iter + 1
        Z_2_return = self.Z_model.Return_Detached_Z()
        self.Manage_Model_Grad_Requirements(verify_disabled=False)
        if self.model_training:    # one more forward with gradients enabled for the model (reference :788-795)
            data['Z'] = Z_2_return
            self.model.feed_data(data, need_GT=False)
            self.model.fake_H = self.model.netG(self.model.model_input)
        return Z_2_return

    def ReturnStatus(self):
        return self.Z_model.PreTanhZ(), self.optimizer
