"""Consistency Enforcing Module (CEM) — same public surface as the reference's codes/CEM/CEMnet.py (CEMnet, CEM_PyTorch,
Filter_Layer, CEM_downsampler, Get_CEM_Conf, Adjust_State_Dict_Keys, ...), written from scratch:

  * construction (NumPy, float64, CPU): blur kernel -> ds_kernel, inv(h^T h) by FFT inversion with a magnitude floor,
    invalidity margins measured on a ones image (reference CEMnet.py:22-49,186-206)
  * run time (MI355X): the three fixed depth-wise filters and the projection
        out = U(K x) + (g - U(K D g))  ==  g + U(K (x - D g))
    are fused HIP kernels (esr_hip/cem_ops.py -> csrc/esr_cem.hip); no padded / zero-stuffed / full-resolution
    intermediates are materialised, and the eval-mode replicate padding is folded into the generator's input packing.

The TensorFlow half of the reference file is dead code there (`tf_loaded = False`, CEMnet.py:3) and is not reproduced.
"""
import collections
import copy

import numpy as np
from scipy.signal import convolve2d as conv2

import torch
import torch.nn as nn

from CEM.imresize_CEM import imresize, calc_strides
from esr_hip import cem_ops

pytorch_loaded, tf_loaded = True, False


def Return_kernel(ds_factor, upscale_kernel=None):
    """Downscaling kernel: rot180 of the (padded) upscale kernel, float32-rounded, / sf^2 (reference CEMnet.py:364-365)."""
    up = imresize(None, [ds_factor, ds_factor], return_upscale_kernel=True, kernel=upscale_kernel)
    return np.rot90(up, 2).astype(np.float32).astype(np.float64) / (int(ds_factor) ** 2)


def Aliased_Down_Sampling(array, factor):
    pre_stride, _ = calc_strides(array, 1 / factor, align_center=True)
    return array[pre_stride[0]::factor, pre_stride[1]::factor, ...]


def Pad_Image(image, margin_size):
    pad = ((margin_size, margin_size), (margin_size, margin_size)) + (((0, 0),) if image.ndim == 3 else ())
    return np.pad(image, pad_width=pad, mode='edge')


def Unpad_Image(image, margin_size):
    return image[margin_size:-margin_size, margin_size:-margin_size, :]


def Get_CEM_Conf(sf):
    class conf:
        scale_factor = sf
        avoid_skip_connections = False
        generate_HR_image = False
        pseudo_CEM_supplement = False
        desired_inv_hTh_energy_portion = 1 - 1e-6
        filter_pertubation_limit = 0.999
        sigmoid_range_limit = False
        lower_magnitude_bound = 0.01   # lower bound on the hTh filter's magnitude in the Fourier domain
    return conf


class CEMnet:
    NFFT_add = 36

    def __init__(self, conf, upscale_kernel=None):
        self.conf = conf
        self.ds_factor = np.array(conf.scale_factor, dtype=np.int32)
        assert np.round(self.ds_factor) == self.ds_factor, 'Currently only supporting integer scale factors'
        assert upscale_kernel is None or isinstance(upscale_kernel, (str, np.ndarray)), \
            'To support given kernels, change the Return_Invalid_Margin_Size_in_LR function and make sure everything else works'
        self.ds_kernel = Return_kernel(self.ds_factor, upscale_kernel=upscale_kernel)
        self.ds_kernel_invalidity_half_size_LR = self.Return_Invalid_Margin_Size_in_LR('ds_kernel', self.conf.filter_pertubation_limit)
        self.compute_inv_hTh()
        self.invalidity_margins_LR = 2 * self.ds_kernel_invalidity_half_size_LR + self.inv_hTh_invalidity_half_size
        self.invalidity_margins_HR = self.ds_factor * self.invalidity_margins_LR

    # ---------------------------------------------------------------- construction helpers
    def Return_Invalid_Margin_Size_in_LR(self, filter, max_allowed_perturbation):
        """How many LR pixels from the border a filter's response to a constant image deviates by more than the allowed
        perturbation (measured on a 100x100 LR ones image, zero padding) — reference CEMnet.py:35-49."""
        T = 100
        assert filter in ['ds_kernel', 'inv_hTh']
        sf = int(self.ds_factor)
        if filter == 'ds_kernel':
            resp = imresize(np.ones([sf * T, sf * T]), [1 / sf], use_zero_padding=True)
        else:
            resp = conv2(np.ones([T, T]), self.inv_hTh, mode='same')
        resp = resp / resp[T // 2, T // 2]
        resp[resp <= 0] = max_allowed_perturbation / 2
        invalid = np.exp(-np.abs(np.log(resp))) < max_allowed_perturbation
        deepest = [np.argwhere(invalid[:T // 2, T // 2])[-1][0] + 1, np.argwhere(invalid[T // 2, :T // 2])[-1][0] + 1]
        return np.max(deepest)

    def compute_inv_hTh(self):
        """inv_hTh = F^-1[ 1 / max-floored F[ (h * h~)|_(down sf) ] ], centred on its peak and cropped to the configured energy
        portion (reference CEMnet.py:186-206)."""
        sf = int(self.ds_factor)
        hTh = Aliased_Down_Sampling(conv2(self.ds_kernel, np.rot90(self.ds_kernel, 2)) * sf ** 2, sf)
        p = self.NFFT_add // 2
        spectrum = np.fft.fft2(np.pad(hTh, ((p, p), (p, p)), mode='constant'))
        spectrum = spectrum * np.maximum(1, self.conf.lower_magnitude_bound / np.abs(spectrum))
        inv = np.real(np.fft.ifft2(1 / spectrum))
        n = inv.shape[0]
        peak_r, peak_c = np.argmax(inv) // n, np.mod(np.argmax(inv), n)
        if not np.all(np.equal(np.ceil(np.array(inv.shape) / 2), np.array([peak_r, peak_c]) - 1)):
            half = np.min([n - peak_r - 1, n - peak_c - 1, peak_r, peak_c])
            inv = inv[peak_r - half:peak_r + half + 1, peak_c - half:peak_c + half + 1]
        self.inv_hTh = inv
        self.inv_hTh_invalidity_half_size = self.Return_Invalid_Margin_Size_in_LR('inv_hTh', self.conf.filter_pertubation_limit)
        drop = self.inv_hTh.shape[0] // 2 - self.Return_Invalid_Margin_Size_in_LR('inv_hTh', self.conf.desired_inv_hTh_energy_portion)
        if drop > 0:
            self.inv_hTh = self.inv_hTh[drop:-drop, drop:-drop]

    # ---------------------------------------------------------------- NumPy batch / image helpers
    def Pad_LR_Batch(self, batch, num_recursion=1):
        m = int(self.invalidity_margins_LR)
        for _ in range(num_recursion):
            batch = 1.0 * np.pad(batch, pad_width=((0, 0), (m, m), (m, m), (0, 0)), mode='edge')
        return batch

    def Unpad_HR_Batch(self, batch, num_recursion=1):
        m = int((self.ds_factor ** num_recursion) * self.invalidity_margins_LR * num_recursion)
        return batch[:, m:-m, m:-m, :]

    def DT_Satisfying_Upscale(self, LR_image):
        """U(K x): the consistent ("orthogonal to the null space") upscale of an HWC LR image (reference CEMnet.py:60-64)."""
        margin = int(2 * self.inv_hTh_invalidity_half_size + self.ds_kernel_invalidity_half_size_LR)
        LR_image = Pad_Image(LR_image, margin)
        filtered = np.stack([conv2(LR_image[:, :, c], self.inv_hTh, mode='same') for c in range(LR_image.shape[-1])], -1)
        HR_image = imresize(filtered, scale_factor=[int(self.ds_factor)])
        if HR_image.ndim < 3:
            HR_image = HR_image[:, :, None]
        return Unpad_Image(HR_image, int(self.ds_factor) * margin)

    def Project_2_ortho_2_NS(self, HR_input):
        sf = int(self.ds_factor)
        down = imresize(HR_input, scale_factor=[1 / sf])
        if down.ndim < HR_input.ndim:
            down = np.reshape(down, list(np.array(HR_input.shape[:2]) // sf) + ([HR_input.shape[2]] if HR_input.ndim > 2 else []))
        return self.DT_Satisfying_Upscale(down)

    def Enforce_DT_on_Image_Pair(self, LR_source, HR_input):
        same = [LR_source.shape[i] == HR_input.shape[i] for i in range(LR_source.ndim)]
        scaled = [int(self.ds_factor) * LR_source.shape[i] == HR_input.shape[i] for i in range(LR_source.ndim)]
        assert np.all(np.logical_or(same, scaled))
        if len(same) == 2:
            LR_source, HR_input = np.expand_dims(LR_source, -1), np.expand_dims(HR_input, -1)
        LR_source = self.DT_Satisfying_Upscale(LR_source) if np.any(scaled) else self.Project_2_ortho_2_NS(LR_source)
        return HR_input - self.Project_2_ortho_2_NS(HR_input) + LR_source

    # ---------------------------------------------------------------- PyTorch wrapping
    def WrapArchitecture_PyTorch(self, generated_image=None, training_patch_size=None, only_padders=False, grayscale=False):
        mL = int(self.invalidity_margins_LR)
        mH = int(self.ds_factor) * mL
        self.LR_padder = torch.nn.ReplicationPad2d((mL, mL, mL, mL))
        self.HR_padder = torch.nn.ReplicationPad2d((mH, mH, mH, mH))
        self.HR_unpadder = lambda x: x[:, :, mH:-mH, mH:-mH]
        self.LR_unpadder = lambda x: x[:, :, mL:-mL, mL:-mL]   # debugging tool
        self.loss_mask = None
        if training_patch_size is not None:
            mask = np.zeros([1, 1, training_patch_size, training_patch_size])
            m = int(self.invalidity_margins_HR)
            mask[:, :, m:-m, m:-m] = 1
            assert np.mean(mask) > 0, 'Loss mask completely nullifies image.'
            print('Using only only %.3f of patch area for learning. The rest is considered to have boundary effects' % (np.mean(mask)))
            self.loss_mask = torch.from_numpy(mask).float()
            if torch.cuda.is_available():
                self.loss_mask = self.loss_mask.cuda()
        if only_padders:
            return
        returnable = CEM_PyTorch(self, generated_image, grayscale=grayscale)
        self.OP_names = [m[0] for m in returnable.named_modules() if 'Filter_OP' in m[0]]
        return returnable

    def Mask_Invalid_Regions_PyTorch(self, im1, im2):
        assert self.loss_mask is not None, 'Mask not defined, probably didn''t pass patch size'
        mask = self.loss_mask.to(im1.device)
        return mask * im1, mask * im2


class Filter_Layer(nn.Module):
    """Depth-wise fixed filter: post(Filter_OP(pre(x))) with Filter_OP a frozen nn.Conv2d(groups=C, bias=False) parameter
    container (key `Filter_OP.weight`, flag `filter_layer=True` that init_weights skips — reference CEMnet.py:243-252,
    networks.py:29-31).  The filtering itself runs in csrc/esr_cem.hip; `role` selects the fused form the CEM uses:
        'lr_filter' : replicate-pad floor(k/2) + correlate                       (Conv_LR_with_Inv_hTh_OP)
        'upscale'   : zero-stuff at (pre,pre) + replicate-pad + correlate        (Upscale_OP)
        'downscale' : replicate-pad + correlate, evaluated at [pre::sf, pre::sf] (DownscaleOP)
        None        : generic — pre_filter_func, 'valid' correlation, post_filter_func."""

    def __init__(self, filter, pre_filter_func, post_filter_func=None, num_channels=3, role=None, sf=1, pre_stride=0):
        super(Filter_Layer, self).__init__()
        filter = np.ascontiguousarray(filter)
        self.Filter_OP = nn.Conv2d(in_channels=num_channels, out_channels=num_channels, kernel_size=filter.shape, bias=False, groups=num_channels)
        self.Filter_OP.weight = nn.Parameter(data=torch.from_numpy(np.tile(filter[None, None], reps=[num_channels, 1, 1, 1])).float(),
                                             requires_grad=False)
        self.Filter_OP.filter_layer = True
        self.pre_filter_func = pre_filter_func
        self.post_filter_func = (lambda x: x) if post_filter_func is None else post_filter_func
        self.role, self.sf, self.pre_stride = role, int(sf), int(pre_stride)

    def taps(self):
        return self.Filter_OP.weight[0, 0]

    def forward(self, x):
        if self.role == 'lr_filter':
            return cem_ops.lr_filter(x, self.taps())
        if self.role == 'upscale':
            return cem_ops.upscale(x, self.taps(), self.sf, self.pre_stride)
        if self.role == 'downscale':
            return cem_ops.downscale(x, self.taps(), self.sf, self.pre_stride)
        x = self.pre_filter_func(x)
        p = self.Filter_OP.weight.shape[-1] // 2
        y = cem_ops.lr_filter(x, self.taps())               # 'same' with replicate pad; its interior is the 'valid' correlation
        return self.post_filter_func(y[:, :, p:y.shape[2] - p, p:y.shape[3] - p])


class CEM_PyTorch(nn.Module):
    def __init__(self, CEMnet, generated_image, grayscale=False):
        super(CEM_PyTorch, self).__init__()
        num_channels = 1 if grayscale else 3
        self.ds_factor = CEMnet.ds_factor
        self.conf = CEMnet.conf
        self.using_SR_model = generated_image is not None
        if self.using_SR_model:
            self.generated_image_model = generated_image
        sf = int(CEMnet.ds_factor)
        pre_stride, post_stride = calc_strides(None, sf)
        assert CEMnet.ds_kernel.shape[0] == CEMnet.ds_kernel.shape[1] and pre_stride[0] == pre_stride[1]
        self.pre_stride = int(pre_stride[0])
        self.Conv_LR_with_Inv_hTh_OP = Filter_Layer(CEMnet.inv_hTh, pre_filter_func=None, num_channels=num_channels, role='lr_filter')
        self.Upscale_OP = Filter_Layer(CEMnet.ds_kernel * sf ** 2, pre_filter_func=None, num_channels=num_channels, role='upscale', sf=sf,
                                       pre_stride=self.pre_stride)
        self.DownscaleOP = Filter_Layer(np.rot90(CEMnet.ds_kernel, 2), pre_filter_func=None, num_channels=num_channels, role='downscale', sf=sf,
                                        pre_stride=self.pre_stride)
        self.LR_padder = CEMnet.LR_padder
        self.HR_padder = CEMnet.HR_padder
        self.HR_unpadder = CEMnet.HR_unpadder
        self.LR_unpadder = CEMnet.LR_unpadder   # debugging tool
        self.margins_LR = int(CEMnet.invalidity_margins_LR)
        self.pre_pad = False   # a flag rather than a forward() argument, as in the reference (set by .train()/.eval())
        self.return_2_components = 'decomposed_output' in self.conf.__dict__ and self.conf.decomposed_output

    @property
    def module(self):
        """Callers written against nn.DataParallel reach the operators through `.module` (GUI.py:1687,2516); one process per
        GPU needs no wrapper, so the module is its own `.module`."""
        return self

    def forward(self, x):
        return_2_components = self.return_2_components and not self.pre_pad
        sf = int(self.ds_factor)
        mL = self.margins_LR if self.pre_pad else 0
        if self.using_SR_model:
            G = self.generated_image_model
            if mL and getattr(G, 'engine', None) is not None:
                generated_image = G(x, pad=mL)           # padding folded into the generator's input packing
            elif mL:
                lat = getattr(G, 'num_latent_channels', 0)
                if x.size(1) != 3 and x.size(1) - 3 != lat:    # HR-domain Z packed by view: pad it at HR (reference CEMnet.py:288-293)
                    z, im = torch.split(x, [x.size(1) - 3, 3], dim=1)
                    z = z.reshape(z.size(0), -1, G.upscale * z.size(2), G.upscale * z.size(3))
                    im = self.LR_padder(im)
                    z = self.HR_padder(z).reshape(z.size(0), z.size(1) * G.upscale ** 2, im.size(2), im.size(3))
                    generated_image = G(torch.cat([z, im], 1))
                else:
                    generated_image = G(self.LR_padder(x))
            else:
                generated_image = G(x)
            lr = x[:, -3:, :, :]
        else:
            lr, generated_image = x[0], x[1]
            if mL:
                generated_image = self.HR_padder(generated_image)
            lr = lr[:, -3:, :, :]
        assert np.all(np.mod(generated_image.size()[2:], sf) == 0)
        rng = None
        if self.conf.sigmoid_range_limit:
            rng = float(self.conf.input_range[1] - self.conf.input_range[0])
        return cem_ops.project(lr, generated_image, self.DownscaleOP.taps(), self.Conv_LR_with_Inv_hTh_OP.taps(), self.Upscale_OP.taps(),
                               sf, self.pre_stride, lr_pad=mL, crop=sf * mL, sigmoid_range=rng, decomposed=return_2_components)

    def train(self, mode=True):
        super(CEM_PyTorch, self).train(mode=mode)
        self.pre_pad = not mode
        return self

    def Image_2_Sigmoid_Range_Converter(self, images, opposite_direction=False):
        lo, hi = self.conf.input_range[0], self.conf.input_range[1]
        if opposite_direction:
            return images * (hi - lo) + lo
        return (torch.clamp(images, min=lo, max=hi) - lo) / (hi - lo)

    def Inverse_Sigmoid(self, images):
        s = self.Image_2_Sigmoid_Range_Converter(images)
        return torch.log(s / (1. - s))


def Adjust_State_Dict_Keys(loaded_state_dict, current_state_dict):
    """Prefix a bare-generator checkpoint with 'generated_image_model.' when the current model is CEM-wrapped, carrying the
    current filter taps along (reference CEMnet.py:403-412)."""
    wrapped = all(('generated_image_model' in k or 'Filter' in k) for k in current_state_dict.keys())
    if wrapped and not any('generated_image_model' in k for k in loaded_state_dict.keys()):
        out = collections.OrderedDict(('generated_image_model.' + k, v) for k, v in loaded_state_dict.items())
        for k in current_state_dict.keys():
            if 'Filter' in k:
                out[k] = current_state_dict[k]
        return out
    return loaded_state_dict


class CEM_downsampler(nn.Module):
    """Downsample [N,C,H,W] images with the CEM's kernel, replicate-padding the HR input to avoid border artefacts
    (reference CEMnet.py:414-428).  One fused kernel: the padding is index clamping."""

    def __init__(self, ds_factor, grayscale=False, differentiable=False):
        super(CEM_downsampler, self).__init__()
        cem = CEMnet(Get_CEM_Conf(ds_factor))
        cem.invalidity_margins_LR = 1 * cem.ds_kernel_invalidity_half_size_LR
        self.CEM = cem.WrapArchitecture_PyTorch(grayscale=grayscale)
        if not differentiable:
            self.CEM.eval()

    def forward(self, input):
        # pad(HR) -> DownscaleOP -> unpad(LR) == DownscaleOP with clamped reads: the padded ring only feeds cropped outputs
        return self.CEM.DownscaleOP(input)
