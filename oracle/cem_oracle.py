"""TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's Consistency Enforcing
Module: tap construction (NumPy/SciPy, float64) and the three fixed depth-wise filters +
projection (torch CPU, fp32).  Pinned against the reference itself through
tests/golden/{cem_taps,cem_filter_ops,cem_forward}.npz (made by oracle/gen_golden.py);
the bicubic taps rest on oracle/cv2_cubic.py, whose header states what is unpinned.

Every function cites the reference lines it follows (paths under /root/reference/codes).
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import convolve2d
from scipy.signal.windows import gaussian as _gauss_win
from scipy.stats import norm as _norm

from oracle import cv2_cubic


# ----------------------------------------------------------------------------- taps
def calc_strides(sf):
    """CEM/imresize_CEM.py:89-102 (align_center=False branch)."""
    post = int(np.floor(sf / 2))
    pre = int(sf - post - 1)
    return pre, post


def cubic_upscale_kernel(sf):
    """CEM/imresize_CEM.py:104-115: resize an 11x11 delta by sf, crop to the non-zero support."""
    n = 11
    delta = np.zeros((n, n))
    c = int(np.ceil(n / 2)) - 1
    delta[c, c] = 1
    up = cv2_cubic.resize(delta, (sf * n, sf * n), interpolation=cv2_cubic.INTER_CUBIC)
    support = np.nonzero(up[sf * int(np.ceil(n / 2)) - 1, :])[0]
    return up[support[0]:support[-1] + 1, support[0]:support[-1] + 1]


def gaussian_2d(sigma):
    """CEM/imresize_CEM.py:117-124."""
    size = int(1 + 2 * np.ceil(-1 * _norm.ppf(0.005, scale=sigma)))
    g = _gauss_win(size, sigma).reshape(1, size) * _gauss_win(size, sigma).reshape(size, 1)
    return g / np.sum(g)


def _energy_distribution(f):
    """CEM/imresize_CEM.py:177-179."""
    e = [np.sqrt(np.sum(f ** 2))] + [np.sqrt(np.sum(f[k:-k, k:-k] ** 2)) for k in range(1, int(np.ceil(f.shape[0] / 2)))]
    return np.array(e) / e[0]


def center_mass(kernel, sf):
    """CEM/imresize_CEM.py:129-175: pad a user kernel so that its centre of mass is central,
    crop to 99 % energy keeping (size-1+[sf even]) divisible by sf, renormalise."""
    r = lambda v: int(np.round(v))
    k = kernel.shape[0]
    assert kernel.shape[0] == kernel.shape[1]
    xg, yg = np.meshgrid(np.arange(k), np.arange(k))
    xg, yg = convolve2d(xg, kernel, mode='valid') + 1, convolve2d(yg, kernel, mode='valid') + 1
    x_pad, y_pad = float(2 * (k / 2 - xg.item())), float(2 * (k / 2 - yg.item()))
    diff = np.round(abs(y_pad)) - np.round(abs(x_pad))
    pre_x, post_x = max(0, -x_pad), max(0, x_pad)
    pre_y, post_y = max(0, -y_pad), max(0, y_pad)

    def wisely(pre, post, d):
        off = np.round(post) - post - (np.round(pre) - pre)
        pre, post = r(pre), r(post)
        if off > 0:
            post += int(np.ceil(d / 2)); pre += int(np.floor(d / 2))
        else:
            pre += int(np.ceil(d / 2)); post += int(np.floor(d / 2))
        return pre, post
    if diff > 0:
        pre_y, post_y = r(pre_y), r(post_y)
        pre_x, post_x = wisely(pre_x, post_x, diff)
    elif diff < 0:
        pre_x, post_x = r(pre_x), r(post_x)
        pre_y, post_y = wisely(pre_y, post_y, -diff)
    kernel = np.pad(kernel, ((r(pre_y), r(post_y)), (r(pre_x), r(post_x))), mode='constant')
    assert kernel.shape[0] == kernel.shape[1]
    m = np.argwhere(_energy_distribution(kernel) < 0.99)[0][0] * np.ones([2]).astype(np.int32)
    j = 0
    while np.mod(kernel.shape[0] - np.sum(m) - 1 + np.mod(sf + 1, 2), sf) != 0:
        m[j] -= 1
        j = (j + 1) % 2
    kernel = kernel[m[0]:-m[1], m[0]:-m[1]]
    return kernel / np.sum(kernel)


def upscale_kernel(sf, kernel=None):
    """CEM/imresize_CEM.py:8-48 with return_upscale_kernel=True (the padded anti-aliasing kernel)."""
    pre, post = calc_strides(sf)
    post_pad, pre_pad = max(0, pre - post), max(0, post - pre)
    if isinstance(kernel, np.ndarray):
        assert abs(1 - np.sum(kernel)) < np.finfo(np.float32).eps, 'Supplied non-default kernel does not sum to 1'
        k = center_mass(np.rot90(kernel, 2), sf) * sf ** 2
        assert k.shape[0] == k.shape[1]
        assert (k.shape[0] + post_pad + pre_pad - 1) % sf == 0
    else:
        assert kernel is None or 'cubic' in kernel
        k = cubic_upscale_kernel(sf)
        if kernel is not None and 'blurry_cubic' in kernel:
            k = convolve2d(k, gaussian_2d(float(kernel[len('blurry_cubic_'):])))
    return np.pad(k, ((pre_pad, post_pad), (pre_pad, post_pad)), mode='constant')


def ds_kernel(sf, kernel=None):
    """CEM/CEMnet.py:364-365 (Return_kernel): float32 rounding of the taps, then / sf**2 in float64."""
    return np.rot90(upscale_kernel(sf, kernel), 2).astype(np.float32).astype(np.float64) / (sf ** 2)


def aliased_down_sampling(a, sf):
    """CEM/CEMnet.py:326-332 + imresize_CEM.py:92-97 (align_center=True, factor<1)."""
    half = np.ceil(np.array(a.shape[:2]) / 2)
    pre = np.mod(half, sf)
    pre[pre == 0] = sf
    pre = (pre - 1).astype(np.int32)
    return a[pre[0]::sf, pre[1]::sf]


def imresize_np(im, sf_up=None, sf_down=None, kernel_up=None, use_zero_padding=False):
    """CEM/imresize_CEM.py:49-87 for one direction; `kernel_up` = upscale_kernel(sf, ...)."""
    sf = sf_up or sf_down
    pre, post = calc_strides(sf)
    ak = kernel_up if sf_up else np.rot90(kernel_up * (1.0 / sf) ** 2, 2)
    pad = np.floor(np.array(ak.shape) / 2).astype(np.int32)
    squeeze = im.ndim < 3
    if squeeze:
        im = im[:, :, None]
    outs = []
    for c in range(im.shape[2]):
        ch = im[:, :, c]
        if sf_up:
            z = np.zeros((ch.shape[0] * sf, ch.shape[1] * sf))
            z[pre::sf, pre::sf] = ch
            ch = z
        if use_zero_padding:
            o = convolve2d(ch, ak, 'same')
        else:
            o = convolve2d(np.pad(ch, ((pad[0], pad[0]), (pad[1], pad[1])), mode='edge'), ak, 'valid')
        if sf_down:
            o = o[pre::sf, pre::sf]
        outs.append(o)
    o = np.stack(outs, -1)
    return o[:, :, 0] if squeeze else o


class CEMTaps:
    """Everything CEMnet.__init__ derives (CEM/CEMnet.py:22-49,186-206)."""
    NFFT_add = 36

    def __init__(self, sf, kernel=None, lower_magnitude_bound=0.01, energy_portion=1 - 1e-6, perturbation_limit=0.999):
        self.sf = int(sf)
        self.up_kernel = upscale_kernel(self.sf, kernel)
        self.ds_kernel = ds_kernel(self.sf, kernel)
        self.pre, self.post = calc_strides(self.sf)
        self.ds_half = self._margin('ds_kernel', perturbation_limit)
        # compute_inv_hTh (CEMnet.py:186-206)
        hTh = convolve2d(self.ds_kernel, np.rot90(self.ds_kernel, 2)) * self.sf ** 2
        hTh = aliased_down_sampling(hTh, self.sf)
        p = self.NFFT_add // 2
        fft = np.fft.fft2(np.pad(hTh, ((p, p), (p, p)), mode='constant'))
        fft = fft * np.maximum(1, lower_magnitude_bound / np.abs(fft))
        inv = np.real(np.fft.ifft2(1 / fft))
        mr, mc = np.argmax(inv) // inv.shape[0], np.mod(np.argmax(inv), inv.shape[0])
        if not np.all(np.equal(np.ceil(np.array(inv.shape) / 2), np.array([mr, mc]) - 1)):
            h = np.min([inv.shape[0] - mr - 1, inv.shape[0] - mc - 1, mr, mc])
            inv = inv[mr - h:mr + h + 1, mc - h:mc + h + 1]
        self.inv_hTh = inv
        self.inv_half = self._margin('inv_hTh', perturbation_limit)
        drop = self.inv_hTh.shape[0] // 2 - self._margin('inv_hTh', energy_portion)
        if drop > 0:
            self.inv_hTh = self.inv_hTh[drop:-drop, drop:-drop]
        self.margins_LR = 2 * self.ds_half + self.inv_half
        self.margins_HR = self.sf * self.margins_LR

    def _margin(self, which, limit):
        """CEMnet.py:35-49 (Return_Invalid_Margin_Size_in_LR)."""
        T = 100
        if which == 'ds_kernel':
            o = imresize_np(np.ones([self.sf * T, self.sf * T]), sf_down=self.sf, kernel_up=self.up_kernel, use_zero_padding=True)
        else:
            o = convolve2d(np.ones([T, T]), self.inv_hTh, mode='same')
        o = o / o[T // 2, T // 2]
        o[o <= 0] = limit / 2
        mask = np.exp(-np.abs(np.log(o))) < limit
        m = [np.argwhere(mask[:T // 2, T // 2])[-1][0] + 1, np.argwhere(mask[T // 2, :T // 2])[-1][0] + 1]
        return int(np.max(m))


# ----------------------------------------------------------------------------- fixed filters (torch CPU fp32)
def _dw(x, k):
    """Depth-wise cross-correlation with one shared 2-D filter (Filter_Layer, CEMnet.py:243-252)."""
    C = x.shape[1]
    w = torch.from_numpy(np.ascontiguousarray(k)).to(torch.float32)[None, None].repeat(C, 1, 1, 1)
    return F.conv2d(x, w, groups=C)


def conv_lr_with_inv_hTh(x, taps):
    """CEMnet.py:261-263: replicate-pad floor(k/2), filter with inv_hTh."""
    p = taps.inv_hTh.shape[0] // 2
    return _dw(F.pad(x, (p, p, p, p), mode='replicate'), taps.inv_hTh)


def upscale_op(x, taps):
    """CEMnet.py:264-271: zero-stuff at sub-index (pre,pre), replicate-pad floor(k/2), filter with ds_kernel*sf^2."""
    sf, pre = taps.sf, taps.pre
    B, C, h, w = x.shape
    z = torch.zeros(B, C, h * sf, w * sf, dtype=x.dtype)
    z[:, :, pre::sf, pre::sf] = x
    p = taps.ds_kernel.shape[0] // 2
    return _dw(F.pad(z, (p, p, p, p), mode='replicate'), taps.ds_kernel * sf ** 2)


def downscale_op(y, taps):
    """CEMnet.py:272-275: replicate-pad, filter with rot90(ds_kernel,2) at every HR pixel, keep [pre::sf, pre::sf]."""
    p = taps.ds_kernel.shape[0] // 2
    f = _dw(F.pad(y, (p, p, p, p), mode='replicate'), np.rot90(taps.ds_kernel, 2))
    return f[:, :, taps.pre::taps.sf, taps.pre::taps.sf]


def cem_project(x, g, taps, pre_pad=False, sigmoid_range=None, decomposed=False):
    """CEM_PyTorch.forward after the generator call (CEMnet.py:297-311); x = LR (last 3 ch), g = generated HR."""
    mL, mH = taps.margins_LR, taps.margins_HR
    if pre_pad:
        x = F.pad(x, (mL,) * 4, mode='replicate')
        g = F.pad(g, (mH,) * 4, mode='replicate')
    return cem_combine(x, g, taps, crop=pre_pad, sigmoid_range=sigmoid_range, decomposed=decomposed)


def cem_combine(x_padded, g, taps, crop, sigmoid_range=None, decomposed=False):
    """CEMnet.py:303-311 given an already padded LR and the generator output on it."""
    x = x_padded[:, -3:] if x_padded.shape[1] > 3 else x_padded
    assert g.shape[2] % taps.sf == 0 and g.shape[3] % taps.sf == 0
    ortho_x = upscale_op(conv_lr_with_inv_hTh(x, taps), taps)
    ortho_g = upscale_op(conv_lr_with_inv_hTh(downscale_op(g, taps), taps), taps)
    ns = g - ortho_g
    if sigmoid_range is not None:
        ns = torch.tanh(ns) * (sigmoid_range[1] - sigmoid_range[0])
    mH = taps.margins_HR
    if decomposed and not crop:
        return [ortho_x, ns]
    out = ortho_x + ns
    return out[:, :, mH:-mH, mH:-mH] if crop else out


def cem_downsampler(y, sf, grayscale=False):
    """CEM_downsampler (CEMnet.py:414-428): margins = ds-kernel half size only; pad HR, downscale, unpad LR."""
    taps = CEMTaps(sf)
    m = taps.ds_half
    yp = F.pad(y, (sf * m,) * 4, mode='replicate')
    d = downscale_op(yp, taps)
    return d[:, :, m:-m, m:-m]
