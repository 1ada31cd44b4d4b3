"""Integer-factor image resizing with an explicit anti-aliasing kernel — the host-side (NumPy, construction time) part of
the Consistency Enforcing Module.  Same public surface and index conventions as the reference's
codes/CEM/imresize_CEM.py (imresize, calc_strides, Cubic_Kernel, Gaussian_2D, Center_Mass, process-global per-scale kernel
cache `imresize.kernels`), written from scratch and without OpenCV: the bicubic taps are evaluated in closed form
(Keys kernel a=-0.75 at the sample phases OpenCV's INTER_CUBIC resize uses, float32 coefficient arithmetic like OpenCV).

Conventions (must stay bit-exact):
  * factor sf: post = floor(sf/2), pre = sf - post - 1; LR sample (i,j) <-> HR pixel (sf*i+pre, sf*j+pre)
  * even sf: the (even-sized) upscale kernel is zero padded by one row/col so that its centre is a pixel
  * downscale = correlate-with-rot180 == convolve, edge ('replicate') padding by floor(k/2), keep [pre::sf, pre::sf]
"""
import numpy as np
from scipy.signal import convolve2d
from scipy.signal.windows import gaussian as _gaussian_window
from scipy.stats import norm as _norm


def calc_strides(array, factor, align_center=False):
    """(pre, post) zero-stuffing strides per axis (reference imresize_CEM.py:89-102)."""
    integer_factor = int(np.maximum(factor, 1 / factor))
    if align_center:
        half = np.ceil(np.array(array.shape[:2]) / 2 * (factor if factor > 1 else 1))
        pre = np.mod(half, integer_factor)
        pre[pre == 0] = integer_factor
        pre = (pre - 1).astype(np.int32)
        post = (integer_factor - pre - 1).astype(np.int32)
    else:
        post = (np.floor(integer_factor / 2) * np.ones([2])).astype(np.int32)
        pre = (integer_factor - post - 1).astype(np.int32)
    return pre, post


def _keys_coefficients_f32(t):
    """The four cubic-convolution weights for fractional offset t (taps at -1, 0, +1, +2), a = -0.75, float32 Horner
    evaluation in the operation order of OpenCV's interpolateCubic."""
    f = np.float32
    a, x, one = f(-0.75), f(t), f(1.0)
    xp = f(x + one)
    xm = f(one - x)
    c0 = f(f(f(f(f(a * xp) - f(f(5) * a)) * xp) + f(f(8) * a)) * xp) - f(f(4) * a)
    c1 = f(f(f(f(f(f(a + f(2)) * x) - f(a + f(3))) * x) * x) + one)
    c2 = f(f(f(f(f(f(a + f(2)) * xm) - f(a + f(3))) * xm) * xm) + one)
    c3 = f(f(f(one - f(c0)) - c1) - c2)
    return [f(c0), c1, c2, c3]


def Cubic_Kernel(sf):
    """2-D bicubic *upscaling* kernel for integer factor sf (sums to sf^2): the response of a x-sf bicubic resize to a unit
    impulse, i.e. what the reference obtains from cv2.resize(delta, INTER_CUBIC) (imresize_CEM.py:104-110)."""
    sf = int(sf)
    n, c = 11, 5                      # impulse at the centre of an 11-sample line
    u = np.zeros(n * sf, dtype=np.float64)
    for d in range(n * sf):
        pos = np.float32((d + 0.5) / sf - 0.5)          # source coordinate of destination sample d
        base = int(np.floor(pos))
        k = c - (base - 1)                              # which of the 4 taps looks at the impulse
        if 0 <= k <= 3:
            u[d] = float(_keys_coefficients_f32(np.float32(pos - np.float32(base)))[k])
    nz = np.nonzero(u)[0]
    u = u[nz[0]:nz[-1] + 1]
    return np.outer(u, u)


def Gaussian_2D(sigma, size=None):
    """Normalised isotropic Gaussian holding ~99 % of its 1-D energy (reference imresize_CEM.py:117-124)."""
    if size is None:
        size = int(1 + 2 * np.ceil(-1 * _norm.ppf(0.005, scale=sigma)))
    else:
        assert (size + 1) / 2 == np.round((size + 1) / 2), 'Size must be odd integer'
    g = np.outer(_gaussian_window(size, sigma), _gaussian_window(size, sigma))
    return g / np.sum(g)


def Return_Filter_Energy_Distribution(filter):
    n = int(np.ceil(filter.shape[0] / 2))
    e = np.array([np.sqrt(np.sum(filter[k:filter.shape[0] - k, k:filter.shape[1] - k] ** 2)) for k in range(n)])
    return e / e[0]


def Round_2_Int(num):
    return int(np.round(num))


def Center_Mass(kernel, ds_factor):
    """Zero-pad a user supplied square kernel so that its centre of mass sits on the central pixel, then trim it to 99 % of
    its energy while keeping (size - 1 + [sf even]) a multiple of sf; renormalise (reference imresize_CEM.py:129-175)."""
    assert kernel.shape[0] == kernel.shape[1], 'Currently supporting only square kernels'
    k = kernel.shape[0]
    idx = np.arange(k, dtype=np.float64)
    # 1-based centre of mass along x (columns) and y (rows)
    cx = float(np.sum(kernel * idx[None, :])) + 1
    cy = float(np.sum(kernel * idx[:, None])) + 1
    need = {'x': 2 * (k / 2 - cx), 'y': 2 * (k / 2 - cy)}          # >0: pad after, <0: pad before
    pre = {ax: max(0.0, -v) for ax, v in need.items()}
    post = {ax: max(0.0, v) for ax, v in need.items()}
    diff = np.round(abs(need['y'])) - np.round(abs(need['x']))      # extra padding needed to stay square

    def spread(p0, p1, extra):
        lean_right = (np.round(p1) - p1) - (np.round(p0) - p0)
        p0, p1 = Round_2_Int(p0), Round_2_Int(p1)
        big, small = int(np.ceil(extra / 2)), int(np.floor(extra / 2))
        return (p0 + small, p1 + big) if lean_right > 0 else (p0 + big, p1 + small)
    if diff > 0:
        pre['x'], post['x'] = spread(pre['x'], post['x'], diff)
    elif diff < 0:
        pre['y'], post['y'] = spread(pre['y'], post['y'], -diff)
    kernel = np.pad(kernel, ((Round_2_Int(pre['y']), Round_2_Int(post['y'])), (Round_2_Int(pre['x']), Round_2_Int(post['x']))), mode='constant')
    assert kernel.shape[0] == kernel.shape[1], 'I caused the kernel to stop being a square...'
    trim = np.argwhere(Return_Filter_Energy_Distribution(kernel) < 0.99)[0][0] * np.ones([2]).astype(np.int32)
    side = 0
    while np.mod(kernel.shape[0] - np.sum(trim) - 1 + np.mod(ds_factor + 1, 2), ds_factor) != 0:
        trim[side] -= 1
        side = (side + 1) % 2
    kernel = kernel[trim[0]:kernel.shape[0] - trim[1], trim[0]:kernel.shape[1] - trim[1]]
    return kernel / np.sum(kernel)


def imresize(im, scale_factor=None, output_shape=None, kernel=None, align_center=False, return_upscale_kernel=False,
             use_zero_padding=False, antialiasing=True, kernel_shift_flag=False):
    """Resize by an integer factor (up: zero-stuff + filter, down: filter + stride).  `kernel`: None / 'cubic' /
    'blurry_cubic_<sigma>' / 'reset_2_default' / a 2-D *downscaling* ndarray that sums to 1.  The upscale kernel in use for a
    scale factor is cached process-wide in `imresize.kernels[str(sf)]` exactly like the reference (imresize_CEM.py:10,23-43):
    a custom kernel replaces the cached one until 'reset_2_default'."""
    assert kernel is None or isinstance(kernel, np.ndarray) or any(word in kernel for word in ['cubic', 'blurry_cubic', 'reset_2_default'])
    imresize.kernels = getattr(imresize, 'kernels', {})
    if scale_factor is None:
        scale_factor = [output_shape[0] / im.shape[0]]
    elif not isinstance(scale_factor, list):
        scale_factor = [scale_factor]
    assert np.round(scale_factor[0]) == scale_factor[0] or np.round(1 / scale_factor[0]) == 1 / scale_factor[0], \
        'Only supporting integer downsampling or upsampling rates'
    assert len(scale_factor) == 1 or scale_factor[0] == scale_factor[1]
    scale_factor = scale_factor[0]
    sf = int(np.round(max(scale_factor, 1 / scale_factor)))
    key = str(sf)
    pre, post = calc_strides(im, scale_factor, align_center)
    k_post, k_pre = np.maximum(0, pre - post), np.maximum(0, post - pre)     # even sf: one extra row/col
    if isinstance(kernel, np.ndarray):
        if key in imresize.kernels:
            print('Overriding previous kernel with given kernel...')
        assert np.abs(1 - np.sum(kernel)) < np.finfo(np.float32).eps, 'Supplied non-default kernel does not sum to 1'
        up = Center_Mass(np.rot90(kernel, 2), ds_factor=sf) * sf ** 2
        assert up.shape[0] == up.shape[1], 'Only square kernels supported for now'
        assert np.all(np.mod(up.shape + k_post + k_pre - 1, sf) == 0), 'Convolution-invalidated size should be an integer multiplication of sf_4_kernel'
        imresize.kernels[key] = up
    elif key not in imresize.kernels or kernel == 'reset_2_default':
        if key in imresize.kernels:
            print('Overriding previous kernel with default kernel...')
        up = Cubic_Kernel(sf)
        if kernel is not None and 'blurry_cubic' in kernel:
            blur = Gaussian_2D(sigma=float(kernel[len('blurry_cubic_'):]))
            imresize.kernels['blur_' + key] = blur
            up = convolve2d(up, blur)
        imresize.kernels[key] = up
    aa = np.pad(imresize.kernels[key], ((k_pre[0], k_post[0]), (k_pre[1], k_post[1])), mode='constant')
    if scale_factor < 1:
        aa = np.rot90(aa * scale_factor ** 2, 2)
    if return_upscale_kernel:
        return aa
    assert output_shape is None or np.all(scale_factor * np.array(im.shape[:2]) == output_shape[:2])
    half = np.floor(np.array(aa.shape) / 2).astype(np.int32)
    target = scale_factor * np.array(im.shape[:2])
    assert np.all(target == np.round(target)), 'Seems like an attempt to downscale with a factor inducing a non-integer image size'
    target = target.astype(np.int32)
    if im.ndim < 3:
        im = np.expand_dims(im, -1)

    def filt(a):
        if use_zero_padding:
            return convolve2d(a, aa, 'same')
        return convolve2d(np.pad(a, ((half[0], half[0]), (half[1], half[1])), mode='edge'), aa, 'valid')
    planes = []
    for c in range(im.shape[2]):
        if scale_factor > 1:
            stuffed = np.zeros(target, dtype=np.float64)
            stuffed[pre[0]::sf, pre[1]::sf] = im[:, :, c]
            planes.append(filt(stuffed))
        else:
            planes.append(filt(im[:, :, c])[pre[0]::sf, pre[1]::sf])
    return np.squeeze(np.stack(planes, -1))
