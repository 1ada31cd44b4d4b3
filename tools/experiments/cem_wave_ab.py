"""Same-process A/B of the wave-streaming separable CEM kernels (csrc/esr_cem.hip: cem_downscale_wave_kernel, cem_lrfilter_wave_kernel, cem_upscale_wave_kernel) against the tile
kernels they replace at the large image sizes: the instrumented build's switch esr_debug_cem_wave (make -C explorable-super-resolution_amd/csrc trace;
ESR_HIP_LIBRARY=explorable-super-resolution_amd/esr_hip/libesr_hip_trace.so).  Per op and for the whole projection at the configs[1] and configs[4] sizes:
time (GPU events over back-to-back launches) and the largest difference between the two forms.
    ESR_HIP_LIBRARY=$PWD/explorable-super-resolution_amd/esr_hip/libesr_hip_trace.so python tools/experiments/cem_wave_ab.py"""
import os, sys, ctypes as Ct
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import CEM.CEMnet as C
from CEM.imresize_CEM import imresize
from esr_hip import cem_ops, _lib

lib = _lib.load_library()
lib.esr_debug_cem_wave.argtypes = [Ct.c_int]; lib.esr_debug_cem_wave.restype = None


def t_us(f, n=100):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


CASES = (('configs[1] x4 bicubic', 4, None, 32, 128), ('configs[4] x8 blurry_cubic_2.0', 8, 'blurry_cubic_2.0', 16, 256), ('x3 bicubic', 3, None, 8, 96),
         ('x4 blurry_cubic_1.0', 4, 'blurry_cubic_1.0', 8, 128))
for name, sf, kernel, B, lr_size in CASES:
    imresize.kernels = {}
    cem = C.CEMnet(C.Get_CEM_Conf(sf), upscale_kernel=kernel)
    net = cem.WrapArchitecture_PyTorch(generated_image=None).cuda().eval()
    m = int(cem.invalidity_margins_LR)
    torch.manual_seed(1)
    lr = torch.rand(B, 3, lr_size, lr_size, device='cuda')
    g = torch.rand(B, 3, sf * (lr_size + 2 * m), sf * (lr_size + 2 * m), device='cuda')
    e = torch.randn(B, 3, lr_size + 2 * m, lr_size + 2 * m, device='cuda')
    td, ti, tu = net.DownscaleOP.taps(), net.Conv_LR_with_Inv_hTh_OP.taps(), net.Upscale_OP.taps()
    pre = sf - sf // 2 - 1
    ops = (('downscale  lr - D(g)', lambda: cem_ops.downscale_raw(g, td, sf, pre, lr=lr, lr_pad=m)),
           ('downscale  D(g)     ', lambda: cem_ops.downscale_raw(g, td, sf, pre)),
           ('upscale  g + U(e), cropped', lambda: cem_ops.upscale_raw(e, tu, sf, pre, g=g, crop=sf * m, mode=1)),
           ('upscale  U(e)', lambda: cem_ops.upscale_raw(e, tu, sf, pre)),
           ('upscale  tanh form (two inputs)', lambda: cem_ops.upscale_raw(e, tu, sf, pre, f2=e * 0.5, g=g, crop=sf * m, mode=2, rng=0.3)),
           ('LR filter  K(e)', lambda: cem_ops.lr_filter_raw(e, ti)),
           ('projection', lambda: cem_ops.project(lr, g, td, ti, tu, sf, pre, lr_pad=m, crop=sf * m)))
    with torch.no_grad():
        for what, f in ops:
            res, tm = {}, {}
            for rep in range(2):
                for wave in (1, 0):
                    lib.esr_debug_cem_wave(wave)
                    tm[wave] = t_us(f)
                    res[wave] = f().clone()
            lib.esr_debug_cem_wave(1)
            diff = float((res[1] - res[0]).abs().max()) / max(1.0, float(res[0].abs().max()))
            print('%-32s %-34s wave %7.1f us   tile %7.1f us   max |diff| / max |.| = %.2e' % (name, what, tm[1], tm[0], diff), flush=True)
