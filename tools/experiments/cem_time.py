"""The CEM filter kernels and the fused projection at the configs[1] size (x4 bicubic, 32 x 128x128 -> 512x512, G output on the padded 148x148
frame = 592x592) and at the configs[4] size (x8, 'blurry_cubic_2.0': 45x45 / 35x35 taps, 16 x 256x256 -> 2048x2048).
esr_hip.cem_ops.USE_SEPARABLE = False times the general 2-D kernels instead of the separable fast path (the ESR_CEM_SEPARABLE variable of round 2 is no longer read)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import CEM.CEMnet as C
from CEM.imresize_CEM import imresize
from esr_hip import cem_ops


def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for name, sf, kernel, B, lr_size in (('configs[1] x4 bicubic', 4, None, 32, 128), ('configs[4] x8 blurry_cubic_2.0', 8, 'blurry_cubic_2.0', 16, 256)):
    imresize.kernels = {}
    cem = C.CEMnet(C.Get_CEM_Conf(sf), upscale_kernel=kernel)
    net = cem.WrapArchitecture_PyTorch(generated_image=None).cuda().eval()
    m = int(cem.invalidity_margins_LR)
    hp = lr_size + 2 * m
    lr = torch.rand(B, 3, lr_size, lr_size, device='cuda')
    lrp = torch.rand(B, 3, hp, hp, device='cuda')
    g = torch.rand(B, 3, sf * hp, sf * hp, device='cuda')
    td, ti, tu = net.DownscaleOP.taps(), net.Conv_LR_with_Inv_hTh_OP.taps(), net.Upscale_OP.taps()
    pre = sf - sf // 2 - 1
    with torch.no_grad():
        print('%s (taps %d / %d, margin %d): downscale %.3f ms  lrfilter %.3f ms  upscale+combine+crop %.3f ms  fused projection as the forward runs it %.3f ms' % (
            name, td.shape[0], ti.shape[0], m, t(lambda: cem_ops.downscale_raw(g, td, sf, pre, lr=lr, lr_pad=m)), t(lambda: cem_ops.lr_filter_raw(lrp, ti)),
            t(lambda: cem_ops.upscale_raw(lrp, tu, sf, pre, g=g, crop=sf * m, mode=1)),
            t(lambda: cem_ops.project(lr, g, td, ti, tu, sf, pre, lr_pad=m, crop=sf * m))))
        keep = cem_ops.PROJECT_CHUNK_IMAGES
        for nb in (0, 16, 8, 4, 2):                       # the projection per chunk of nb images (0: the whole batch at once)
            cem_ops.PROJECT_CHUNK_IMAGES = nb
            print('    project(), chunks of %2d images: %.3f ms' % (nb, t(lambda: cem_ops.project(lr, g, td, ti, tu, sf, pre, lr_pad=m, crop=sf * m))))
        cem_ops.PROJECT_CHUNK_IMAGES = keep
    del g, lr, lrp
    torch.cuda.empty_cache()
