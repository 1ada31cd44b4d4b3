"""Same-process A/B of the LR filter folded into the upscale launch (esr_hip.cem_ops.FUSE_FILTER_UPSCALE -> esr_cem_filter_upscale_sep; VERDICT r5
item 6) against the two launches: the projection at the configs[1] and configs[4] sizes, and configs[0] on the GPU (RRDB-3 x4 + CEM on one 32 x 32 frame).
    python tools/experiments/cem_fold_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import CEM.CEMnet as C
from CEM.imresize_CEM import imresize
from esr_hip import cem_ops
import bench


def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for name, sf, kernel, B, lr_size in (('configs[1] x4 bicubic', 4, None, 32, 128), ('configs[4] x8 blurry_cubic_2.0', 8, 'blurry_cubic_2.0', 16, 256)):
    imresize.kernels = {}
    cem = C.CEMnet(C.Get_CEM_Conf(sf), upscale_kernel=kernel)
    net = cem.WrapArchitecture_PyTorch(generated_image=None).cuda().eval()
    m = int(cem.invalidity_margins_LR)
    lr = torch.rand(B, 3, lr_size, lr_size, device='cuda')
    g = torch.rand(B, 3, sf * (lr_size + 2 * m), sf * (lr_size + 2 * m), device='cuda')
    td, ti, tu = net.DownscaleOP.taps(), net.Conv_LR_with_Inv_hTh_OP.taps(), net.Upscale_OP.taps()
    pre = sf - sf // 2 - 1
    with torch.no_grad():
        for rep in range(3):
            for fuse in (True, False):
                cem_ops.FUSE_FILTER_UPSCALE = fuse
                print('%s: projection, filter folded into the upscale launch %-5s: %.3f ms' % (name, fuse, t(lambda: cem_ops.project(lr, g, td, ti, tu, sf, pre, lr_pad=m, crop=sf * m))), flush=True)
cem, G = bench.build_model('cuda', nb=3)
x = torch.rand(1, 3, 32, 32, device='cuda')
G.generated_image_model.set_precision('bf16')
with torch.no_grad():
    for rep in range(3):
        for fuse in (True, False):
            cem_ops.FUSE_FILTER_UPSCALE = fuse
            print('configs[0] RRDB-3 x4 + CEM, 1 x 32x32, bf16, folded %-5s: %.1f us per forward' % (fuse, t(lambda: G(x), 200) * 1e3), flush=True)
