"""Fixed cost of a small conv launch (32 x 52 x 52, 32 output channels, bf16: 256 tiles = one workgroup per CU): time per launch as a function of
the input groups (K chunks) for the forward form (bias + LeakyReLU), the data-gradient form (LeakyReLU' mask from another buffer) and variants,
each as 200 back-to-back launches on alternating buffers.  a + b * chunks: what a launch costs besides its K loop.
    python tools/experiments/small_conv_fixed_cost.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
from esr_hip import act as A

dev = torch.device('cuda')
B, H, W = 32, 52, 52
split = False
torch.manual_seed(0)
bufs = [A.ActBuf(B, 24, H, W, dev, split) for _ in range(6)]
for b in bufs:
    b.hi[:, :, 1:-1, 1:-1] = (torch.randn(B, 24, H, W, 8, device=dev) * 0.5).to(torch.bfloat16).view(torch.int16)
masks = [A.ActBuf(B, 24, H, W, dev, split) for _ in range(6)]
for b in masks:
    b.hi[:, :, 1:-1, 1:-1] = (torch.randn(B, 24, H, W, 8, device=dev)).to(torch.bfloat16).view(torch.int16)
zb = A.ActBuf(B, 1, H, W, dev, split)


def run(fn, n=200):
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, kw_of in (('forward form (bias, LeakyReLU)', lambda i: dict(act_slope=0.2)),
                    ('forward form + latent group in0', lambda i: dict(act_slope=0.2, in0=zb.view())),
                    ('plain (no bias, no activation)', lambda i: dict(use_bias=False)),
                    ('data-gradient form (mask from a cold buffer)', lambda i: dict(use_bias=False, mask_src=masks[i % 6].view(8, 4), mask_cg=(0, 4), mask_slope=0.2)),
                    ('data-gradient form (mask from ONE buffer)', lambda i: dict(use_bias=False, mask_src=masks[0].view(8, 4), mask_cg=(0, 4), mask_slope=0.2))):
    out = []
    for g in (8, 12, 16, 20):
        w = torch.randn(32, g * 8 + (3 if 'latent' in name else 0), 3, 3, device=dev) * 0.05
        pc = A.PackedConv(w, torch.zeros(32, device=dev), 3 if 'latent' in name else 0, split=split).get()
        t = run(lambda i: A.conv3x3(pc, bufs[i % 6].view(0, g), B, H, W, 32, out=bufs[i % 6].view(g, 4), **kw_of(i)))
        out.append(t)
    b = (out[3] - out[0]) / 6
    print('%-48s groups 8/12/16/20: %s us   -> %.2f us per chunk, fixed %.2f us' % (name, ' '.join('%.1f' % t for t in out), b, out[0] - 4 * b))
