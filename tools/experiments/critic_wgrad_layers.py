"""Per-layer time of the critic's batched weight gradient (each layer as its own launch) at 32 x 3 x 128 x 128.
    python tools/experiments/critic_wgrad_layers.py [bf16|split]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import models.modules.architecture as arch
from esr_hip import critic as K

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
torch.manual_seed(0)
netD = arch.Discriminator_VGG_128(3, 64).cuda().train()
eng = K.CriticEngine(netD, prec)
x = torch.rand(32, 3, 128, 128, device='cuda')
out = K.critic_forward(eng, x)
out.mean().backward()
feat, S = K._fwd_pass(eng, x, True)
K._bwd_pass(eng, S, torch.randn_like(feat), None, False, True)
bs = S.bs


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


pairs = [(L, bs.dy[i], bs.z[i - 1] if i > 0 else bs.t0) for i, L in enumerate(eng.layers)]
tot = 0.0
for i, p in enumerate(pairs):
    ws = K._WgradSet(eng, bs, [p])
    t = timed(ws.run)
    L = p[0]
    h, w = bs.hw[i]
    macs = 32 * h * w * L.cout * L.cin * (16 if L.strided else 9)
    tot += t
    print('layer %d  %4d -> %4d %s  map %3dx%-3d  %7.1f us   %6.1f GMAC  %.3f PF' % (i, L.cin, L.cout, 's2 4x4' if L.strided else '   3x3', h, w, t, macs / 1e9, 2 * macs / t / 1e9))
ws = K._WgradSet(eng, bs, pairs)
print('sum of layers %.1f us; one batched launch %.1f us' % (tot, timed(ws.run)))
