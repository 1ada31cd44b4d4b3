"""Diagnostic: error of the critic's forward features and of d(sum feat*cot)/dx after the first k conv blocks, HIP 'split' vs float64."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
from oracle.weights import fill_formula_weights, seeded_uniform
import models.modules.architecture as arch
from esr_hip import critic as K

size, batch = 64, 4
netD = arch.Discriminator_VGG_128(3, 64, input_patch_size=size)
fill_formula_weights(netD, gain=1.0)
netD = netD.cuda().train()
net64 = copy.deepcopy(netD).double()
eng = K.CriticEngine(netD, sys.argv[1] if len(sys.argv) > 1 else 'split')
eng.refresh()
x0 = seeded_uniform((batch, 3, size, size), 11).cuda()
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
mods = list(net64.features)
mods32 = list(netD.features)
ends = [i for i, m in enumerate(mods) if isinstance(m, torch.nn.LeakyReLU)]
for k in range(1, len(eng.layers) + 1):
    xr = x0.double().requires_grad_(True)
    h = xr
    for m in mods[:ends[k - 1] + 1]:
        h = m(h) if not isinstance(m, torch.nn.LeakyReLU) else torch.nn.functional.leaky_relu(h, 0.2)
    cot = seeded_uniform(tuple(h.shape), 100 + k).cuda().double() - 0.5
    (h * cot).sum().backward()
    x32 = x0.clone().requires_grad_(True)
    h32 = x32
    for m in mods32[:ends[k - 1] + 1]:
        h32 = m(h32) if not isinstance(m, torch.nn.LeakyReLU) else torch.nn.functional.leaky_relu(h32, 0.2)
    (h32 * cot.float()).sum().backward()
    xa = x0.clone().requires_grad_(True)
    t = K._PackIn.apply(xa, eng.planes)
    for i, L in enumerate(eng.layers[:k]):
        nxt = (i + 1 < k) and eng.layers[i + 1].strided
        y = K._Conv.apply(eng, L, t, L.conv.weight, L.conv.bias)
        t = K._BNAct.apply(eng, L, y, *((L.bn.weight, L.bn.bias) if L.bn is not None else (None, None)), nxt, True)
    feat = K._UnpackOut.apply(t, eng.layers[k - 1].cout)
    (feat * cot.float()).sum().backward()
    flips = int(((feat.detach() > 0) != (h.detach() > 0)).sum())
    print('k=%2d  feat: hip %.2e torch32 %.2e   dx: hip %.2e torch32 %.2e   sign flips of the features %d / %d' % (
        k, rel(feat, h), rel(h32, h), rel(xa.grad, xr.grad), rel(x32.grad, xr.grad), flips, feat.numel()))
