"""Diagnostic: the critic's backward pieces one by one against torch autograd in float64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from oracle.weights import fill_formula_weights, seeded_uniform
import models.modules.architecture as arch
from esr_hip import critic as K

netD = arch.Discriminator_VGG_128(3, 64, input_patch_size=64)
fill_formula_weights(netD, gain=1.0)
netD = netD.cuda().train()
eng = K.CriticEngine(netD, 'split')
eng.refresh()
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
L0, L1, L2, L3 = eng.layers[:4]
B, H, W = 3, 32, 48
# ---- dgrad of a 3x3 layer (64 -> 128) and its weight gradient
x = seeded_uniform((B, 64, H, W), 1).cuda() - 0.5
dy = seeded_uniform((B, 128, H, W), 2).cuda() - 0.5
xa, dya = K._PackIn.apply(x, 2), K._PackIn.apply(dy, 2)
print('pack/unpack roundtrip', rel(K._UnpackOut.apply(xa, 64), x))
xr = x.double().requires_grad_(True)
w = L2.conv.weight.double().detach().requires_grad_(True)
yr = F.conv2d(xr, w, None, padding=1)
yr.backward(dy.double())
with torch.no_grad():
    print('3x3 fwd  ', rel(K._UnpackOut.apply(eng.conv_fwd(L2, xa, use_bias=False), 128), yr))
    print('3x3 dgrad', rel(K._UnpackOut.apply(eng.conv_dgrad(L2, dya), 64), xr.grad))
    dw, db = eng.conv_wgrad(L2, dya, xa)
    print('3x3 wgrad', rel(dw, w.grad), 'bgrad', rel(db, dy.double().sum((0, 2, 3))))
# ---- strided layer (64 -> 64, 4x4 s2): input stored space-to-depth
x = seeded_uniform((B, 64, H, W), 3).cuda() - 0.5
dy = seeded_uniform((B, 64, H // 2, W // 2), 4).cuda() - 0.5
xs = x.view(B, 8, 8, H // 2, 2, W // 2, 2).permute(0, 1, 4, 6, 2, 3, 5).reshape(B, 256, H // 2, W // 2).contiguous()
xa, dya = K._PackIn.apply(xs, 2), K._PackIn.apply(dy, 2)
xr = x.double().requires_grad_(True)
w = L1.conv.weight.double().detach().requires_grad_(True)
yr = F.conv2d(xr, w, None, stride=2, padding=1)
yr.backward(dy.double())
with torch.no_grad():
    print('4x4s2 fwd  ', rel(K._UnpackOut.apply(eng.conv_fwd(L1, xa, use_bias=False), 64), yr))
    gx = K._UnpackOut.apply(eng.conv_dgrad(L1, dya), 256)
    gx = gx.view(B, 8, 2, 2, 8, H // 2, W // 2).permute(0, 1, 4, 5, 2, 6, 3).reshape(B, 64, H, W)
    print('4x4s2 dgrad', rel(gx, xr.grad))
    dw, db = eng.conv_wgrad(L1, dya, xa)
    print('4x4s2 wgrad', rel(dw, w.grad))
# ---- BatchNorm + LeakyReLU backward and double backward (layer 2's norm: 128 channels)
y = seeded_uniform((B, 128, H, W), 5).cuda() * 3 - 1
dz = seeded_uniform((B, 128, H, W), 6).cuda() - 0.5
u = seeded_uniform((B, 128, H, W), 7).cuda() - 0.5
g0, b0 = L2.bn.weight, L2.bn.bias
yr = y.double().requires_grad_(True)
gr, br = g0.double().detach().requires_grad_(True), b0.double().detach().requires_grad_(True)
zr = F.leaky_relu(F.batch_norm(yr, None, None, gr, br, training=True, eps=L2.bn.eps), 0.2)
dzr = dz.double().requires_grad_(True)
dyr, dgr = torch.autograd.grad(zr, [yr, gr], dzr, create_graph=True)
gy, gg, gdz = torch.autograd.grad(dyr, [yr, gr, dzr], u.double())
ya = K._PackIn.apply(y, 2).requires_grad_(True)
gp_, bp_ = g0.detach().clone().requires_grad_(True), b0.detach().clone().requires_grad_(True)
za = K._BNAct.apply(eng, L2, ya, gp_, bp_, False, True)
print('bn fwd', rel(K._UnpackOut.apply(za.detach(), 128), zr))
dza = K._PackIn.apply(dz, 2).requires_grad_(True)
dya, dga = torch.autograd.grad(za, [ya, gp_], dza, create_graph=True)
print('bn bwd  dy', rel(K._UnpackOut.apply(dya.detach(), 128), dyr), 'dgamma', rel(dga, dgr))
ua = K._PackIn.apply(u, 2)
gya, gga, gdza = torch.autograd.grad(dya, [ya, gp_, dza], ua)
print('bn bwd2 g_y', rel(K._UnpackOut.apply(gya, 128), gy), 'g_gamma', rel(gga, gg), 'g_dz', rel(K._UnpackOut.apply(gdza, 128), gdz))
