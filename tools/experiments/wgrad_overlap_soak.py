"""Soak of the two-stream backward: two generators with the same weights (RRDB-23 x4 + CEM, bf16, 32 x 52x52), one with RRDBEngine.wgrad_overlap = 3, one with
the single launch, N optimiser steps on fresh random inputs; every step's gradients and the final weights must be bit-identical.
    python tools/experiments/wgrad_overlap_soak.py [steps]"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import CEM.CEMnet as CEMnet
import models.modules.architecture as arch
import models.networks as networks
import contextlib, io
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev, B, lat, h = 'cuda', 32, 3, 52


def make():
    torch.manual_seed(0)
    cem = CEMnet.CEMnet(CEMnet.Get_CEM_Conf(4))
    net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=23, gc=32, upscale=4, norm_type=None, act_type='leakyrelu', mode='CNA', upsample_mode='upconv',
                       latent_input='all_layers_HR_downscaled', num_latent_channels=lat)
    G = cem.WrapArchitecture_PyTorch(net)
    with contextlib.redirect_stdout(io.StringIO()):
        networks.init_weights(G, init_type='kaiming', scale=0.1)
    G = G.to(dev).train()
    net.set_precision('bf16')
    return G, net


Ga, na = make()
Gb, nb_ = make()
na.engine.wgrad_overlap, nb_.engine.wgrad_overlap = 3, 0
oa = torch.optim.Adam([p for p in Ga.parameters() if p.requires_grad], lr=1e-4)
ob = torch.optim.Adam([p for p in Gb.parameters() if p.requires_grad], lr=1e-4)
gen = torch.Generator(device=dev).manual_seed(1)
bad = 0
for it in range(N):
    x = torch.rand(B, 3, h, h, device=dev, generator=gen); z = torch.rand(B, lat * 16, h, h, device=dev, generator=gen) * 2 - 1
    tgt = torch.rand(B, 3, 4 * h, 4 * h, device=dev, generator=gen)
    inp = torch.cat([z, x], 1)
    for G, o in ((Ga, oa), (Gb, ob)):
        o.zero_grad(set_to_none=True)
        (G(inp) - tgt).abs().mean().backward()
    same = all(torch.equal(p.grad, q.grad) for p, q in zip(Ga.parameters(), Gb.parameters()) if p.grad is not None)
    bad += not same
    oa.step(); ob.step()
    if it % 20 == 0 or not same:
        print('step %3d: gradients identical %s' % (it, same), flush=True)
w_same = all(torch.equal(p, q) for p, q in zip(Ga.parameters(), Gb.parameters()))
print('%d steps: %d with differing gradients; final weights identical: %s' % (N, bad, w_same))
sys.exit(1 if (bad or not w_same) else 0)
