"""Input-gradient accuracy of the precision modes against the reference's golden gradients (fixture F4) and, at full depth (RRDB-23,
kaiming x0.1 weights), against autograd through the fp32 CPU oracle.  RRDBEngine.mixed_bwd = 'bf16' | 'f16' selects the data-gradient format of 'mixed' (the ESR_MIXED_BWD variable of round 2 is no longer read)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_gpu_backward as T
from oracle import rrdb_oracle as ro

def metrics(got, ref):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    rms = np.sqrt((ref ** 2).mean())
    return np.median(np.abs(got - ref)) / rms, float(np.linalg.norm(got - ref) / np.linalg.norm(ref))

g = T.load('rrdb_fwd_bwd.npz')
for prec in sys.argv[1:] or ['split', 'mixed']:
    for name, nb, sf, lat in T.F4_CASES:
        net = T._rrdb(nb, sf, lat).to('cuda')
        net.set_precision(prec)
        for p in net.parameters(): p.requires_grad_(False)
        x = T._f4_input(nb, sf, lat).to('cuda').requires_grad_(True)
        y = net(x)
        cot = T.seeded_uniform(tuple(y.shape), 41 + nb + sf + lat, -1.0, 1.0).to('cuda')
        (y * cot).sum().backward()
        med, l2 = metrics(x.grad.cpu().numpy(), g[name + '/dx'])
        print('%-6s %-12s dx vs reference golden: median/rms %.2e  rel_l2 %.2e' % (prec, name, med, l2))
    # full depth, training-scale weights, latent 3
    import models.modules.architecture as arch, models.networks as N
    torch.manual_seed(0)
    net = arch.RRDBNet(3, 3, 64, 23, gc=32, upscale=4, latent_input='all_layers_HR_downscaled', num_latent_channels=3)
    N.init_weights(net, 'kaiming', scale=0.1)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    x0 = torch.rand(1, 3 + 48, 24, 24); x0[:, :48] = x0[:, :48] * 2 - 1
    xc = x0.clone().requires_grad_(True)
    yr = ro.rrdb_forward(sd, xc, 23, 4, 3)
    cot = torch.rand_like(yr) * 2 - 1
    (yr * cot).sum().backward()
    net = net.to('cuda'); net.set_precision(prec)
    for p in net.parameters(): p.requires_grad_(False)
    xg = x0.clone().cuda().requires_grad_(True)
    (net(xg) * cot.cuda()).sum().backward()
    med, l2 = metrics(xg.grad.cpu().numpy(), xc.grad.numpy())
    print('%-6s RRDB-23 lat3 dx vs oracle autograd: median/rms %.2e  rel_l2 %.2e  (tiny cotangent: %s)' % (prec, med, l2, 'n/a'))
    # the same with a cotangent of size 1e-7 (a mean-reduced loss): exercises the power-of-two gradient scaling
    xg2 = x0.clone().cuda().requires_grad_(True)
    (net(xg2) * (cot.cuda() * 1e-7)).sum().backward()
    med, l2 = metrics(xg2.grad.cpu().numpy() * 1e7, xc.grad.numpy())
    print('%-6s RRDB-23 lat3 dx, cotangent x1e-7:    median/rms %.2e  rel_l2 %.2e' % (prec, med, l2))
