"""Parameter-gradient accuracy of the training-capable precision modes vs autograd through the fp32 CPU oracle (RRDB-nb, kaiming x0.1
weights, latent 3, mean-reduced L1-like cotangent).  Prints the worst and the median per-tensor relative L2 over all conv weights/biases."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import models.modules.architecture as arch, models.networks as N
from oracle import rrdb_oracle as ro
nb, lat = int(os.environ.get('NB', 4)), 3
torch.manual_seed(3)
net = arch.RRDBNet(3, 3, 64, nb, gc=32, upscale=4, latent_input='all_layers_HR_downscaled', num_latent_channels=lat)
N.init_weights(net, 'kaiming', scale=0.1)
for m in net.modules():
    if isinstance(m, torch.nn.Conv2d): torch.nn.init.normal_(m.bias, 0, 0.05)
sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
x0 = torch.rand(4, 3 + 16 * lat, 20, 24); x0[:, :16 * lat] = x0[:, :16 * lat] * 2 - 1
cot = torch.sign(torch.rand(4, 3, 80, 96) - 0.5) / (4 * 3 * 80 * 96)        # d(mean |.|)
(ro.rrdb_forward(sd, x0, nb, 4, lat) * cot).sum().backward()
ref = {k: v.grad.numpy() for k, v in sd.items()}
net = net.cuda()
for prec in sys.argv[1:] or ['split', 'mixed', 'bf16']:
    net.set_precision(prec)
    net.zero_grad(set_to_none=True)
    (net(x0.cuda()) * cot.cuda()).sum().backward()
    errs = []
    for k, p in net.named_parameters():
        g, r = p.grad.cpu().numpy().astype(np.float64), ref[k].astype(np.float64)
        errs.append((float(np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30)), k))
    if os.environ.get('VERBOSE'):
        for e, k in errs:
            if 'RDB' not in k or 'sub.0.RDB1' in k or 'sub.%d.RDB3' % (nb - 1) in k: print('   %-44s %.2e' % (k, e))
    errs.sort()
    print('%-6s per-tensor rel_l2 of dW/db vs oracle autograd: median %.2e  worst %.2e (%s)  finite %s' % (
        prec, errs[len(errs) // 2][0], errs[-1][0], errs[-1][1], all(np.isfinite(p.grad.cpu().numpy()).all() for p in net.parameters())))
