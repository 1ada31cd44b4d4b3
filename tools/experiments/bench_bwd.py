"""Time forward+backward of the RRDB-23 x4 generator (+CEM) at the C3 per-GPU shape (32 x 52x52 LR, lat 3, train mode, weight grads)
and one Z-search iteration at a C4-like shape (B x 128x128 LR eval mode, dZ only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import CEM.CEMnet as CEMnet
import models.modules.architecture as arch
import models.networks as networks
dev = 'cuda'
which = sys.argv[1] if len(sys.argv) > 1 else 'c3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if which == 'c3' else 16)
lat = 3
torch.manual_seed(0)
cem = CEMnet.CEMnet(CEMnet.Get_CEM_Conf(4))
net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=23, gc=32, upscale=4, norm_type=None, act_type='leakyrelu', mode='CNA', upsample_mode='upconv',
                   latent_input='all_layers_HR_downscaled', num_latent_channels=lat)
G = cem.WrapArchitecture_PyTorch(net)
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    networks.init_weights(G, init_type='kaiming', scale=0.1)
G = G.to(dev)
def ev(): return torch.cuda.Event(enable_timing=True)
if which == 'c3':
    G.train()
    h = 52
    x = torch.rand(B, 3, h, h, device=dev); z = torch.rand(B, lat, 4 * h, 4 * h, device=dev) * 2 - 1
    tgt = torch.rand(B, 3, 4 * h, 4 * h, device=dev)
    inp = torch.cat([z.view(B, lat * 16, h, h), x], 1)
    def step():
        for p in G.parameters(): p.grad = None
        y = G(inp)
        loss = (y - tgt).abs().mean()
        loss.backward()
        return loss
else:
    G.eval()
    for p in G.parameters(): p.requires_grad_(False)
    h = 128
    x = torch.rand(1, 3, h, h, device=dev).expand(B, -1, -1, -1).contiguous()
    z = (torch.rand(B, lat, 4 * h, 4 * h, device=dev) * 2 - 1).requires_grad_(True)
    def step():
        z.grad = None
        y = G(torch.cat([z.view(B, lat * 16, h, h), x], 1))
        loss = -torch.std(torch.clamp(y, 0, 1), dim=(1, 2, 3)).mean()
        loss.backward()
        return loss
for _ in range(2): step()
torch.cuda.synchronize()
n = 5
e0, e1 = ev(), ev()
e0.record()
for _ in range(n): l = step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
# forward only for comparison
with torch.no_grad():
    inp2 = torch.cat([z.detach().view(B, lat * 16, h, h), x], 1)
    for _ in range(2): G(inp2)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): G(inp2)
    e1.record(); torch.cuda.synchronize()
print('%s B=%d h=%d: fwd+bwd %.1f ms/step, fwd only %.1f ms, loss %.4f, peak mem %.1f GB' % (which, B, h, ms, e0.elapsed_time(e1) / n, float(l), torch.cuda.max_memory_allocated() / 2**30))
