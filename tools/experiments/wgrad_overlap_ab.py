import os, sys, time
ROOT='/root/repo'
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import CEM.CEMnet as CEMnet
import models.modules.architecture as arch
import models.networks as networks
dev, B, lat, h = 'cuda', 32, 3, 52
torch.manual_seed(0)
cem = CEMnet.CEMnet(CEMnet.Get_CEM_Conf(4))
net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=23, gc=32, upscale=4, norm_type=None, act_type='leakyrelu', mode='CNA', upsample_mode='upconv',
                   latent_input='all_layers_HR_downscaled', num_latent_channels=lat)
G = cem.WrapArchitecture_PyTorch(net)
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    networks.init_weights(G, init_type='kaiming', scale=0.1)
G = G.to(dev).train(); net.set_precision('bf16')
x = torch.rand(B, 3, h, h, device=dev); z = torch.rand(B, lat, 4 * h, 4 * h, device=dev) * 2 - 1
tgt = torch.rand(B, 3, 4 * h, 4 * h, device=dev)
inp = torch.cat([z.view(B, lat * 16, h, h), x], 1)
params = list(G.parameters())
def step():
    for p in params: p.grad = None
    loss = (G(inp) - tgt).abs().mean(); loss.backward(); return loss
def timed(n=10):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
eng = net.engine
from esr_hip import _lib
if os.environ.get('OVL_NORMAL_FORM'):          # side groups in the two-workgroups-per-CU form
    _lib.lib.esr_conv3x3_wgrad_batch_run_side = _lib.lib.esr_conv3x3_wgrad_batch_run
if os.environ.get('OVL_PRIORITY'):
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
    eng._side = torch.cuda.Stream(priority=int(os.environ['OVL_PRIORITY']))
    print('side stream priority', eng._side.priority)
ref = None
for g in [0] + [int(v) if v.isdigit() else tuple(float(f) for f in v.split(',')) for v in sys.argv[1:]] + [0]:
    eng.wgrad_overlap = g
    t = timed()
    step(); torch.cuda.synchronize()
    grads = [p.grad.clone() for p in params if p.grad is not None]
    if ref is None: ref = grads
    same = all(torch.equal(a, b) for a, b in zip(ref, grads))
    print('groups %-22s: fwd + bwd %.2f ms   gradients bit-identical to one launch: %s' % (str(g), t, same), flush=True)
