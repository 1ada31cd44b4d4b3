"""Is the forward host-bound at small launch sizes?  Eager vs HIP-graph replay of the RRDB-23 generator (no CEM) per precision at the configs[2]
per-GPU shape (32 x 52 x 52, latent 3) and for one 128 x 128 image (the GUI's case)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch, contextlib, io
import models.modules.architecture as arch, models.networks as networks
from esr_hip.graph import GraphedForward


def t(f, x, n=30):
    for _ in range(5): f(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for lat, shape in [(3, (32, 3 + 48, 52, 52)), (0, (1, 3, 128, 128)), (0, (1, 3, 32, 32))]:
    torch.manual_seed(0)
    net = arch.RRDBNet(3, 3, 64, 23, upscale=4, latent_input='all_layers_HR_downscaled' if lat else None, num_latent_channels=lat)
    with contextlib.redirect_stdout(io.StringIO()):
        networks.init_weights(net, 'kaiming', 0.1)
    net = net.cuda().eval()
    x = torch.rand(*shape, device='cuda')
    for prec in ('split', 'mixed', 'bf16'):
        net.set_precision(prec)
        fast = GraphedForward(net)
        with torch.no_grad():
            print('%s %s: eager %.3f ms, graph replay %.3f ms' % (shape, prec, t(net, x), t(fast, x)), flush=True)
