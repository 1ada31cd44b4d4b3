import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch, contextlib, io
import CEM.CEMnet as CEMnet, models.modules.architecture as arch, models.networks as networks
from esr_hip.graph import GraphedForward
for nb, shape in [(3, (1, 3, 32, 32)), (23, (1, 3, 32, 32)), (23, (4, 3, 64, 64))]:
    torch.manual_seed(0)
    cem = CEMnet.CEMnet(CEMnet.Get_CEM_Conf(4))
    G = cem.WrapArchitecture_PyTorch(arch.RRDBNet(3, 3, 64, nb, upscale=4, num_latent_channels=0))
    with contextlib.redirect_stdout(io.StringIO()):
        networks.init_weights(G, 'kaiming', 0.1)
    G = G.cuda().eval()
    x = torch.rand(*shape, device='cuda')
    fast = GraphedForward(G)
    with torch.no_grad():
        y0 = G(x).clone()
        y1 = fast(x).clone()
        assert torch.equal(y0, y1), float((y0 - y1).abs().max())
        x2 = torch.rand(*shape, device='cuda')
        assert torch.equal(G(x2), fast(x2))
        def t(f, n=50):
            for _ in range(5): f(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): f(x)
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
        print('RRDB-%d %s: eager %.3f ms, graph replay %.3f ms' % (nb, shape, t(G), t(fast)))
