"""What a captured HIP graph buys for a Z-search style step at interactive sizes (one 128x128 LR image, B Z samples): forward through
G + CEM, a scalar objective, backward to the input (weights frozen), all replayed as ONE graph launch.  Checks the replayed gradient
against the eager one bit for bit and times both.  Experiment only: the product's Z_optimizer runs eagerly."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch, contextlib, io
import CEM.CEMnet as CEMnet, models.modules.architecture as arch, models.networks as networks
prec = sys.argv[1] if len(sys.argv) > 1 else 'mixed'
for B in (1, 4):
    torch.manual_seed(0)
    cem = CEMnet.CEMnet(CEMnet.Get_CEM_Conf(4))
    G = cem.WrapArchitecture_PyTorch(arch.RRDBNet(3, 3, 64, 23, upscale=4, latent_input='all_layers_HR_downscaled', num_latent_channels=3))
    with contextlib.redirect_stdout(io.StringIO()):
        networks.init_weights(G, 'kaiming', 0.1)
    G = G.cuda().eval()
    G.generated_image_model.set_precision(prec)
    for p in G.parameters(): p.requires_grad_(False)
    x = torch.rand(B, 51, 128, 128, device='cuda'); x[:, :48] = x[:, :48] * 2 - 1
    xs = x.clone().requires_grad_(True)

    def step():
        xs.grad = None
        out = G(xs)
        loss = out.std(dim=(1, 2, 3)).sum()
        loss.backward()
        return loss

    for _ in range(3): step()
    torch.cuda.synchronize()
    g_eager = xs.grad.clone()
    # capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    xs.grad = None
    with torch.cuda.graph(graph):
        out = G(xs)
        loss = out.std(dim=(1, 2, 3)).sum()
        loss.backward()
    graph.replay(); torch.cuda.synchronize()
    same = torch.equal(xs.grad, g_eager)
    def t(f, n=30):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print('%s B=%d: eager %.2f ms, graph replay %.2f ms, gradient identical: %s (max diff %.2e)' % (prec, B, t(step), t(graph.replay), same, float((xs.grad - g_eager).abs().max())))
