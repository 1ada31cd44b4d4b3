"""Host-side cost of one generator forward + backward at the configs[2] per-GPU shape (the launch planner is Python): cProfile of
engine.run_forward(keep=True) + engine.run_backward(need_dw=True) called directly in the main thread, and the wall time of the pair with and
without a device synchronisation in between (host-bound if the two are equal)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch, contextlib
import models.modules.architecture as arch, models.networks as networks

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
torch.manual_seed(0)
net = arch.RRDBNet(3, 3, 64, 23, upscale=4, latent_input='all_layers_HR_downscaled', num_latent_channels=3)
with contextlib.redirect_stdout(io.StringIO()):
    networks.init_weights(net, 'kaiming', 0.1)
net = net.cuda().train()
net.set_precision(prec)
eng = net.engine
x = torch.rand(32, 51, 52, 52, device='cuda')
dg = torch.rand(32, 3, 208, 208, device='cuda')


def step():
    g, bufs = eng.run_forward(x, 0, keep=True)
    return eng.run_backward(tuple(x.shape), 0, bufs, dg, need_dx=False, need_dw=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t_host = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 10
print('%s: host enqueue time %.1f ms per forward+backward, wall incl. GPU drain %.1f ms' % (prec, t_host * 1e3, t_all * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print('\n'.join(l[:150] for l in s.getvalue().splitlines()[4:40]))
