"""Host-side cost of the configs[2] training step (SRRaGANModel.optimize_parameters) at a small batch, where the GPU work is short and the
step time IS the host time: wall per step and a cProfile of the main thread.
    python tools/experiments/host_profile_step.py [bf16|split] [batch] [with_D]"""
import cProfile, contextlib, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import bench_paths
import models

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
with_D = len(sys.argv) > 3 and sys.argv[3] == '1'
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = models.create_model(bench_paths.make_opt(True, with_D=with_D))
if prec != 'split':
    model.netG.generated_image_model.set_precision(prec)
if with_D and prec == 'bf16':
    model.D_dtype = torch.bfloat16
data = {'LR': torch.rand(B, 3, 52, 52).cuda(), 'HR': torch.rand(B, 3, 208, 208).cuda(), 'Z': (torch.rand(B, 3, 208, 208) * 2 - 1).cuda()}


def step():
    model.feed_data(data); model.optimize_parameters()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t_host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 20
print('%s batch %d with_D %s: host enqueue %.1f ms per step, wall incl. GPU drain %.1f ms' % (prec, B, with_D, t_host * 1e3, t_all * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30)
print('\n'.join(l[:160] for l in s.getvalue().splitlines()[4:48]))
