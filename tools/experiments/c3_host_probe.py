"""Where the configs[2] step waits for the host: per step, host time spent inside optimize_parameters() (no synchronisation), GPU-side
step time, and how often the one-launch Adam re-uploads its table (esr_adam_upload blocks the host until the stream has drained).
    python tools/experiments/c3_host_probe.py [steps]"""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import bench_paths
import models
from esr_hip import _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = models.create_model(bench_paths.make_opt(True, with_D=True))
model.netG.generated_image_model.set_precision('bf16')
model.D_dtype = torch.bfloat16
g = torch.Generator().manual_seed(2000)
data = {'LR': torch.rand(32, 3, 52, 52, generator=g).cuda(), 'HR': torch.rand(32, 3, 208, 208, generator=g).cuda(),
        'Z': (torch.rand(32, 3, 208, 208, generator=g) * 2 - 1).cuda()}
uploads = {'adam': 0, 'wgrad': 0, 'pack': 0}
lib = _lib.lib
for name, key in (('esr_adam_table', 'adam'), ('esr_conv3x3_wgrad_batch_upload', 'wgrad'), ('esr_pack_batch_upload', 'pack')):
    f = getattr(lib, name)
    def wrap(*a, _f=f, _k=key):
        uploads[_k] += 1
        return _f(*a)
    setattr(lib, name, wrap)
for _ in range(4):
    model.feed_data(data); model.optimize_parameters()
torch.cuda.synchronize()
for k in uploads:
    uploads[k] = 0
host = []
t0 = time.perf_counter()
for _ in range(steps):
    h0 = time.perf_counter()
    model.feed_data(data); model.optimize_parameters()
    host.append((time.perf_counter() - h0) * 1e3)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tg = time.perf_counter() - t0
print('steps %d: wall %.2f ms/step (host loop returned after %.2f ms/step); host per step min %.2f median %.2f max %.2f' %
      (steps, tg / steps * 1e3, th / steps * 1e3, min(host), sorted(host)[len(host) // 2], max(host)))
print('table uploads in the timed steps:', uploads)
