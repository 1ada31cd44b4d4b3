"""Per-workgroup phase stamps of the batched weight-gradient launch of one generator training step at the configs[2] per-GPU shape (DESIGN.md
3.3): copy issue / landing wait / barrier / MFMA phase per tile, workgroup durations, shader clock.  Needs the instrumented build:
    make -C explorable-super-resolution_amd/csrc trace
    ESR_HIP_LIBRARY=explorable-super-resolution_amd/esr_hip/libesr_hip_trace.so python tools/experiments/trace_wgrad.py [split|bf16]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np, torch, contextlib, io
import models, bench_paths
from esr_hip import _lib
prec = sys.argv[1] if len(sys.argv) > 1 else 'split'
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    m = models.create_model(bench_paths.make_opt(True))
if prec != 'split':
    m.netG.generated_image_model.set_precision(prec)
dev = 'cuda'
data = {'LR': torch.rand(32, 3, 52, 52).to(dev), 'HR': torch.rand(32, 3, 208, 208).to(dev), 'Z': (torch.rand(32, 3, 208, 208) * 2 - 1).to(dev)}
for _ in range(4):
    m.feed_data(data); m.optimize_parameters()
torch.cuda.synchronize()
lib = _lib.load_library()
nwg = 4096
tb = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
lib.esr_debug_trace_wgrad.argtypes = [C.c_void_p]; lib.esr_debug_trace_wgrad.restype = None
lib.esr_debug_trace_wgrad(tb.data_ptr())
m.feed_data(data); m.optimize_parameters(); torch.cuda.synchronize()
lib.esr_debug_trace_wgrad(None)
t = tb.cpu().numpy().reshape(nwg, 64).astype(np.int64)
if os.environ.get('TRACE_BLOCKS') == '1':      # the block-form launch's slots (conv3x3_wgrad_block_kernel writes from slot 2304 on)
    t = t[2304:]
else:
    t = t[:2304]
t = t[t[:, 1] > 0]
n = len(t)
rt0, rt1 = t[:, 62], t[:, 63]
print('%s: %d workgroups traced; launch span %.2f ms; workgroup duration ms: median %.2f p10 %.2f p90 %.2f max %.2f' % (
    prec, n, (rt1.max() - rt0.min()) / 1e5, np.median(rt1 - rt0) / 1e5, np.percentile(rt1 - rt0, 10) / 1e5, np.percentile(rt1 - rt0, 90) / 1e5, (rt1 - rt0).max() / 1e5))
ts = t[:, 2:57]                       # 11 tiles x 5 stamps
ph = {}
names = ['issue', 'wait', 'barrier1', 'mfma', 'barrier2+loop']
for i, nm in enumerate(names):
    if i < 4:
        ph[nm] = np.array([ts[:, 5 * k + i + 1] - ts[:, 5 * k + i] for k in range(10)]).T
    else:
        ph[nm] = np.array([ts[:, 5 * (k + 1)] - ts[:, 5 * k + 4] for k in range(10)]).T
tot = sum(a.mean() for a in ph.values())
for nm, a in ph.items():
    print('  %-14s mean %7.0f  median %7.0f  p90 %7.0f cycles per tile' % (nm, a.mean(), np.median(a), np.percentile(a, 90)))
print('  sum per tile %.0f cycles; tiles per workgroup 448 (trunk layers); shader clock %.2f GHz (median workgroup: cycles of its first 10 tiles / their wall time is not recorded, so: per-tile cycles x 448 / median duration)' % (tot, tot * 448 / (np.median(rt1 - rt0) * 10)))
start = (rt0 - rt0.min()) / 1e5
print('  workgroup start times ms: p50 %.2f p90 %.2f max %.2f' % (np.median(start), np.percentile(start, 90), start.max()))
xcc = t[:, 0] & 0xf
print('  workgroups per XCC id:', np.bincount(xcc)[:8])
