"""configs[0] on the GPU: RRDB-3 x4 + CEM (eval) on ONE 32 x 32 frame (G runs on 52 x 52) — every conv launch is a handful of workgroups, the
regime where a launch's fixed cost is all there is.  Per forward: wall time of back-to-back forwards (launch-list replay) and the GPU span
between events; per precision.  Used for the before / after of VERDICT r4 item 1 (ESR_HIP_LIBRARY selects the build)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import bench

cem, G = bench.build_model('cuda', nb=3)
x = torch.rand(1, 3, 32, 32, device='cuda')
for prec in ('split', 'bf16', 'mixed'):
    G.generated_image_model.set_precision(prec)
    with torch.no_grad():
        for _ in range(10):
            G(x)
        torch.cuda.synchronize()
        n = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(n):
            G(x)
        e1.record(); torch.cuda.synchronize()
        print('configs[0] RRDB-3 x4 + CEM, 1 x 32x32, %-5s: %.1f us per forward (wall), %.1f us (GPU span)' % (prec, (time.perf_counter() - t0) / n * 1e6, e0.elapsed_time(e1) / n * 1e3))
