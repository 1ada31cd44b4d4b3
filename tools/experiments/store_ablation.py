"""What do the conv epilogue's stores cost?  -DESR_ABL_NOSTORE (timing only) skips them, but then every activation buffer stays at its initial
zeros and all-zero MFMA operands draw far less power (the clock rises: mfma_peak.hip shows 2.4 vs 1.8 PFLOP/s for constant vs random operands), which would
be credited to the stores.  This script pre-fills every activation buffer of the engine with random bf16 / fp16 patterns of realistic magnitude
before timing, so that the no-store build multiplies realistic operands.  Run it with the normal library and with the ablation build
(ESR_HIP_LIBRARY=...) on the same GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import bench

prec = sys.argv[1] if len(sys.argv) > 1 else 'split'
cem, G = bench.build_model('cuda')
net = G.generated_image_model
net.set_precision(prec)
x = torch.rand(32, 3, 128, 128, device='cuda')
with torch.no_grad():
    for _ in range(3):
        G(x)


def fill(buf):
    dt = torch.bfloat16 if buf.fmt == 0 else torch.float16
    shape = buf.hi.shape
    interior = torch.zeros(shape, dtype=dt, device=buf.hi.device)
    interior[:, :, 1:-1, 1:-1, :] = (torch.rand(shape[0], shape[1], shape[2] - 2, shape[3] - 2, shape[4], device=buf.hi.device) * 2 - 1).to(dt)
    buf.hi.copy_(interior.view(torch.int16))
    if buf.lo is not None:
        interior[:, :, 1:-1, 1:-1, :] = ((torch.rand(shape[0], shape[1], shape[2] - 2, shape[3] - 2, shape[4], device=buf.hi.device) * 2 - 1) * 0.003).to(dt)
        buf.lo.copy_(interior.view(torch.int16))


for d in net.engine._bufs.values():
    for k, v in d.items():
        for b in (v if isinstance(v, list) else [v]):
            if hasattr(b, 'hi'):
                fill(b)
torch.cuda.synchronize()
with torch.no_grad():
    G(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        G(x)
    torch.cuda.synchronize()
print('%s lib=%s: %.2f ms per forward (buffers pre-filled with random activations)' % (prec, os.environ.get('ESR_HIP_LIBRARY', 'default'), (time.perf_counter() - t0) * 100))
