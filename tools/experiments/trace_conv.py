import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd'))
import numpy as np, torch
from esr_hip import _lib, act
dev = 'cuda'
B, H, W = [int(v) for v in os.environ.get("SHAPE", "32,148,148").split(",")]
SPLIT = os.environ.get("SPLIT", "split") == "split"          # split (bf16 hi+lo, 3 MFMAs) | bf16 (one plane)
cin, cout = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(0)
buf = act.ActBuf(B, 24, H, W, dev, split=SPLIT)
buf.hi[:, :, 1:-1, 1:-1].copy_((torch.randn(B, 24, H, W, 8, device=dev) * 0.5).to(torch.bfloat16).view(torch.int16))
if buf.lo is not None: buf.lo[:, :, 1:-1, 1:-1].copy_((torch.randn(B, 24, H, W, 8, device=dev) * 0.002).to(torch.bfloat16).view(torch.int16))
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
bias = torch.randn(cout, device=dev) * 0.1
pc = act.PackedConv(w, bias, 0, split=SPLIT).get()
obuf = act.ActBuf(B, 8, H, W, dev, split=SPLIT)
out = buf.view(cin // 8, (cout + 7) // 8) if cin + cout <= 192 else obuf.view()
res1 = buf.view(0, 8) if cout == 64 else None
def run():
    act.conv3x3(pc, buf.view(0, cin // 8), B, H, W, cout, act_slope=0.2 if cout == 32 else 1.0, alpha=1.0 if cout == 32 else 0.2,
                res1=res1, beta1=1.0, out=out)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); [run() for _ in range(10)]; e1.record(); torch.cuda.synchronize()
print('cin %d cout %d: %.1f us per launch' % (cin, cout, e0.elapsed_time(e1) * 100))
lib = _lib.load_library()
if hasattr(lib, 'esr_debug_trace'):
    nwg = 4096
    tb = torch.zeros(nwg * 128, dtype=torch.int64, device=dev)
    lib.esr_debug_trace.argtypes = [C.c_void_p]; lib.esr_debug_trace.restype = None
    lib.esr_debug_trace(tb.data_ptr())
    run(); torch.cuda.synchronize()
    lib.esr_debug_trace(None)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    np.save(os.path.join(ROOT, 'gpurun_out', 'trace_%d_%d.npy' % (cin, cout)), tb.cpu().numpy().reshape(nwg, 128))
