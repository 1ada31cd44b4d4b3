import os, sys, ctypes, json
os.environ['ESR_CONV_DBG'] = '3'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch, numpy as np
import bench
from esr_hip import _lib
dev = torch.device('cuda', 0)
cem, G = bench.build_model(dev)
x = torch.rand(32, 3, 128, 128, device=dev)
h = _lib.load_library()
h.esr_debug_read_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros((256, 8), dtype=np.uint64)
with torch.no_grad():
    G(x); torch.cuda.synchronize()
    h.esr_debug_read_prof(buf.ctypes.data, 1)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); G(x); t1.record(); torch.cuda.synchronize()
    h.esr_debug_read_prof(buf.ctypes.data, 1)
print('forward ms', t0.elapsed_time(t1))
m = buf.astype(np.float64).mean(0)
print('per-WG mean ticks: wait %.0f  taps %.0f  epilogue %.0f | steps %.0f tiles %.0f' % (m[0], m[1], m[2], m[3], m[4]))
tot = m[0] + m[1] + m[2]
print('fractions: wait %.3f taps %.3f epilogue %.3f ; total ticks %.0f' % (m[0] / tot, m[1] / tot, m[2] / tot, tot))
print('ticks per step: wait %.1f taps %.1f ; per tile: epilogue %.1f' % (m[0] / m[3], m[1] / m[3], m[2] / m[4]))
