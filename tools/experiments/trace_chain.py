"""(Runs against the tree with tools/experiments/patches/r06_three_experiments.patch applied = commit bc221d6: the experiment lost and is not in the shipping tree.)
The fused dense-block chain (esr_conv3x3_chain, csrc/esr_chain.hip) against its four separate launches at the training-crop shape, and where a
workgroup of the fused launch spends its time (phase stamps of the instrumented build: make -C explorable-super-resolution_amd/csrc trace;
ESR_HIP_LIBRARY=explorable-super-resolution_amd/esr_hip/libesr_hip_trace.so).

    SHAPE=32,52,52 SPLIT=bf16 python tools/experiments/trace_chain.py

Stamps of the chain kernel (wave 0): entry | first copies issued | per pass: seeded, K loop done, next pass set up + its first copies issued, stored.
Printed per pass for the workgroups with the full number of passes (interior tiles: 2 + 2 + 2 + 1 at TH = 7)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd'))
import numpy as np, torch
from esr_hip import _lib, act
dev = 'cuda'
B, H, W = [int(v) for v in os.environ.get("SHAPE", "32,52,52").split(",")]
SPLIT = os.environ.get("SPLIT", "bf16") == "split"
LAT = int(os.environ.get("LAT", "1"))                       # a latent group in front of every layer's input (configs[2]: yes)
torch.manual_seed(0)
buf = act.ActBuf(B, 24, H, W, dev, split=SPLIT)
buf.hi[:, :8, 1:-1, 1:-1].copy_((torch.randn(B, 8, H, W, 8, device=dev) * 0.5).to(torch.bfloat16).view(torch.int16))
z = act.ActBuf(B, 1, H, W, dev, split=SPLIT)
z.hi[:, :, 1:-1, 1:-1].copy_((torch.randn(B, 1, H, W, 8, device=dev) * 0.5).to(torch.bfloat16).view(torch.int16))
packs = []
for i in range(4):
    w = torch.randn(32, 64 + 32 * i + (3 if LAT else 0), 3, 3, device=dev) * 0.05
    packs.append(act.PackedConv(w, torch.randn(32, device=dev) * 0.1, 3 if LAT else 0, split=SPLIT).get())


def run(chains):
    act.CHAINS = chains
    with act.chain():
        for i in range(4):
            act.conv3x3(packs[i], buf.view(0, 8 + 4 * i), B, H, W, 32, in0=z.view() if LAT else None, act_slope=0.2, out=buf.view(8 + 4 * i, 4), reverse=False)


for chains in (True, False, True, False):
    for _ in range(3): run(chains)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [run(chains) for _ in range(20)]; e1.record(); torch.cuda.synchronize()
    print('dense block convs 1-4, %d x %d x %d, %s, latent group %d, %s: %.1f us' % (B, H, W, 'split' if SPLIT else 'bf16', LAT, 'ONE fused launch' if chains else 'four launches', e0.elapsed_time(e1) * 50))
lib = _lib.load_library()
if hasattr(lib, 'esr_debug_trace'):
    nwg = 4096
    tb = torch.zeros(nwg * 128, dtype=torch.int64, device=dev)
    lib.esr_debug_trace.argtypes = [C.c_void_p]; lib.esr_debug_trace.restype = None
    lib.esr_debug_trace(tb.data_ptr())
    run(True); torch.cuda.synchronize()
    lib.esr_debug_trace(None)
    t = tb.cpu().numpy().reshape(nwg, 128).astype(np.int64)
    t = t[t[:, 2] != 0]
    ns = (t[:, 2:126] != 0).sum(1)
    full = t[ns == ns.max()]
    wall = (full[:, 127] - full[:, 126]) / 100.0
    T = full[:, 2:2 + ns.max()]
    clk = np.median((T[:, -1] - T[:, 0]) / wall) / 1e3
    print('%d workgroups, %d with the full %d passes; their stamped span %.2f us (median), launch span %.2f us, shader clock %.2f GHz' % (
        len(t), len(full), (ns.max() - 2) // 4, np.median(wall), (t[:, 127].max() - t[:, 126].min()) / 100.0, clk))
    c = clk * 1e3
    print('  entry -> first copies issued %6.0f cycles %.2f us' % ((T[:, 1] - T[:, 0]).mean(), (T[:, 1] - T[:, 0]).mean() / c))
    names = ['select + seed', 'K loop', 'next pass set-up + copies', 'epilogue stores']
    for p in range((ns.max() - 2) // 4):
        d = [T[:, 2 + 4 * p + k] - T[:, 1 + 4 * p + k] for k in range(4)]
        print('  pass %d: ' % p + '   '.join('%s %5.0f (%.2f us)' % (n, v.mean(), v.mean() / c) for n, v in zip(names, d)))
