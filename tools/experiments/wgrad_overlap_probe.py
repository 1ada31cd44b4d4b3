"""Feasibility probe (TIMING ONLY — the overlapped variant's gradients are stale): can the generator's batched weight-gradient launch run on a
second stream UNDER the data-gradient chain of the configs[2] backward (276 + 69 small launches, one workgroup per CU, latency-bound) instead of
behind it?  The weight-gradient launch of the previous step's tables is enqueued on a side stream when the backward starts and its command is
taken out of the recorded launch list; the main stream joins the side stream at the end.

    python tools/experiments/wgrad_overlap_probe.py [conv LDS stages of the small launches: 0 (shipping) | 1]
    ESR_WGRAD_LDS=<KB>  (scratch build of esr_bwd.hip, see README): LDS request of a weight-gradient workgroup, to cap its residency per CU

Co-residency on a CU needs LDS for both: a two-stage data-gradient workgroup holds 98 KB, a weight-gradient workgroup 80 KB (its end-of-tile
reduction) — they only fit together with one-stage small launches (49 KB)."""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import CEM.CEMnet as CEMnet
import models.modules.architecture as arch
import models.networks as networks
from esr_hip import act as A, _lib

stages = int(sys.argv[1]) if len(sys.argv) > 1 else 0
A.LDS_STAGES = stages
dev, B, lat, h = 'cuda', 32, 3, 52
torch.manual_seed(0)
cem = CEMnet.CEMnet(CEMnet.Get_CEM_Conf(4))
net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=23, gc=32, upscale=4, norm_type=None, act_type='leakyrelu', mode='CNA', upsample_mode='upconv',
                   latent_input='all_layers_HR_downscaled', num_latent_channels=lat)
G = cem.WrapArchitecture_PyTorch(net)
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    networks.init_weights(G, init_type='kaiming', scale=0.1)
G = G.to(dev).train()
net.set_precision('bf16')
x = torch.rand(B, 3, h, h, device=dev); z = torch.rand(B, lat, 4 * h, 4 * h, device=dev) * 2 - 1
tgt = torch.rand(B, 3, 4 * h, 4 * h, device=dev)
inp = torch.cat([z.view(B, lat * 16, h, h), x], 1)
side = torch.cuda.Stream()
hook = [None]


def step():
    for p in G.parameters():
        p.grad = None
    y = G(inp)
    loss = (y - tgt).abs().mean()
    if hook[0]:
        hook[0]()
    loss.backward()
    if hook[0]:
        torch.cuda.current_stream().wait_stream(side)
    return loss


def timed(n=10):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


base = timed()
eng = net.engine
entry = [v for bufs in eng._bufs.values() for k, v in bufs['_plans'].items() if isinstance(v, tuple)]
assert len(entry) == 1, len(entry)
plan, wg = entry[0]
tables = wg._tables
dummy = torch.zeros(64, dtype=torch.uint8, device=dev)
found = []
for item in plan.items:
    if callable(item):
        continue
    for i in range(len(item)):
        if item[i].op == _lib.OP_WGRAD_BATCH_RUN:
            found.append((item, i))
print('shipping order (small-launch LDS stages %d): fwd + bwd %.2f ms; %d weight-gradient command(s) in the backward list of %d commands' % (stages, base, len(found), plan.n_cmds))

# (1) backward without the weight-gradient launch at all: the floor of any overlap
saved = []
for item, i in found:
    saved.append((item[i].u.wgrad_batch_run.workspace, C.string_at(C.addressof(item[i].u), C.sizeof(_lib.CmdWgradBatchRun))))
    item[i].op = _lib.OP_ZERO
    item[i].u.zero.p = dummy.data_ptr()
    item[i].u.zero.n16 = 1
print('  without the weight-gradient launch:       %.2f ms' % timed())


# (2) the launch on the side stream, enqueued when the backward starts (previous step's tables)
def side_launch():
    side.wait_stream(torch.cuda.current_stream())
    for arr, ws, p in tables:
        _lib.check(_lib.lib.esr_conv3x3_wgrad_batch_run(ws.data_ptr(), C.byref(p), side.cuda_stream), 'esr_conv3x3_wgrad_batch_run')


hook[0] = side_launch
print('  weight-gradient launch on a side stream:  %.2f ms   (ESR_WGRAD_LDS=%s)' % (timed(), os.environ.get('ESR_WGRAD_LDS', '-')))
# (3) the side launch alone (backward list still without it), to see its own duration at that residency
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(side):
    e0.record()
    for _ in range(5):
        for arr, ws, p in tables:
            _lib.lib.esr_conv3x3_wgrad_batch_run(ws.data_ptr(), C.byref(p), side.cuda_stream)
    e1.record()
torch.cuda.synchronize()
print('  the weight-gradient launch alone:         %.2f ms' % (e0.elapsed_time(e1) / 5))
