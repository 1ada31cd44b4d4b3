"""configs[1] forward, split: depth-first SUB-BATCHES ON SEVERAL STREAMS.  n_streams model replicas (own activation buffers, same weights), each
runs the whole network on `sub` images at a time on its own stream; sub x n_streams images are in flight (23 MB of dense-block working set per
image against the 256 MB Infinity Cache) and a stream's launch ramps / tails overlap with the other streams' work.  Against the whole batch
per layer on one stream.  Usage: python tools/experiments/substream_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd'))
import torch
import bench
from esr_hip import act as A
A.LDS_STAGES = int(os.environ.get('STAGES', '0'))      # 1: every launch in the single-stage, two-workgroups-per-CU form (sub-batches of < 256 tiles from two streams then share the CUs)

dev = torch.device('cuda:0')
NS = 4
models = []
for i in range(NS):
    cem, G = bench.build_model(dev)
    G.generated_image_model.set_precision('split')
    models.append(G)
streams = [torch.cuda.Stream() for _ in range(NS)]
x = torch.rand(32, 3, 128, 128, device=dev)


def whole():
    with torch.no_grad():
        return models[0](x)


def split(sub, ns):
    outs = [None] * (32 // sub)
    cur = torch.cuda.current_stream()
    for s in streams[:ns]:
        s.wait_stream(cur)
    with torch.no_grad():
        for j, i in enumerate(range(0, 32, sub)):
            k = j % ns
            with torch.cuda.stream(streams[k]):
                outs[j] = models[k](x[i:i + sub])
    for s in streams[:ns]:
        cur.wait_stream(s)
    return torch.cat(outs, 0)


ref = whole()
for rep in range(2):
    for name, fn in [('whole batch, 1 stream', whole)] + [('sub %d x %d streams' % (sub, ns), (lambda sub=sub, ns=ns: split(sub, ns)))
                                                         for sub, ns in ((8, 2), (4, 2), (4, 3), (4, 4), (2, 4))]:
        for _ in range(2):
            y = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = fn()
        e1.record()
        torch.cuda.synchronize()
        print('%-26s %.2f ms per 32 images   max |diff| %.1e' % (name, e0.elapsed_time(e1) / 5, float((y - ref).abs().max())), flush=True)
