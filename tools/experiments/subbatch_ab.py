"""configs[1] forward (RRDB-23 x4 + CEM, 32 x 128^2, split): the whole batch per layer (351 launches of 1920 tiles) against DEPTH-FIRST
sub-batches (the whole network on n images at a time) whose dense-block working set (23 MB per image: the 24-group hi+lo buffer + the next
block's input groups) fits the 256 MB Infinity Cache.  Same results (images are independent); what changes is where the re-reads of a dense
block's growing prefix come from.  Usage: python tools/experiments/subbatch_ab.py [sub-batch sizes ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd'))
import torch
import bench

dev = torch.device('cuda:0')
cem, G = bench.build_model(dev)
G.generated_image_model.set_precision(os.environ.get('PREC', 'split'))
x = torch.rand(32, 3, 128, 128, device=dev)
subs = [int(a) for a in sys.argv[1:]] or [32, 16, 8, 4]
ref = None
for rep in range(2):
    for nb in subs:
        def step():
            with torch.no_grad():
                return torch.cat([G(x[i:i + nb]) for i in range(0, 32, nb)], 0) if nb < 32 else G(x)
        for _ in range(2):
            y = step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = step()
        e1.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        print('sub-batch %2d: %.2f ms per 32 images   max |diff| vs whole batch %.1e' % (nb, e0.elapsed_time(e1) / 5, float((y - ref).abs().max())), flush=True)
