"""Emulate operand-precision schemes on the CPU oracle (RRDB-23 x4, one 48x48 image): relative L2 error of the generator output
vs the fp32 oracle when weights / activations are rounded as the MFMA operand scheme would round them."""
import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/explorable-super-resolution_amd')
import torch.nn.functional as F
from oracle import rrdb_oracle as ro
from oracle.weights import fill_formula_weights
import bench
torch.manual_seed(0)
def q(x, dt, terms):
    if terms == 0: return x
    hi = x.to(dt).float()
    if terms == 1: return hi
    return hi + (x - hi).to(dt).float()
orig = F.conv2d
def run(sd, x, wq, xq):
    def conv(inp, w, b=None, *a, **k):
        return orig(xq(inp), wq(w), b, *a, **k)
    F.conv2d = conv; torch.nn.functional.conv2d = conv
    try:
        return ro.rrdb_forward(sd, x, 23, 4, 0, prefix='generated_image_model.model')
    finally:
        F.conv2d = orig
cem, G = bench.build_model('cpu')
for init in ['kaiming0.1', 'formula']:
    if init == 'formula': fill_formula_weights(G, gain=1.0)
    sd = {k: v.detach() for k, v in G.state_dict().items()}
    x = torch.rand(1, 3, 40, 40)
    ref = run(sd, x, lambda w: w, lambda a: a)
    for name, wq, xq in [('bf16x3 (W2,X2)', lambda w: q(w, torch.bfloat16, 2), lambda a: q(a, torch.bfloat16, 2)),
                         ('f16x2 (W1,X2)', lambda w: q(w, torch.float16, 1), lambda a: q(a, torch.float16, 2)),
                         ('f16 (W1,X1)', lambda w: q(w, torch.float16, 1), lambda a: q(a, torch.float16, 1)),
                         ('bf16x2 (W1,X2)', lambda w: q(w, torch.bfloat16, 1), lambda a: q(a, torch.bfloat16, 2)),
                         ('bf16 (W1,X1)', lambda w: q(w, torch.bfloat16, 1), lambda a: q(a, torch.bfloat16, 1))]:
        y = run(sd, x, wq, xq)
        print('%-10s %-16s rel_l2 %.2e  rel_max %.2e' % (init, name, float((y - ref).norm() / ref.norm()), float((y - ref).abs().max() / ref.abs().max())))
