"""Emulate the 'f16f8' operand scheme on the CPU oracle (RRDB-23 x4, one small image) before building it:
   product  W*X  ~=  Whi*Xhi  (fp16 x fp16)  +  Whi8*Xlo8  +  Wlo8*Xhi8   (fp8 e4m3, fixed power-of-two scales)
   storage  X   =  fp16(X) + e4m3((X - fp16(X)) * 2^14) / 2^14            (16 significant bits)
Prints the relative error of the generator output vs the fp32 oracle next to split-bf16's, over weight scales."""
import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/explorable-super-resolution_amd')
import torch.nn.functional as F
from oracle import rrdb_oracle as ro
from oracle.weights import fill_formula_weights
import bench
torch.manual_seed(0)
F8 = torch.float8_e4m3fn
BX, SX, AW, TW = 3, 14, 7, 18          # scales: Xhi8 = e4m3(X*2^BX), Xlo8 = e4m3(Xlo*2^SX), Whi8 = e4m3(W*2^AW), Wlo8 = e4m3(Wlo*2^TW); BX+TW == SX+AW
def f8(x, s):
    return (x * 2.0 ** s).clamp(-448, 448).to(F8).float() * 2.0 ** -s
def bf2(x):
    hi = x.to(torch.bfloat16).float()
    return hi + (x - hi).to(torch.bfloat16).float()
orig = F.conv2d
def conv_f16f8(inp, w, b=None, *a, **k):
    xh = inp.to(torch.float16).float(); xl = inp - xh
    wh = w.to(torch.float16).float(); wl = w - wh
    y = orig(xh.double(), wh.double(), None, *a, **k) + orig(f8(xl, SX).double(), f8(w, AW).double(), None, *a, **k) + orig(f8(inp, BX).double(), f8(wl, TW).double(), None, *a, **k)
    y = y.float()
    return y + b.view(1, -1, 1, 1) if b is not None else y
def conv_bf16x3(inp, w, b=None, *a, **k):
    xh = inp.to(torch.bfloat16).float(); xl = (inp - xh).to(torch.bfloat16).float()
    wh = w.to(torch.bfloat16).float(); wl = (w - wh).to(torch.bfloat16).float()
    y = (orig(xh.double(), wh.double(), None, *a, **k) + orig(xl.double(), wh.double(), None, *a, **k) + orig(xh.double(), wl.double(), None, *a, **k)).float()
    return y + b.view(1, -1, 1, 1) if b is not None else y
def store_f16f8(x):
    h = x.to(torch.float16).float()
    return h + f8(x - h, SX)
def run(sd, x, conv, store, nb=23):
    o_conv, o_rdb, o_rrdb = ro._conv, ro._rdb, ro._rrdb
    F.conv2d = conv
    ro._conv = lambda *a, **k: store(o_conv(*a, **k))
    ro._rdb = lambda *a, **k: store(o_rdb(*a, **k))
    ro._rrdb = lambda *a, **k: store(o_rrdb(*a, **k))
    try:
        return ro.rrdb_forward(sd, x, nb, 4, 0, prefix='generated_image_model.model')
    finally:
        F.conv2d = orig; ro._conv, ro._rdb, ro._rrdb = o_conv, o_rdb, o_rrdb
cem, G = bench.build_model('cpu')
x = torch.rand(1, 3, 40, 40)
sd0 = {k: v.detach().clone() for k, v in G.state_dict().items()}
cases = [('kaiming x0.1', sd0)]
for gain in (0.1, 0.6, 1.0, 2.0):
    fill_formula_weights(G, gain=gain)
    cases.append(('formula gain %.1f' % gain, {k: v.detach().clone() for k, v in G.state_dict().items()}))
# heavy-tailed: kaiming x0.1 with 1 % of the weights multiplied by 30
sdh = {k: v.clone() for k, v in sd0.items()}
g = torch.Generator().manual_seed(5)
for k, v in sdh.items():
    if k.endswith('.weight') and v.dim() == 4 and 'generated_image_model' in k:
        m = torch.rand(v.shape, generator=g) < 0.01
        v[m] *= 30.0
cases.append(('heavy-tailed', sdh))
for name, sd in cases:
    ref = run(sd, x, orig, lambda t: t)
    for sname, conv, store in [('bf16x3', conv_bf16x3, bf2), ('f16f8', conv_f16f8, store_f16f8)]:
        y = run(sd, x, conv, store)
        print('%-18s %-8s rel_l2 %.2e  rel_max %.2e   (out absmax %.3g)' % (name, sname, float((y - ref).norm() / ref.norm()), float((y - ref).abs().max() / ref.abs().max()), float(ref.abs().max())), flush=True)
