"""Which layers' weight rounding (fp16, single plane) dominates the f16x2 error of RRDB-23?  Emulated on the CPU oracle: weights of the
selected layers are rounded to fp16, everything else stays fp32 (activations hi+lo fp16 = effectively exact)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd'))
from oracle import rrdb_oracle as ro
import bench
torch.manual_seed(0)
cem, G = bench.build_model('cpu')
sd = {k: v.detach() for k, v in G.state_dict().items()}
x = torch.rand(1, 3, 40, 40)
P = 'generated_image_model.model'
ref = ro.rrdb_forward(sd, x, 23, 4, 0, prefix=P)
def run(sel):
    sd2 = {k: (v.half().float() if (k.endswith('.weight') and sel(k)) else v) for k, v in sd.items()}
    y = ro.rrdb_forward(sd2, x, 23, 4, 0, prefix=P)
    return float((y - ref).norm() / ref.norm())
trunk = lambda k: '.1.sub.' in k and 'RDB' in k
sets = {
    'all layers': lambda k: True,
    'RDB convs (345 layers)': trunk,
    'only conv4 (the 64-out conv) of each RDB': lambda k: trunk(k) and '.convs.4.' in k,
    'only convs 0-3 of each RDB': lambda k: trunk(k) and '.convs.4.' not in k,
    'fea + LR_conv + upconvs + HR convs (6 layers)': lambda k: not trunk(k),
    'first 8 RRDBs': lambda k: trunk(k) and int(k.split('.sub.')[1].split('.')[0]) < 8,
    'last 8 RRDBs': lambda k: trunk(k) and int(k.split('.sub.')[1].split('.')[0]) >= 15,
}
for name, sel in sets.items():
    print('%-50s rel_l2 %.2e' % (name, run(sel)))
