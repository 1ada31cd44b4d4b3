"""Emulation on the CPU oracle (RRDB-23): 'mixed' plus the dense blocks' INTERMEDIATE activations (the outputs of convs 1-4 of every RDB,
consumed only inside that RDB) stored as one fp16 plane; the trunk / RDB inputs stay fp16 hi+lo (exact here)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd'))
from oracle import rrdb_oracle as ro
from oracle.weights import fill_formula_weights
import bench
P = 'generated_image_model.model'
x = torch.rand(1, 3, 40, 40)
trunk = lambda k: '.1.sub.' in k and 'RDB' in k
orig_rdb = ro._rdb
def rdb_q(sd, pre, x, nc=64):
    outs = [x]
    for i in range(5):
        y = ro._conv(sd, '%s.convs.%d.0' % (pre, i), torch.cat(outs, 1), act=i < 4)
        if i < 4: y = y.half().float()          # c_{i+1} stored as ONE fp16 plane
        outs.append(y)
    return outs[-1] * 0.2 + outs[0][:, -nc:]
for init in ['kaiming0.1', 'kaiming0.3', 'formula']:
    torch.manual_seed(0)
    cem, G = bench.build_model('cpu')
    if init == 'formula': fill_formula_weights(G, gain=1.0)
    if init == 'kaiming0.3':
        import contextlib, io, models.networks as networks
        with contextlib.redirect_stdout(io.StringIO()): networks.init_weights(G, 'kaiming', scale=0.3)
    sd = {k: v.detach() for k, v in G.state_dict().items()}
    ref = ro.rrdb_forward(sd, x, 23, 4, 0, prefix=P)
    sdw = {k: (v.half().float() if (k.endswith('.weight') and trunk(k)) else v) for k, v in sd.items()}
    e_w = float((ro.rrdb_forward(sdw, x, 23, 4, 0, prefix=P) - ref).norm() / ref.norm())
    ro._rdb = rdb_q
    e_a = float((ro.rrdb_forward(sd, x, 23, 4, 0, prefix=P) - ref).norm() / ref.norm())
    e_wa = float((ro.rrdb_forward(sdw, x, 23, 4, 0, prefix=P) - ref).norm() / ref.norm())
    ro._rdb = orig_rdb
    print('%-11s dense-block weights fp16: %.2e | intermediate activations fp16: %.2e | both: %.2e' % (init, e_w, e_a, e_wa))
