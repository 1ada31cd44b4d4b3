import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch, bench
from oracle import rrdb_oracle as ro
from oracle.weights import fill_formula_weights
torch.manual_seed(0)
cem, G = bench.build_model('cuda')
for init in ['kaiming0.1', 'formula']:
    if init == 'formula': fill_formula_weights(G, gain=1.0)
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    x = torch.rand(1, 3, 40, 40)
    ref = ro.rrdb_forward(sd, x, 23, 4, 0, prefix='generated_image_model.model')
    net = G.generated_image_model
    for prec in ['split', 'mixed', 'f16x2', 'f16', 'bf16']:
        net.set_precision(prec)
        with torch.no_grad():
            y = net(x.cuda()).cpu()
        print('%-10s %-6s generator output rel_l2 vs fp32 CPU oracle %.2e  rel_max %.2e' % (init, prec, float((y - ref).norm() / ref.norm()), float((y - ref).abs().max() / ref.abs().max())))
