"""(Runs against the tree with tools/experiments/patches/r06_three_experiments.patch applied = commit bc221d6: the experiment lost and is not in the shipping tree.)
Same-process A/B of the fused dense-block chains (esr_conv3x3_chain, csrc/esr_chain.hip): esr_hip.act.CHAINS on / off,
alternating, on (a) the configs[2] G + D training step at its per-GPU shape (bench.run_c3's model and data: 32 crops of 52 x 52, bf16),
with the step's phase times, and (b) configs[0] on the GPU (RRDB-3 x4 + CEM on one 32 x 32 frame: tools/experiments/c1_latency.py) per precision.

    python tools/experiments/chain_ab.py [repeats]
"""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import bench, bench_paths, models
from esr_hip import act as A

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda')

# ---- (a) configs[2]
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = models.create_model(bench_paths.make_opt(True, with_D=True))
model.netG.generated_image_model.set_precision('bf16')
model.D_dtype = torch.bfloat16
g = torch.Generator().manual_seed(2000)
data = {'LR': torch.rand(32, 3, 52, 52, generator=g).to(dev), 'HR': torch.rand(32, 3, 208, 208, generator=g).to(dev),
        'Z': (torch.rand(32, 3, 208, 208, generator=g) * 2 - 1).to(dev)}


def steps(n):
    for _ in range(n):
        model.feed_data(data); model.optimize_parameters()


for r in range(REP):
    for chains in (True, False):
        A.CHAINS = chains
        A.CHAIN_LOG = log = []
        steps(3)                                  # (the first toggle records the launch lists of this form)
        A.CHAIN_LOG = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps(10)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        model.timing = {}
        steps(5)
        ph = {k: round(v / 5, 2) for k, v in model.timing.items()}
        model.timing = None
        print('configs[2] step, chains %-5s: %.2f ms per step   phases %s   (blocks recorded while warming up: %d, fused %d)' % (
            chains, ms, ph, len(log), sum(v == 1 for v in log)), flush=True)

# ---- (b) configs[0] on the GPU
del model
cem, G = bench.build_model('cuda', nb=3)
x = torch.rand(1, 3, 32, 32, device='cuda')
for prec in ('split', 'bf16'):
    G.generated_image_model.set_precision(prec)
    for r in range(REP):
        for chains in (True, False):
            A.CHAINS = chains
            with torch.no_grad():
                for _ in range(10):
                    G(x)
                torch.cuda.synchronize()
                n = 200
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    G(x)
                e1.record(); torch.cuda.synchronize()
            print('configs[0] RRDB-3 x4 + CEM, 1 x 32x32, %-5s, chains %-5s: %.1f us per forward (GPU span)' % (prec, chains, e0.elapsed_time(e1) / n * 1e3), flush=True)
