"""(Runs against the tree with tools/experiments/patches/r06_three_experiments.patch applied = commit bc221d6: the experiment lost and is not in the shipping tree.)
Same-process A/B of the critic's real-image forward under the generator's forward (SRRaGANModel.overlap_D_real -> esr_hip.critic.critic_prefetch_first;
VERDICT r5 item 3) on the configs[2] G + D training step at its per-GPU shape (bench.run_c3's model and data: 32 crops of 52 x 52, bf16), alternating,
with the step's phase times (GPU time between events on the main stream: with the overlap the real third of the critic's forward leaves 'D_step'
and shows — as far as it delays the generator's launches — in 'G_forward').

    python tools/experiments/critic_overlap_ab.py [repeats]
"""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import bench_paths, models

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda')
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


from esr_hip import critic as K
for r in range(REP):
    for overlap in (True, 'main', False):          # 'main': the split schedule on ONE stream (what the split itself costs, without the concurrency)
        model.overlap_D_real = bool(overlap)
        K.PREFETCH_ON_MAIN = overlap == 'main'
        steps(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps(30)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
        model.timing = {}
        steps(5)
        ph = {k: round(v / 5, 2) for k, v in model.timing.items()}
        model.timing = None
        print('configs[2] step, critic real pass under the generator forward %-5s: %.2f ms per step   phases %s' % (overlap, ms, ph), flush=True)
