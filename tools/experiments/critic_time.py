"""GPU time of the critic's passes at the configs[2] shape (32 x 3 x 128 x 128): forward, forward + backward (parameters), and the whole
WGAN-GP critic step (three forwards, penalty with double backward, backward), HIP engine vs the nn.Module on MIOpen (bf16 autocast).
    python tools/experiments/critic_time.py [bf16|split] [batch]"""
import contextlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import models.modules.architecture as arch
import models.networks as networks
from esr_hip import critic as K

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(0)
netD = arch.Discriminator_VGG_128(3, 64).cuda().train()
eng = K.CriticEngine(netD, prec)
real, fake = torch.rand(B, 3, 128, 128, device='cuda'), torch.rand(B, 3, 128, 128, device='cuda')
pt = torch.rand(B, 1, 1, 1, device='cuda')
params = list(netD.parameters())


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3


def passes(run, grouped=False):
    def fwd():
        with torch.no_grad():
            run(real)

    def fwd_bwd():
        for p in params:
            p.grad = None
        run(real).mean().backward()

    def d_step():
        for p in params:
            p.grad = None
        pr, pf = run(real), run(fake)
        interp = (pt * fake + (1 - pt) * real).requires_grad_(True)
        crit = run(interp)
        with K.input_grad_only():
            g = torch.autograd.grad(crit, interp, torch.ones_like(crit), create_graph=True, retain_graph=True)[0]
        gp = 10.0 * ((g.reshape(B, -1).norm(2, dim=1) - 1) ** 2).mean()
        (pf.mean() - pr.mean() + gp).backward()
    def d_step_grouped():          # the three calls as one grouped pass (what SRRaGANModel.optimize_parameters does on the HIP engine)
        for p in params:
            p.grad = None
        interp = (pt * fake + (1 - pt) * real).requires_grad_(True)
        pr, pf, crit = K.critic_forward_group(eng, [real, fake, interp])
        with K.input_grad_only(group=2):
            g = torch.autograd.grad(crit, interp, torch.ones_like(crit), create_graph=True, retain_graph=True)[0]
        gp = 10.0 * ((g.reshape(B, -1).norm(2, dim=1) - 1) ** 2).mean()
        (pf.mean() - pr.mean() + gp).backward()
    out = [('forward', fwd), ('forward + backward', fwd_bwd), ('WGAN-GP critic step', d_step)]
    if grouped:
        out.append(('critic step, grouped', d_step_grouped))
    return out


def stock(x):
    ctx = torch.autocast('cuda', dtype=torch.bfloat16) if prec == 'bf16' else contextlib.nullcontext()
    with ctx:
        return netD(x).float()


variants = [('hip %s' % prec, lambda x: K.critic_forward(eng, x)), ('torch/MIOpen %s' % ('bf16 autocast' if prec == 'bf16' else 'fp32'), stock)]
if os.environ.get('CRITIC_ONLY') == 'hip':
    variants = variants[:1]
if os.environ.get('CRITIC_ONLY') == 'step':          # profiling runs: the critic step alone, 10 + 3 iterations
    variants = []
    timed(dict(passes(lambda x: K.critic_forward(eng, x), True))['critic step, grouped' if os.environ.get('CRITIC_GROUPED', '1') != '0' else 'WGAN-GP critic step'])
for name, run in variants:
    for what, fn in passes(run, name.startswith('hip')):
        wall, host = timed(fn)
        print('%-28s %-22s %7.2f ms   (host enqueue %6.2f ms)' % (name, what, wall, host))

if os.environ.get('CRITIC_PROFILE'):
    import cProfile, io, pstats
    run = lambda x: K.critic_forward(eng, x)
    fn = dict(passes(run))['WGAN-GP critic step']
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
    print('\n'.join(l[:170] for l in s.getvalue().splitlines()[4:44]))
