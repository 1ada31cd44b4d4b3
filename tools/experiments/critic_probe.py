"""Diagnostic: the HIP critic ('split' / 'bf16') and torch's fp32 module against the SAME module in float64 — logits, d critic / d input,
the WGAN-GP penalty and the parameter gradients of the full D loss (which include the double backward)."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
from oracle.weights import fill_formula_weights, seeded_uniform
import models.modules.architecture as arch
from esr_hip import critic as K

size, batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 4
gain = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
netD = arch.Discriminator_VGG_128(3, 64, input_patch_size=size)
fill_formula_weights(netD, gain=gain)
netD = netD.cuda().train()
net64 = copy.deepcopy(netD).double()
eng = K.CriticEngine(netD, 'split')
real, fake, pt = seeded_uniform((batch, 3, size, size), 11).cuda(), seeded_uniform((batch, 3, size, size), 12).cuda(), seeded_uniform((batch, 1, 1, 1), 13).cuda()


def step(run, net, dt):
    params = list(net.parameters())
    for p in params:
        p.grad = None
    r, f, t = real.to(dt), fake.to(dt), pt.to(dt)
    pr, pf = run(r), run(f)
    interp = (t * f + (1 - t) * r).requires_grad_(True)
    crit = run(interp)
    with K.input_grad_only():
        g = torch.autograd.grad(crit, interp, torch.ones_like(crit), create_graph=True, retain_graph=True)[0]
    gp = 10.0 * ((g.reshape(g.size(0), -1).norm(2, dim=1) - 1) ** 2).mean()
    (pf.mean() - pr.mean() + gp).backward()
    return pr.detach().double(), g.detach().double(), gp.detach().double(), [p.grad.double().clone() for p in params]


rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
ref = step(lambda x: net64(x), net64, torch.float64)
for name, run in (('torch fp32', lambda x: netD(x)), ('hip split', lambda x: K.critic_forward(eng, x)), ('hip bf16', None)):
    if run is None:
        eng.set_precision('bf16')
        run = lambda x: K.critic_forward(eng, x)
    got = step(run, netD, torch.float32)
    scale = max(float(g.norm()) for g in ref[3])
    errs = [float((a - b).norm()) / max(float(b.norm()), 1e-3 * scale) for a, b in zip(got[3], ref[3])]
    print('%-10s logits %.2e  dD/dx %.2e  gp %.6g (ref %.6g, rel %.2e)  param grads: worst %.2e median %.2e' % (
        name, rel(got[0], ref[0]), rel(got[1], ref[1]), float(got[2]), float(ref[2]), abs(float(got[2] - ref[2])) / abs(float(ref[2])), max(errs), sorted(errs)[len(errs) // 2]))
    if name == 'hip split':
        names = [n for n, _ in netD.named_parameters()]
        print('   worst tensors:', sorted(zip(errs, names), reverse=True)[:4])

# ---- how sensitive is the float64 result itself to perturbations of the size of the 16-bit-operand feature error (3e-5)?
torch.manual_seed(1)
for eps in (1e-6, 3e-5):
    netp = copy.deepcopy(net64)
    with torch.no_grad():
        for p in netp.parameters():
            p.mul_(1 + eps * torch.randn_like(p))
    got = step(lambda x: netp(x), netp, torch.float64)
    scale = max(float(g.norm()) for g in ref[3])
    errs = [float((a - b).norm()) / max(float(b.norm()), 1e-3 * scale) for a, b in zip(got[3], ref[3])]
    print('float64 with weights perturbed by %.0e (relative, random): dD/dx %.2e  gp rel %.2e  param grads: worst %.2e median %.2e' % (
        eps, rel(got[1], ref[1]), abs(float(got[2] - ref[2])) / abs(float(ref[2])), max(errs), sorted(errs)[len(errs) // 2]))
