"""TEST INFRASTRUCTURE ONLY.  The activation-pattern instrument of the backward parity checks (tests/test_gpu_backward.py,
__graft_entry__.smoke()).

A LeakyReLU network (reference codes/models/modules/block.py:18, used by every conv_block of RRDBNet, architecture.py:228-302) is piecewise
linear: its gradient is a function of the activation PATTERN — which side of zero every pre-activation fell on.  Two correct forwards that
differ by rounding can disagree on the pattern at pre-activations within rounding distance of zero, and then their gradients differ by
O(1e-3) inside those activations' receptive fields: a plain gradient comparison cannot tell such a flip from a bug.  This module removes the
ambiguity: it reads the pattern the HIP forward actually took (the sign of its STORED activations), forces it on the fp64 oracle
(rrdb_oracle._lrelu is a module-level hook for exactly this), and what is left between the two backward passes is arithmetic error alone —
which has to meet the plain relative-L2 bar.
"""
import torch

from . import rrdb_oracle as ro


def rrdb_node(t):
    """The autograd node of the generator (esr_hip.autograd._RRDBFunction) behind tensor `t`: it is the Function's ctx, so it carries `.bufs`
    (the activation buffers the forward kept) — lets a check that went through the module API (CEM wrapper, autograd) read the pattern."""
    seen, todo = set(), [t.grad_fn]
    while todo:
        n = todo.pop()
        if n is None or n in seen:
            continue
        seen.add(n)
        if '_RRDBFunction' in type(n).__name__:
            return n
        todo.extend(f for f, _ in n.next_functions)
    raise LookupError('no generator node behind this tensor')


def stored_lrelu_outputs(bufs, nb, nf=64):
    """The HIP forward's stored LeakyReLU outputs (fp32 NCHW on the CPU) in the ORACLE's call order of rrdb_oracle._lrelu: convs 0-3 of every
    RDB, the upconvs, HR_conv0.  `bufs`: RRDBEngine.run_forward(..., keep=True)[1]; nf: the generator's stream width."""
    out = []
    for j in range(3 * nb):
        for i in range(4):
            out.append(bufs['rdb'][j].to_nchw(32, cg0=nf // 8 + 4 * i).cpu())
    return out + [b.to_nchw(nf).cpu() for b in bufs['ups']] + [bufs['hr0'].to_nchw(nf).cpu()]


class capture_preactivations:
    """with capture_preactivations() as pre: <oracle forward>  ->  pre = the pre-activation of every LeakyReLU, in call order."""

    def __enter__(self):
        self.pre, self.orig = [], ro._lrelu
        ro._lrelu = lambda y: (self.pre.append(y.detach()), self.orig(y))[1]
        return self.pre

    def __exit__(self, *exc):
        ro._lrelu = self.orig


class forced_pattern:
    """with forced_pattern(stored): <oracle forward + backward>  — every LeakyReLU takes the branch the HIP forward took at that element
    (stored > 0: identity, else slope 0.2; torch's leaky_relu'(0) = slope)."""

    def __init__(self, stored):
        self.stored = stored

    def __enter__(self):
        self.orig = ro._lrelu
        it = iter(self.stored)
        ro._lrelu = lambda y: torch.where(next(it).to(y.device) > 0, y, 0.2 * y)
        return self

    def __exit__(self, *exc):
        ro._lrelu = self.orig


def pattern_flips(pre, stored):
    """(number of elements where the fp64 pattern and the stored pattern disagree, the largest |fp64 pre-activation| / layer rms among them)."""
    assert len(pre) == len(stored), (len(pre), len(stored))
    flips, worst = 0, 0.0
    for p64, s in zip(pre, stored):
        assert p64.shape == s.shape, (p64.shape, s.shape)
        differ = (p64 > 0) != (s > 0)
        n = int(differ.sum())
        flips += n
        if n:
            worst = max(worst, float(p64[differ].abs().max()) / float(p64.double().pow(2).mean().sqrt()))
    return flips, worst


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))
