"""Upper bound of VERDICT r4 item 8 (BatchNorm statistics gathered in the producing conv's epilogue instead of bn_reduce_kernel<0>):
the grouped WGAN-GP critic step at the configs[2] shape (3 x 32 x 3 x 128 x 128, bf16) as shipped, and with the forward statistics
launches REMOVED from the recorded launch lists — (a) the reductions only, (b) the reductions and bn_finalize.  The ablated passes normalise
with the sums / statistics the last complete pass left in the set's buffers (their forward does not zero them; same values every step:
finite, realistic activations — zeroed statistics would blank every activation, and this part runs faster on blank data); nothing is
added to the conv epilogues, so the difference is the most an epilogue form could gain before paying for its own cross-lane sums.
    python tools/experiments/bn_stats_ablation.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd')); sys.path.insert(0, ROOT)
import torch
import models.modules.architecture as arch
from esr_hip import critic as K, _lib, act as A

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = 32
torch.manual_seed(0)
netD = arch.Discriminator_VGG_128(3, 64).cuda().train()
eng = K.CriticEngine(netD, 'bf16')
real, fake = torch.rand(B, 3, 128, 128, device='cuda'), torch.rand(B, 3, 128, 128, device='cuda')
pt = torch.rand(B, 1, 1, 1, device='cuda')
params = list(netD.parameters())


def d_step():
    for p in params:
        p.grad = None
    interp = (pt * fake + (1 - pt) * real).requires_grad_(True)
    pr, pf, crit = K.critic_forward_group(eng, [real, fake, interp])
    with K.input_grad_only(group=2):
        g = torch.autograd.grad(crit, interp, torch.ones_like(crit), create_graph=True, retain_graph=True)[0]
    gp = 10.0 * ((g.reshape(B, -1).norm(2, dim=1) - 1) ** 2).mean()
    (pf.mean() - pr.mean() + gp).backward()


def forward_only():
    with torch.no_grad():
        K.critic_forward_group(eng, [real, fake, real])


def timed(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def drop_plans():
    for sets in eng._free_sets.values():
        for bs in sets:
            bs.plans = {}


emit_bn, emit, zero_region = K._emit_bn, A.Recorder.emit, K._zero_region
dropped = [0]


def variant(skip_reduce, skip_finalize, fuse=True):
    K.FUSE_FINALIZE = fuse if not skip_finalize else False
    def _emit_bn(rec, op, d, mode, sums=None):
        if skip_reduce and op == _lib.OP_BN_REDUCE and mode == 0:
            dropped[0] += 1
            return
        emit_bn(rec, op, d, mode, sums)

    def _emit(self, op, st, refs=()):
        if skip_finalize and op == _lib.OP_BN_FINALIZE:
            dropped[0] += 1
            return
        return emit(self, op, st, refs)

    def _zero(rec, bs, region):
        # the ablated forward must not wipe the sums / statistics it re-uses: it zeroes the backward's sums instead (same launch, same size class)
        return zero_region(rec, bs, bs.bwd_zero if (skip_reduce and region == bs.fwd_zero) else region)
    K._emit_bn, A.Recorder.emit, K._zero_region = _emit_bn, _emit, _zero
    drop_plans()


if os.environ.get('BN_ABL_ONLY'):            # profiling runs: one variant, 4 + 20 steps
    v = int(os.environ['BN_ABL_ONLY'])
    variant(False, False); d_step(); torch.cuda.synchronize()
    variant(v >= 1, v >= 2, v != 3)
    print('variant %d: step %.3f ms' % (v, timed(d_step)))
    sys.exit(0)
VARIANTS = (('reduce, finalize, apply (three launches)', False, False, False), ('shipped: reduce, finalize+apply', False, False, True),
            ('no bn_reduce<0> (ablation)', True, False, True), ('no bn_reduce<0>, no finalize (ablation)', True, True, False))
rows = []
for r in range(reps):
    row = []
    for name, sr, sf, fu in VARIANTS:
        variant(False, False); d_step(); forward_only(); torch.cuda.synchronize()        # a complete pass leaves real statistics behind
        dropped[0] = 0
        variant(sr, sf, fu)
        row.append((name, timed(d_step), timed(forward_only), dropped[0]))
    rows.append(row)
    print('rep %d: ' % r + '   '.join('%s: step %.3f ms, forward %.3f ms (%d launches dropped from the lists)' % x for x in row), flush=True)
print('median over %d reps:' % reps)
for j in range(len(VARIANTS)):
    st = sorted(r[j][1] for r in rows)[reps // 2]; fw = sorted(r[j][2] for r in rows)[reps // 2]
    print('  %-48s critic step %.3f ms   grouped forward %.3f ms' % (rows[0][j][0], st, fw))
