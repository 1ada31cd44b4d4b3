"""GPU tests (-m gpu) at the FULL per-GPU sizes of BASELINE.json's configs[2] and configs[3], through the product's model wrapper:
  configs[3]  Z_optimizer on 64 Z samples of 512x512 (one 128x128 LR image, RRDB-23 x4, latent 3): 3 iterations
  configs[2]  SRRaGANModel.optimize_parameters() on 32 crops of 52x52 (HR 208x208, latent 3, RRDB-23): the generator step in bf16 against the
              fp32-class split precision, and one whole G + D (Discriminator_VGG_128, WGAN-GP) step with the bf16 critic against fp32
The CPU oracle is far too slow at these sizes; the checks are size-independent properties and consistency between precisions, each with
its stated tolerance."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _model(is_train, with_D=False, seed=0):
    import bench_paths
    import models
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        return models.create_model(bench_paths.make_opt(is_train, with_D=with_D))


def test_configs3_z_search_at_full_size_and_its_shard():
    """64 Z samples of 512x512, objective STD_increase, 3 Adam iterations: every activation of the 69 dense blocks is kept for the backward
    (nothing recomputed), so this is also the memory test of the 288 GB part.  Then the shard a rank of an 8-GPU run would own (samples 0..7,
    same 1/64 loss scale) is run alone and must reproduce those samples BIT-exactly (independent samples, no data-path collective)."""
    import Z_optimization
    from Z_optimization import Z_optimizer
    m = _model(False)
    B = 64
    g = torch.Generator().manual_seed(11)
    lr = torch.rand(1, 3, 128, 128, generator=g).to(DEV)
    z0 = (torch.rand(B, 3, 512, 512, generator=g) * 0.2 - 0.1).to(DEV)

    def search(shard):
        lo, hi = shard
        real = Z_optimization.esr_dist.shard_range
        Z_optimization.esr_dist.shard_range = lambda n, r=None, w=None: (lo, hi)
        try:
            m.feed_data({'LR': lr.expand(hi - lo, -1, -1, -1), 'Z': z0[lo:hi].clone()}, need_GT=False)
            m.test()
            zo = Z_optimizer(objective='STD_increase', Z_size=[512, 512], model=m, Z_range=1, max_iters=3, data={'LR': lr, 'STD_increment': 0.01},
                             initial_Z=z0.clone(), initial_LR=0.1, batch_size=B)
            return zo.optimize(), list(zo.loss_values)
        finally:
            Z_optimization.esr_dist.shard_range = real
    torch.cuda.reset_peak_memory_stats()
    z_full, losses = search((0, B))
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    assert z_full.shape == (B, 3, 512, 512) and bool(torch.isfinite(z_full).all())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert float((z_full - z0).abs().max()) > 5e-3                      # it moved (the search keeps its min-loss iterate)
    assert peak < 288, peak
    keep = z_full[:8].clone()
    del z_full
    torch.cuda.empty_cache()
    z_shard, _ = search((0, 8))
    assert torch.equal(z_shard, keep)
    print('configs[3] full size: peak %.1f GB, loss %s' % (peak, ['%.3e' % v for v in losses]))


def _train_data(B=32, seed=21):
    g = torch.Generator().manual_seed(seed)
    return {'LR': torch.rand(B, 3, 52, 52, generator=g).to(DEV), 'HR': torch.rand(B, 3, 208, 208, generator=g).to(DEV),
            'Z': (torch.rand(B, 3, 208, 208, generator=g) * 2 - 1).to(DEV)}


def _cos(a, b):
    return float((a.double().flatten() @ b.double().flatten()) / (a.double().norm() * b.double().norm() + 1e-300))


def test_configs2_generator_step_bf16_against_split():
    """The per-GPU generator step of configs[2] (32 x 52x52, latent 3, RRDB-23, L1 + range losses) in bf16 (what the config names; one MFMA per
    product) and in split (fp32-class): same weights, same batch.  Stated tolerance: loss within 2 %, every parameter gradient's direction
    within cosine 0.98 of the fp32-class one, norms within 10 % (bf16 operands: ~1e-2 per layer)."""
    data = _train_data()
    res = {}
    for prec in ('split', 'bf16'):
        m = _model(True)
        m.netG.generated_image_model.set_precision(prec)
        for _ in range(2):                               # first call idles (no discriminator), second steps
            m.feed_data(data)
            m.optimize_parameters()
        params = [p for n, p in m.netG.named_parameters() if 'Filter_OP' not in n]
        res[prec] = (m.get_current_log()['l_g_pix'], [p.grad.clone() for p in params])
        del m
        torch.cuda.empty_cache()
    (ls, gs), (lb, gb) = res['split'], res['bf16']
    assert np.isfinite(lb) and abs(lb - ls) < 2e-2 * abs(ls), (ls, lb)
    gmax = max(float(a.norm()) for a in gs)
    cos = [_cos(a, b) for a, b in zip(gs, gb) if float(a.norm()) > 1e-4 * gmax]
    ratio = [float(b.norm() / a.norm()) for a, b in zip(gs, gb) if float(a.norm()) > 1e-4 * gmax]
    assert min(cos) > 0.98 and 0.9 < min(ratio) and max(ratio) < 1.1, (min(cos), min(ratio), max(ratio))
    print('configs[2] G step: l_g_pix split %.5f bf16 %.5f, gradient cosine min %.4f, norm ratio %.3f..%.3f' % (ls, lb, min(cos), min(ratio), max(ratio)))


def test_configs2_generator_plus_discriminator_step_bf16_against_fp32():
    """One whole configs[2] step per GPU — critic step (3 critic forwards, WGAN-GP double backward) + generator step (pixel, range, GAN terms)
    — with the generator in bf16 and the critic under bf16 autocast, against the same step with the split generator and the fp32 critic.
    Same weights, same batch, same interpolation points.  Stated tolerance: critic losses within 3 % (+0.02 absolute), the gradient penalty
    within 10 %, G's and D's parameter gradients within cosine 0.95 of the fp32-class ones for every tensor that carries signal."""
    data = _train_data()
    pts = torch.rand(32, 1, 1, 1, generator=torch.Generator().manual_seed(31)).to(DEV)
    res = {}
    for name, gprec, dprec in (('ref', 'split', None), ('bf16', 'bf16', torch.bfloat16)):
        m = _model(True, with_D=True)
        m.netG.generated_image_model.set_precision(gprec)
        m.D_dtype = dprec
        m._draw_interp_points = lambda n: pts
        for _ in range(2):                               # D_init_iters = 0: call 1 steps D only, call 2 steps D and G
            m.feed_data(data)
            m.optimize_parameters()
        # both critics ran on libesr_hip's kernels, in the precision this arm names (no fallback to the stock module on MIOpen)
        from esr_hip.critic import CriticEngine
        assert isinstance(m.D_engine, CriticEngine) and m.D_engine_fallback is None, m.D_engine_fallback
        assert m.D_engine.precision == ('bf16' if dprec is torch.bfloat16 else 'split'), m.D_engine.precision
        log = m.get_current_log()
        gp = [p.grad.clone() for n, p in m.netG.named_parameters() if 'Filter_OP' not in n]
        dp = [p.grad.clone() for p in m.netD.parameters()]
        res[name] = (log, gp, dp)
        del m
        torch.cuda.empty_cache()
    (lr_, gr, dr), (lb, gb, db) = res['ref'], res['bf16']
    for k in ('l_d_real', 'l_d_fake', 'l_g_gan', 'l_g_pix'):
        assert np.isfinite(lb[k]) and abs(lb[k] - lr_[k]) < 3e-2 * abs(lr_[k]) + 2e-2, (k, lr_[k], lb[k])
    assert abs(lb['l_d_gp'] - lr_['l_d_gp']) < 0.1 * abs(lr_['l_d_gp']) + 1e-3, (lr_['l_d_gp'], lb['l_d_gp'])
    for what, a_list, b_list in (('G', gr, gb), ('D', dr, db)):
        top = max(float(a.norm()) for a in a_list)
        cos = [_cos(a, b) for a, b in zip(a_list, b_list) if float(a.norm()) > 1e-4 * top]
        assert len(cos) >= 5 and min(cos) > 0.95, (what, len(cos), min(cos))
    print('configs[2] G+D step: ' + ', '.join('%s %.4f/%.4f' % (k, lr_[k], lb[k]) for k in ('l_d_real', 'l_d_fake', 'l_d_gp', 'l_g_gan', 'l_g_pix')))


def test_bench_c3_workload_prints_the_contract_line(capsys):
    """`python bench.py --workload c3` (configs[2] G+D step at its per-GPU shape): one JSON line with the contract's keys, phase times and losses."""
    import json
    import bench
    bench.main(['--workload', 'c3', '--steps', '2', '--warmup', '2'])
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['value'] > 0 and 'bf16' in d['dtype'] and 'workload' in d['config']
    assert set(d['phases_ms']) >= {'G_forward', 'D_step', 'G_losses_and_backward', 'G_allreduce_and_Adam'}
    assert all(np.isfinite(v) for v in d['losses'].values()) and {'l_d_real', 'l_d_gp', 'l_g_gan', 'l_g_pix'} <= set(d['losses'])
