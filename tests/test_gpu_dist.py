"""GPU test (-m gpu) of the data-parallel training path on REAL device tensors: two ranks (processes) share the one GPU of the test box and
exchange through the gloo backend (RCCL refuses two ranks on one device; gloo all-reduces / broadcasts device tensors through the host).  What
runs on every rank is exactly what runs under torchrun on N GPUs: rank-0 weights broadcast at construction (weight packs invalidated), a batch
shard per rank through the HIP generator forward / backward, the bucketed all-reduce IN PLACE on the flat buffer the batched weight-gradient
launch wrote, Adam.  Checked: both ranks hold bit-identical gradients and weights afterwards, and the averaged gradient equals the gradient of
the same step run by one process on the whole batch (the L1 loss is a mean, the shards have equal sizes)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle.check_golden import rel_l2
from oracle.weights import seeded_uniform

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB, PER_RANK = 2, 2


def _paths():
    for p in (ROOT, os.path.join(ROOT, 'explorable-super-resolution_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)


def _batch(world):
    n = PER_RANK * world
    return seeded_uniform((n, 3, 24, 28), 301), seeded_uniform((n, 3, 96, 112), 302)


def _step(seed, lr, hr, precision=None, early=False):
    """One SRRaGANModel generator step (two calls: without a discriminator the first one is idle, as in the reference) on the given shard."""
    _paths()
    import models
    from test_host_api import _opt
    opt = _opt(nb=NB, lat=0, cem=True, is_train=True)
    opt['gpu_ids'] = [0]
    torch.manual_seed(seed)                      # rank-dependent initial weights: the constructor's broadcast has to make them rank 0's
    m = models.create_model(opt)
    if precision is not None:
        m.netG.generated_image_model.set_precision(precision)
    if early:
        m.grad_reducer.ENABLED = True
    for _ in range(2):
        m.feed_data({'LR': lr, 'HR': hr})
        m.optimize_parameters()
    names = [k for k, v in m.netG.named_parameters() if v.requires_grad]
    got = dict(m.netG.named_parameters())
    grads = {k: got[k].grad.detach().cpu().numpy().copy() for k in names}
    weights = {k: got[k].detach().cpu().numpy().copy() for k in names}
    return m, names, grads, weights


def _worker(rank, world, port, q, precision=None, early=False):
    try:
        _paths()
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
        torch.cuda.set_device(0)
        from esr_hip import dist as D
        D.init_from_env(backend='gloo')
        lr, hr = _batch(world)
        lo, hi = D.shard_range(lr.size(0))
        m, names, grads, weights = _step(rank, lr[lo:hi], hr[lo:hi], precision, early)
        pick = [names[0], names[len(names) // 2], names[-2]]
        q.put((rank, (lo, hi), {k: grads[k] for k in pick}, {k: weights[k] for k in pick},
               float(sum(float(np.abs(g).sum()) for g in grads.values())), int(m.grad_reducer.in_place), len(m.grad_reducer.buckets),
               float(m.get_current_log()['l_g_pix']), int(m.grad_reducer.early_buckets)))
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                       # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put((rank, 'error', traceback.format_exc() + repr(e)))


def test_two_ranks_on_one_gpu_average_the_generator_gradients_in_place():
    world = 2
    lr, hr = _batch(world)
    _, names, g_full, _ = _step(0, lr, hr)       # one process, the whole batch, rank 0's initial weights
    torch.cuda.synchronize()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    [p.join(timeout=120) for p in procs]
    for r in res:
        assert not (isinstance(r[1], str) and r[1] == 'error'), r[2]
    (_, sh0, g0, w0, n0, inplace0, nb0, l0, _), (_, sh1, g1, w1, n1, inplace1, nb1, l1, _) = res
    assert sh0 == (0, PER_RANK) and sh1 == (PER_RANK, 2 * PER_RANK)
    assert inplace0 == nb0 >= 1 and inplace1 == nb1          # every bucket was reduced in place on the flat weight-gradient buffer
    assert n0 == n1
    for k in g0:
        assert np.array_equal(g0[k], g1[k]) and np.array_equal(w0[k], w1[k]), k          # bit-identical after the all-reduce / after Adam
        ref = g_full[k]
        gmax = max(float(np.abs(v).max()) for v in g_full.values())
        if np.abs(ref).max() < 1e-5 * gmax:          # analytically zero gradients (last bias under the CEM) hold rounding noise on both sides
            assert np.abs(g0[k]).max() < 1e-4 * gmax
        else:
            # mean of the two shards' gradients == gradient of the whole-batch mean loss; the contraction order differs, the bar is the
            # fp32-class 1e-3 of the weight-gradient tests
            assert rel_l2(g0[k], ref) < 1e-3, (k, rel_l2(g0[k], ref))
    assert l0 != l1                                            # the ranks really saw different shards


def _two_ranks(precision, early):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, precision, early)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    [p.join(timeout=120) for p in procs]
    for r in res:
        assert not (isinstance(r[1], str) and r[1] == 'error'), r[2]
    return res


@pytest.mark.parametrize('precision', ['mixed', 'bf16'])
def test_early_exchange_on_two_ranks_with_different_shards_equals_the_exchange_after_the_backward(precision):
    """'mixed' back-propagates under power-of-two gradient scales that every RANK derives from its own gradients; a flat dW
    buffer that is all-reduced before those scales are undone mixes differently scaled sums and then divides by the local scale — different,
    wrong gradients on every rank, invisible at world size 1.  The engine therefore leaves scaled (and permuted) gradient sets to the exchange
    after the backward.  Two ranks, different shards: with train.early_gradient_exchange switched on both ranks must end with the gradients
    and weights of the late exchange, bit for bit — for 'mixed' through the late path (no early bucket), for 'bf16' through the two-stream
    backward's groups (three early buckets; a two-rank sum does not depend on how the buffer is cut)."""
    late, early = _two_ranks(precision, False), _two_ranks(precision, True)
    for r in range(2):
        for k in late[r][2]:
            assert np.array_equal(late[r][2][k], early[r][2][k]) and np.array_equal(late[r][3][k], early[r][3][k]), (precision, r, k)
    for k in early[0][2]:
        assert np.array_equal(early[0][2][k], early[1][2][k]) and np.array_equal(early[0][3][k], early[1][3][k]), k       # the replicas stay replicas
    assert early[0][7] != early[1][7]                          # different shards (different pixel losses)
    assert [r[8] for r in late] == [0, 0]
    assert [r[8] for r in early] == ([0, 0] if precision == 'mixed' else [3, 3]), [r[8] for r in early]


def _run_bench(extra, timeout):
    import json
    import subprocess
    env = dict(os.environ, ESR_BENCH_SHARE_GPU='1', OMP_NUM_THREADS='4')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'] + extra, env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]                    # rank 0 prints ONE json line
    return json.loads(lines[0])


def test_bench_training_step_workload_on_two_ranks():
    """configs[2] through bench.py on two ranks: generator + critic step per rank, both gradient sets all-reduced (in place for the generator)."""
    d = _run_bench(['--workload', 'c3'], 1500)
    assert d['n_gpus'] == 2 and d['world_size_seen'] == 2
    assert set(d['phases_ms']) >= {'G_forward', 'D_step', 'G_losses_and_backward', 'G_allreduce_and_Adam'}
    assert all(np.isfinite(v) for v in d['losses'].values()), d['losses']
    assert abs(d['value'] - 2 * 32 / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']


def test_bench_launches_its_own_two_ranks_and_its_default_line_carries_the_communication_diagnosis():
    """the first N > 1 record has to answer the communication questions by itself.  `bench.py --gpus 2` (default workload)
    appends `extra_workloads.c3` — the only workload with data-path collectives — with a `comm` block: the step with the gradient exchange
    after the backward, from inside it, and without any exchange, and the exposed communication time derived from them; the ranks' backend is
    in `ranks`.  The headline keys are those of the N = 1 line."""
    d = _run_bench(['--no-cpu-baseline', '--no-alt-precision'], 1800)
    # `python bench.py --gpus 2` started as a plain process (the way the driver starts N = 1): it re-executes itself under torch.distributed.run,
    # every rank times its own shard between barriers, rank 0 reports the max over ranks and the aggregate.  ESR_BENCH_SHARE_GPU lets the two
    # ranks share this box's one GPU (gloo instead of RCCL); the logic is the one the 8-GPU run uses, the numbers are not a measurement.
    assert d['n_gpus'] == 2 and d['world_size_seen'] == 2 and d['config']['workload'].startswith('configs[1]')
    assert d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert len(d['ms_per_step_per_rank']) == 2 and abs(max(d['ms_per_step_per_rank']) - d['ms_per_step']) < 1e-6
    assert abs(d['value'] - 2 * 32 * 512 * 512 / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']        # aggregate over both ranks
    assert d['roofline']['bound'] == 'hbm' and 0 < d['roofline']['frac'] < 1
    assert all(r['backend'] == 'gloo' for r in d['ranks']) and len(d['ranks']) == 2
    c3 = d['extra_workloads']['c3']
    assert 'error' not in c3, c3
    assert set(d['extra_workloads']) == {'c3'}                  # the single-GPU blocks (c5, c4) are not repeated per N
    assert c3['n_gpus'] == 2 and 'G_allreduce_and_Adam' in c3['phases_ms']
    comm = c3['comm']
    for k in ('ms_per_step_exchange_after_backward', 'ms_per_step_exchange_inside_backward', 'ms_per_step_no_exchange', 'exposed_comm_ms'):
        assert np.isfinite(comm[k]), (k, comm)
    assert comm['ms_per_step_exchange_after_backward'] > 0 and comm['ms_per_step_exchange_inside_backward'] > 0 and comm['ms_per_step_no_exchange'] > 0
    assert comm['early_buckets'] >= 1 and comm['G_gradient_bytes'] > 60e6 and comm['backend'] == 'gloo'
    assert abs(comm['exposed_comm_ms'] - (comm['ms_per_step_exchange_after_backward'] - comm['ms_per_step_no_exchange'])) < 1e-9


def _zsearch(D, batch):
    """STD_increase Z search of `batch` latent samples of one LR image on the real generator (RRDB-1, latent 3); this rank's shard."""
    _paths()
    import models
    from oracle.weights import fill_formula_weights
    from test_host_api import _opt
    from Z_optimization import Z_optimizer
    opt = _opt(nb=1, lat=3, cem=True, is_train=False)
    opt['gpu_ids'] = [0]
    m = models.create_model(opt)
    fill_formula_weights(m.netG, gain=1.0)
    lr = seeded_uniform((1, 3, 20, 24), 311)
    lo, hi = D.shard_range(batch)
    z0 = seeded_uniform((batch, 3, 80, 96), 312, -0.3, 0.3)[lo:hi]
    m.feed_data({'LR': lr.expand(hi - lo, -1, -1, -1), 'Z': z0}, need_GT=False)
    m.test()
    zo = Z_optimizer(objective='STD_increase', Z_size=[80, 96], model=m, Z_range=1, max_iters=4, data={'LR': lr, 'STD_increment': 0.02},
                     initial_LR=0.05, batch_size=batch, initial_Z=z0.to(m.device))
    Z = zo.optimize()
    return Z.detach().cpu().numpy(), [float(v) for v in zo.loss_values], (lo, hi)


def _worker_z(rank, world, port, q):
    try:
        _paths()
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
        torch.cuda.set_device(0)
        from esr_hip import dist as D
        D.init_from_env(backend='gloo')
        q.put((rank,) + _zsearch(D, 4))
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, 'error', traceback.format_exc() + repr(e)))


def test_sharded_z_search_on_the_real_generator_matches_one_process():
    """configs[3] across ranks: every rank optimises its own Z samples (own Adam state, no data-path collective), the loss history is the
    all-reduced batch mean, per-sample gradients keep the 1/B_global scale of the reference's batch-mean backward."""
    _paths()
    from esr_hip import dist as D
    Z_ref, loss_ref, _ = _zsearch(D, 4)
    torch.cuda.synchronize()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_z, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    [p.join(timeout=120) for p in procs]
    for r in res:
        assert not (isinstance(r[1], str) and r[1] == 'error'), r[2]
    assert [r[3] for r in res] == [(0, 2), (2, 4)]
    Z = np.concatenate([r[1] for r in res], 0)
    # the same kernels on the same samples (a shard changes the batch size of the launches, not a sample's arithmetic)
    d = np.abs(Z - Z_ref)
    assert np.median(d) < 1e-4 and np.mean(d > 1e-2) < 0.02, (float(np.median(d)), float(np.mean(d > 1e-2)))
    np.testing.assert_allclose(res[0][2], loss_ref, rtol=1e-3)
    np.testing.assert_allclose(res[1][2], loss_ref, rtol=1e-3)


def _worker_gan(rank, world, port, q):
    try:
        _paths()
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
        torch.cuda.set_device(0)
        from esr_hip import dist as D
        D.init_from_env(backend='gloo')
        import models
        from tools.bench_paths import make_opt
        opt = make_opt(True, lat=3, nb=1, with_D=True)
        opt['train']['D_verification'] = 'current'            # G steps gated on the CURRENT batch's critic statistics (reference :392-400)
        opt['train']['min_D_prob_ratio_4_G'] = 1.0
        torch.manual_seed(rank)
        m = models.create_model(opt)
        g = torch.Generator().manual_seed(50 + rank)          # a different shard per rank: local statistics would disagree
        decisions = []
        for it in range(4):
            m.step = it
            m.feed_data({'LR': torch.rand(2, 3, 52, 52, generator=g), 'HR': torch.rand(2, 3, 208, 208, generator=g)})
            m.optimize_parameters()
            decisions.append(bool(m.generator_step))
        log = m.get_current_log()
        w = [p for n, p in m.netG.named_parameters() if 'Filter_OP' not in n][0]     # (requires_grad is toggled per step by the gating)
        q.put((rank, decisions, float(log['D_logits_diff']), float(log['Correctly_distinguished']), float(w.detach().abs().sum())))
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, 'error', traceback.format_exc() + repr(e)))


def test_ranks_take_the_same_generator_step_decisions_from_global_critic_statistics():
    """configs[2] with the reference's D-verification gating on two ranks: the statistics that decide whether a generator step happens
    (per-sample critic logit differences of the current batch; their logged means for the 'past' mode) are made global before they are used —
    ranks deciding differently would enter different collectives.  Four G + D steps on different shards: same decisions, same logged
    statistics, same generator weights on both ranks, no hang."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_gan, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=900) for _ in range(2)), key=lambda t: t[0])
    [p.join(timeout=120) for p in procs]
    for r in res:
        assert not (isinstance(r[1], str) and r[1] == 'error'), r[2]
    assert res[0][1] == res[1][1] and len(res[0][1]) == 4
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]
    assert res[0][4] == res[1][4]
