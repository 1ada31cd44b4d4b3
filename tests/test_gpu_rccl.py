"""GPU tests (-m gpu) that EXECUTE the RCCL code path on the hardware there is: a job started by a launcher runs its
collectives whatever the world size is (esr_hip.dist.is_distributed), so one rank on the test box's one GPU really calls
init_process_group(backend='nccl', device_id=...), the in-place bucket all-reduce on the flat weight-gradient buffer, broadcast_parameters and
gather_scalars through RCCL — and, every collective being the identity at one rank, must reproduce the plain single process bit for bit.
The two-GPU twins at the bottom become live on any box with device_count() >= 2 (same workers, LOCAL_RANK = rank, backend nccl)."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle.weights import seeded_uniform

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _paths():
    for p in (ROOT, os.path.join(ROOT, 'explorable-super-resolution_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _digest(tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().float().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()


def _generator_steps(shard=None, bucket_kb=None, precision=None, early=False):
    """Two SRRaGANModel generator steps (RRDB-2 + CEM, L1): digests of the gradients after the all-reduce and of the weights after Adam."""
    _paths()
    import models
    from esr_hip import dist as D
    from test_host_api import _opt
    opt = _opt(nb=2, lat=0, cem=True, is_train=True)
    opt['gpu_ids'] = [0]
    torch.manual_seed(D.rank())                  # rank-dependent initial weights: the constructor's broadcast makes them rank 0's
    m = models.create_model(opt)
    if precision is not None:
        m.netG.generated_image_model.set_precision(precision)
    if early:
        m.grad_reducer.ENABLED = True
    if bucket_kb is not None:                    # small buckets: the backward's weight-gradient launch is cut into several, each followed by its all-reduce
        m.grad_reducer.bucket_bytes = bucket_kb * 1024
        m.grad_reducer.ENABLED = True             # (off by default: train.early_gradient_exchange)
    lr, hr = seeded_uniform((4, 3, 24, 28), 301), seeded_uniform((4, 3, 96, 112), 302)
    lo, hi = shard if shard is not None else D.shard_range(4)
    for _ in range(3):
        m.feed_data({'LR': lr[lo:hi], 'HR': hr[lo:hi]})
        m.optimize_parameters()
    ps = [p for p in m.netG.parameters() if p.requires_grad]
    return {'grads': _digest([p.grad for p in ps]), 'weights': _digest(ps), 'in_place': int(m.grad_reducer.in_place), 'buckets': len(m.grad_reducer.buckets), 'early': int(m.grad_reducer.early_buckets),
            'l_g_pix': float(m.get_current_log()['l_g_pix']), 'gsum': float(sum(p.grad.double().abs().sum() for p in ps))}


def _gan_steps():
    """Three G + D steps of the configs[2] kind (critic on libesr_hip, WGAN-GP, D_verification = 'current': gather_scalars gates the G steps)."""
    _paths()
    import models
    from esr_hip import dist as D
    from esr_hip.critic import CriticEngine
    from tools.bench_paths import make_opt
    opt = make_opt(True, lat=3, nb=1, with_D=True)
    opt['train']['D_verification'] = 'current'
    opt['train']['min_D_prob_ratio_4_G'] = 1.0
    torch.manual_seed(D.rank())
    m = models.create_model(opt)
    assert isinstance(m.D_engine, CriticEngine) and m.D_engine_fallback is None
    g = torch.Generator().manual_seed(50)
    pts = torch.rand(3, 2, 1, 1, 1, generator=torch.Generator().manual_seed(7)).to(m.device)
    decisions = []
    for it in range(3):
        m.step = it
        m._draw_interp_points = lambda n, it=it: pts[it]
        m.feed_data({'LR': torch.rand(2, 3, 52, 52, generator=g), 'HR': torch.rand(2, 3, 208, 208, generator=g)})
        m.optimize_parameters()
        decisions.append(bool(m.generator_step))
    log = m.get_current_log()
    gw = [p for n, p in m.netG.named_parameters() if 'Filter_OP' not in n]
    return {'decisions': decisions, 'D_logits_diff': float(log['D_logits_diff']), 'l_d_gp': float(log['l_d_gp']),
            'G': [float(p.detach().double().abs().sum()) for p in gw[:6]], 'D': [float(p.detach().double().abs().sum()) for p in list(m.netD.parameters())[:6]],
            'D_in_place': int(m.grad_reducer_D.in_place), 'D_buckets': len(m.grad_reducer_D.buckets)}


def _worker(rank, world, port, what, q):
    try:
        _paths()
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                          HSA_ENABLE_IPC_MODE_LEGACY='0')
        from esr_hip import dist as D
        D.init_from_env(single_rank=True)         # default backend with a GPU: 'nccl' (= RCCL), device_id = cuda:LOCAL_RANK; one rank: collectives ON
        import torch.distributed as dist
        info = {'backend': dist.get_backend(), 'world': dist.get_world_size(), 'device': torch.cuda.current_device(),
                'rccl': '.'.join(str(v) for v in torch.cuda.nccl.version())}
        # the collectives the trainer uses, directly, on device tensors
        t = torch.arange(8, dtype=torch.float32, device='cuda') + rank
        rows = D.gather_scalars(t[:3])
        info['gather_rows'] = rows.cpu().tolist()
        info['mean_scalar'] = D.all_reduce_mean_scalar(2.5 + rank, torch.device('cuda'))
        res = {'g': _generator_steps, 'g_small': lambda: _generator_steps(bucket_kb=256), 'g_overlap': lambda: _generator_steps(precision='bf16', early=True),
               'g_bf16': lambda: _generator_steps(precision='bf16'), 'gan': _gan_steps}[what]()
        q.put((rank, info, res))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                       # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put((rank, 'error', traceback.format_exc() + repr(e)))


def _spawn(world, what, timeout=900):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, what, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=timeout) for _ in range(world)), key=lambda t: t[0])
    [p.join(timeout=120) for p in procs]
    for r in res:
        assert not (isinstance(r[1], str) and r[1] == 'error'), r[2]
    return res


def test_one_rank_over_rccl_reproduces_the_plain_process_generator_step_bit_for_bit():
    ref = _generator_steps(shard=(0, 4))         # this process: no process group, no collectives
    assert ref['in_place'] == 0
    torch.cuda.synchronize()
    (_, info, got), = _spawn(1, 'g')
    assert info['backend'] == 'nccl' and info['world'] == 1 and info['rccl'][0].isdigit(), info
    assert info['gather_rows'] == [[0.0, 1.0, 2.0]] and info['mean_scalar'] == 2.5
    assert got['in_place'] == got['buckets'] >= 1            # every bucket all-reduced IN PLACE on the flat weight-gradient buffer, through RCCL
    assert got['grads'] == ref['grads'] and got['weights'] == ref['weights'], (got, ref)
    assert got['l_g_pix'] == ref['l_g_pix']


def test_bucketed_weight_gradient_launches_with_early_all_reduce_are_bit_identical():
    """under a process group the generator's backward launches its weight gradients bucket by bucket and starts every bucket's
    in-place all-reduce behind its launch (RRDBEngine.wgrad_exchange / esr_hip.dist.EarlyBucketReducer) — here with 256 KB buckets, so that the
    RRDB-2 generator's 1.6 MB of gradients are several launches and several RCCL collectives: gradients and weights after three steps are those of
    the plain process (one launch, no collective) bit for bit."""
    ref = _generator_steps(shard=(0, 4))
    assert ref['early'] == 0
    torch.cuda.synchronize()
    (_, info, got), = _spawn(1, 'g_small')
    assert info['backend'] == 'nccl'
    assert got['early'] >= 3, got                # exchanged from inside the backward, in several buckets
    assert got['grads'] == ref['grads'] and got['weights'] == ref['weights'], (got, ref)


def test_two_stream_backward_exchanges_its_groups_as_they_finish_bit_identically():
    """The bf16 generator's recorded backward launches its weight gradients in RRDBEngine.wgrad_overlap groups, two of them on the second stream under
    the data-gradient chain; with the early exchange switched on every group's stretch of the flat gradient buffer is all-reduced (in place, RCCL)
    behind its own launch on the stream that computed it, and the main stream joins at the end: gradients and weights after three steps are those of
    the plain process (no process group, exchange off) bit for bit."""
    ref = _generator_steps(shard=(0, 4), precision='bf16')
    assert ref['early'] == 0
    torch.cuda.synchronize()
    (_, info, got), = _spawn(1, 'g_overlap')
    assert info['backend'] == 'nccl'
    assert got['early'] == 3, got                # = the engine's groups
    assert got['grads'] == ref['grads'] and got['weights'] == ref['weights'], (got, ref)


def test_one_rank_over_rccl_runs_the_generator_plus_critic_step_like_the_plain_process():
    ref = _gan_steps()
    torch.cuda.synchronize()
    (_, info, got), = _spawn(1, 'gan')
    assert info['backend'] == 'nccl' and info['world'] == 1
    assert got['decisions'] == ref['decisions'] and len(got['decisions']) == 3
    assert got['D_in_place'] + 0 >= 0 and got['D_buckets'] >= 1
    # same kernels, same order; the critic's torch-side pieces (two Linear layers on rocBLAS) are not promised to be run-to-run bit-stable
    np.testing.assert_allclose(got['G'], ref['G'], rtol=1e-6)
    np.testing.assert_allclose(got['D'], ref['D'], rtol=1e-6)
    np.testing.assert_allclose([got['D_logits_diff'], got['l_d_gp']], [ref['D_logits_diff'], ref['l_d_gp']], rtol=1e-4)


def _bench_under_launcher(nproc, extra, timeout):
    env = dict(os.environ, OMP_NUM_THREADS='4', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'ESR_BENCH_SHARE_GPU'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', str(nproc), '--steps', '2', '--warmup', '1'] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize('workload', ['c2', 'c3'])
def test_bench_under_the_launcher_with_one_rank_talks_rccl(workload):
    """What the driver does for N > 1, with N = 1: `python -m torch.distributed.run ... bench.py --gpus 1`.  The rank initialises the nccl
    backend on its device, the barriers / per-rank time exchange (and, for c3, the gradient all-reduces of both networks) run through RCCL,
    and the JSON line says so."""
    extra = ['--no-cpu-baseline', '--no-alt-precision', '--batch', '4'] if workload == 'c2' else ['--workload', 'c3']
    d = _bench_under_launcher(1, extra, 1500)
    assert d['n_gpus'] == 1 and d['world_size_seen'] == 1 and d['distinct_gpus'] == 1
    r0 = d['ranks'][0]
    assert r0['backend'] == 'nccl' and r0['rccl_version'][0].isdigit(), r0
    assert d['value'] > 0 and len(d['ms_per_step_per_rank']) == 1


needs2 = pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: live on the multi-GPU node')


@needs2
def test_two_gpus_over_rccl_average_the_generator_gradients_in_place():
    res = _spawn(2, 'g')
    (_, i0, g0), (_, i1, g1) = res
    assert i0['backend'] == i1['backend'] == 'nccl' and i0['world'] == 2 and {i0['device'], i1['device']} == {0, 1}
    assert i0['gather_rows'] == [[0.0, 1.0, 2.0], [1.0, 2.0, 3.0]] and i0['mean_scalar'] == 3.0
    assert g0['in_place'] == g0['buckets'] and g1['in_place'] == g1['buckets']
    assert g0['grads'] == g1['grads'] and g0['weights'] == g1['weights']          # bit-identical on both ranks after the all-reduce / after Adam
    assert g0['l_g_pix'] != g1['l_g_pix']                                           # they saw different shards
    ref = _generator_steps(shard=(0, 4))                                            # one process on the whole batch: the mean of the shards' gradients
    assert abs(g0['gsum'] - ref['gsum']) < 2e-3 * ref['gsum'], (g0['gsum'], ref['gsum'])


@needs2
def test_two_gpus_two_stream_backward_with_early_exchange_equals_the_exchange_after_the_backward():
    """The early gradient exchange of the two-stream backward (three groups all-reduced behind their own launches on RCCL's stream, the main
    stream joining at the end) on two REAL ranks with different shards: gradients and weights after three steps are those of the same job with
    the exchange after the backward, bit for bit, and the two replicas agree."""
    late, early = _spawn(2, 'g_bf16'), _spawn(2, 'g_overlap')
    assert all(r[1]['backend'] == 'nccl' for r in late + early)
    assert [r[2]['early'] for r in early] == [3, 3] and [r[2]['early'] for r in late] == [0, 0]
    assert early[0][2]['grads'] == early[1][2]['grads'] == late[0][2]['grads'] == late[1][2]['grads']
    assert early[0][2]['weights'] == early[1][2]['weights'] == late[0][2]['weights']
    assert early[0][2]['l_g_pix'] != early[1][2]['l_g_pix']


@needs2
def test_bench_default_line_on_two_gpus_carries_the_communication_diagnosis():
    d = _bench_under_launcher(2, ['--no-cpu-baseline', '--no-alt-precision'], 1800)
    comm = d['extra_workloads']['c3']['comm']
    assert comm['backend'] == 'nccl' and comm['early_buckets'] == 3
    assert all(np.isfinite(comm[k]) and comm[k] > 0 for k in ('ms_per_step_exchange_after_backward', 'ms_per_step_exchange_inside_backward', 'ms_per_step_no_exchange'))


@needs2
def test_bench_on_two_gpus_over_rccl():
    d = _bench_under_launcher(2, ['--no-cpu-baseline', '--no-alt-precision', '--batch', '4'], 1500)
    assert d['n_gpus'] == 2 and d['world_size_seen'] == 2 and d['distinct_gpus'] == 2
    assert all(r['backend'] == 'nccl' for r in d['ranks'])
