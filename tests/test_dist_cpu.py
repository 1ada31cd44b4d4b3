"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel path: gradient bucket all-reduce, batch sharding, and the
sharded Z search reproducing the single-process result.  The same code runs over RCCL on GPUs (backend "nccl")."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, 'explorable-super-resolution_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from esr_hip import dist as D
    D.init_from_env(backend='gloo')
    return D


def _worker_allreduce(rank, world, port, q):
    D = _setup(rank, world, port)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.Conv2d(8, 5, 3), torch.nn.Linear(7, 3))
    if rank == 1:                                   # de-synchronise, then broadcast from rank 0
        for p in net.parameters():
            p.data.add_(1.0)
    D.broadcast_parameters(net)
    params = list(net.parameters())
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    params[1].grad = None if rank == 1 else params[1].grad      # a rank without a gradient contributes zeros
    red = D.GradBucketAllReducer(params, bucket_mb=0.0005)      # tiny buckets: several collectives
    red()
    q.put((rank, [float(p.grad.flatten()[0]) for p in params], [float(p.data.flatten()[0]) for p in params], len(red.buckets),
           D.shard_range(7), D.all_reduce_mean_scalar(float(rank)), D.gather_scalars(torch.tensor([rank + 1.0, 10.0 * rank])).tolist()))
    dist.destroy_process_group()


def test_grad_bucket_allreduce_and_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_allreduce, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    (r0, g0, w0, nb0, s0, m0, gs0), (r1, g1, w1, nb1, s1, m1, gs1) = res
    assert gs0 == gs1 == [[1.0, 0.0], [2.0, 10.0]]                   # every rank sees every rank's bookkeeping scalars, in rank order
    assert g0 == g1 and w0 == w1                                     # identical gradients and weights on every rank
    expect = [1.5 * (i + 1) for i in range(len(g0))]
    expect[1] = 0.5 * 2                                              # only rank 0 had a gradient for parameter 1: (2*1 + 0)/2
    assert np.allclose(g0, expect)
    assert nb0 > 1
    assert s0 == (0, 4) and s1 == (4, 7)                             # contiguous shards, sizes differ by at most one
    assert m0 == m1 == 0.5


def _run_zopt(D, batch):
    from test_host_api import _ToyModel
    from Z_optimization import Z_optimizer
    torch.manual_seed(0)
    model = _ToyModel()
    lr = torch.rand(1, 3, 4, 4)
    lo, hi = D.shard_range(batch)
    model.feed_data({'LR': lr.expand(hi - lo, -1, -1, -1), 'Z': torch.zeros(hi - lo, 1, 16, 16)})
    model.test()
    z0 = (torch.arange(batch).float().view(-1, 1, 1, 1) * 0.1 - 0.1) * torch.ones(batch, 1, 16, 16)
    zo = Z_optimizer(objective='STD_increase', Z_size=[16, 16], model=model, Z_range=1, max_iters=6, data={'LR': lr, 'STD_increment': 0.02},
                     initial_LR=0.05, batch_size=batch, initial_Z=z0)
    Z = zo.optimize()
    return Z, zo.loss_values, (lo, hi)


def _worker_zopt(rank, world, port, q):
    D = _setup(rank, world, port)
    Z, losses, shard = _run_zopt(D, 4)
    q.put((rank, Z.numpy(), losses, shard))
    dist.destroy_process_group()


def test_sharded_z_search_matches_single_process():
    for p in (ROOT, os.path.join(ROOT, 'explorable-super-resolution_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    from esr_hip import dist as D
    Z_ref, loss_ref, _ = _run_zopt(D, 4)                              # single process, whole batch
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_zopt, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    [p.join(timeout=60) for p in procs]
    Z = np.concatenate([r[1] for r in res], 0)
    assert [r[3] for r in res] == [(0, 2), (2, 4)]
    np.testing.assert_allclose(Z, Z_ref.numpy(), atol=1e-6)           # each Z sample has its own Adam state: sharding is exact
    np.testing.assert_allclose(res[0][2], loss_ref, rtol=1e-5)        # the (all-reduced) loss history is the global batch mean
    np.testing.assert_allclose(res[1][2], loss_ref, rtol=1e-5)


def _worker_flat(rank, world, port, q):
    D = _setup(rank, world, port)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.Conv2d(8, 5, 3))
    params = list(net.parameters())
    # gradients handed out as consecutive views of ONE flat buffer, as the engine's batched weight-gradient launch does (engine.WGrad)
    flat = torch.arange(sum(p.numel() for p in params), dtype=torch.float32) * (rank + 1)
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    red = D.GradBucketAllReducer(params, bucket_mb=0.001)        # 2 buckets: each one a contiguous slice of the flat buffer
    ptrs = [p.grad.data_ptr() for p in params]
    red()
    first = (red.in_place, len(red.buckets), flat.clone().numpy(), ptrs == [p.grad.data_ptr() for p in params])
    # (ii) a reducer over a SUBSET of the parameters whose gradients share the flat buffer (the middle two of four), after a second
    # backward pass accumulated into the same views (gradient accumulation: AccumulateGrad adds in place): only the subset's range of the
    # buffer may change
    flat2 = torch.ones(sum(p.numel() for p in params), dtype=torch.float32) * (rank + 1)
    off = 0
    for p in params:
        p.grad = flat2[off:off + p.numel()].view_as(p)
        off += p.numel()
    for p in params:
        p.grad.add_(torch.full_like(p, 2.0 * (rank + 1)))          # accumulation step 2, in place
    sub = D.GradBucketAllReducer(params[1:3], bucket_mb=32.0)
    sub()
    lo, hi = params[0].numel(), params[0].numel() + params[1].numel() + params[2].numel()
    q.put((rank,) + first + (sub.in_place, flat2.clone().numpy(), lo, hi))
    dist.destroy_process_group()


def test_grad_buckets_reduce_in_place_out_of_the_flat_gradient_buffer():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_flat, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    for rank, in_place, nb, flat, same_storage, sub_in_place, flat2, lo, hi in res:
        assert nb >= 2 and in_place == nb and same_storage          # every bucket reduced in place: no gather / scatter copies
        assert np.allclose(flat, np.arange(flat.size) * 1.5)        # mean of (1x, 2x)
        own = 3.0 * (rank + 1)                                      # this rank's accumulated gradient (1 + 2) * (rank + 1)
        assert sub_in_place == 1
        assert np.allclose(flat2[lo:hi], 4.5)                       # the subset: mean of 3 and 6, in place
        assert np.allclose(flat2[:lo], own) and np.allclose(flat2[hi:], own)     # its neighbours in the same buffer: untouched


def _worker_single_rank(port, q):
    for p in (ROOT, os.path.join(ROOT, 'explorable-super-resolution_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    from esr_hip import dist as D
    D.init_from_env(backend='gloo')                # launcher variables of a one-rank job: no group, no collectives
    a = (dist.is_initialized(), D.is_distributed())
    D.init_from_env(backend='gloo', single_rank=True)        # the explicit opt-in of bench.py / the RCCL tests
    b = (dist.is_initialized(), D.is_distributed(), D.all_reduce_mean_scalar(3.5))
    dist.destroy_process_group()
    q.put((a, b))


def test_a_single_rank_job_runs_no_collectives_unless_asked_to():
    """WORLD_SIZE = 1 in the environment (torchrun --nproc-per-node 1, SLURM) must not cost a process group and per-step collectives;
    init_from_env(single_rank=True) is the opt-in that keeps the collective code path testable on one GPU."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker_single_rank, args=(_free_port(), q))
    p.start()
    a, b = q.get(timeout=120)
    p.join(30)
    assert a == (False, False)
    assert b == (True, True, 3.5)


def _worker_early(rank, world, port, q):
    D = _setup(rank, world, port)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 7, 3, 9)]
    flat = torch.arange(24, dtype=torch.float32) * (rank + 1)                 # the "flat weight-gradient buffer" of a backward pass
    red = D.EarlyBucketReducer(params, bucket_mb=1e-5)
    # what the engine does: one start() per bucket behind its launch, finish() behind the last
    red.start(0, flat[:12])
    red.start(1, flat[12:])
    red.finish()
    early = (red.early_buckets, flat.clone().numpy())
    red()                                                                      # the step's exchange already happened: a no-op
    after = flat.clone().numpy()
    # a step whose backward did not go through the engine: the late, bucketed exchange over .grad
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    red()
    q.put((rank, early, after, [float(p.grad[0]) for p in params], red.early_buckets))
    dist.destroy_process_group()


def test_early_bucket_reducer_averages_in_place_and_falls_back_to_the_late_exchange():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_early, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(30) for p in procs]
    want = np.arange(24, dtype=np.float32) * 1.5                               # mean of x1 and x2
    for rank, (nb, flat), after, late, early_after in res:
        assert nb == 2 and np.array_equal(flat, want) and np.array_equal(after, want)
        assert late == [1.5, 3.0, 4.5, 6.0] and early_after == 0
