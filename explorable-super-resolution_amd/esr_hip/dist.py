"""Data parallelism the MI355X way: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), instead of the
reference's single-process nn.DataParallel (codes/models/networks.py:120-123) which re-broadcasts ~68 MB of parameters every
forward and gathers outputs / reduces gradients on GPU 0.

  * inference and latent (Z) search shard independent images / Z samples over the ranks — no data-path collective
  * training all-reduces the generator gradients once per optimiser step, in few large buckets (xGMI is point-to-point:
    ring collectives are per-link bound, so fewer, larger messages; RRDB-23 has 16.7 M parameters = 67 MB fp32)
Works with backend "gloo" on CPU tensors, which is how the logic is tested without GPUs.
"""
import os

import torch
import torch.distributed as dist


# A one-rank group runs no collectives unless this is set (init_from_env(single_rank=True) sets it; bench.py and the RCCL tests do, so that the
# one-GPU box the tests have executes the RCCL code path itself — with one rank every collective is the identity, results are bit-identical to
# the plain single process).  A single-GPU job that merely inherits launcher variables (torchrun --nproc-per-node 1, SLURM) pays nothing.
SINGLE_RANK_COLLECTIVES = False


def is_distributed():
    """True when this process takes part in collectives: a process group exists and has more than one rank — or has one and
    SINGLE_RANK_COLLECTIVES is set."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or SINGLE_RANK_COLLECTIVES


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def init_from_env(backend=None, single_rank=False):
    """Initialise the default process group from torchrun-style env vars (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  WORLD_SIZE <= 1 creates
    no group (nothing to exchange) unless single_rank=True, which also switches the one-rank collectives on (SINGLE_RANK_COLLECTIVES)."""
    global SINGLE_RANK_COLLECTIVES
    if 'WORLD_SIZE' not in os.environ or 'RANK' not in os.environ or (dist.is_available() and dist.is_initialized()):
        return                    # not under a launcher (plain single process), or already initialised
    if int(os.environ['WORLD_SIZE']) <= 1:
        if not single_rank:
            return
        SINGLE_RANK_COLLECTIVES = True
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    kw = {}
    if backend == 'nccl':
        local = int(os.environ.get('LOCAL_RANK', 0))
        torch.cuda.set_device(local)
        kw['device_id'] = torch.device('cuda', local)
    dist.init_process_group(backend=backend, **kw)


def shard_range(n, r=None, w=None):
    """[start, end) of the n independent units (images, Z samples) that rank r owns; contiguous, sizes differ by at most one."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n, w)
    start = r * base + min(r, rem)
    return start, start + base + (1 if r < rem else 0)


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s weights (once, at construction: afterwards identical updates keep them equal)."""
    if not is_distributed():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src)
    # a collective writing into a parameter does not necessarily bump its version counter: tell the kernels' weight packs explicitly
    for m in module.modules():
        if hasattr(m, 'invalidate_packs'):
            m.invalidate_packs()


def _flat_view(grads):
    """One 1-D tensor aliasing all of `grads` when they tile a single contiguous range of one storage in order (the engine's batched
    weight-gradient launch writes every layer's dW / db into one flat buffer and hands out views of it), else None.  The returned view
    spans EXACTLY the given tensors — each must start where the previous one ends — so reducing / scaling it in place never touches a
    neighbour in the same buffer that belongs to somebody else (a parameter outside this reducer, another bucket)."""
    g0 = grads[0]
    if any(g is None or not g.is_contiguous() or g.dtype != g0.dtype or g.device != g0.device for g in grads):
        return None
    st = g0.untyped_storage()
    if any(g.untyped_storage().data_ptr() != st.data_ptr() for g in grads):
        return None
    off = g0.storage_offset()
    for g in grads:
        if g.storage_offset() != off:
            return None
        off += g.numel()
    return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(st, g0.storage_offset(), (off - g0.storage_offset(),))


class GradBucketAllReducer:
    """Averages .grad of the given parameters across ranks in flat buckets of ~bucket_mb megabytes (68 MB of generator gradients -> 3
    collectives: xGMI rings are per-link bound, so few large messages), issued back to back as async collectives that pipeline on the RCCL
    stream.  Nothing is overlapped with the backward pass, by construction of the backward: the weight gradients of ALL layers come out of ONE
    batched launch at its very end (esr_conv3x3_wgrad_batch), so there is no earlier moment at which a bucket is complete.  When the
    gradients of a bucket are consecutive views of one buffer — the flat buffer that launch writes — the collective runs in place on that
    buffer: no gather copy before, no scatter copy after."""

    def __init__(self, params, bucket_mb=32.0):
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []
        self.in_place = 0          # buckets of the last call that were reduced in place (diagnostics / tests)
        cur, cur_bytes, limit = [], 0, int(bucket_mb * 2 ** 20)
        for p in self.params:
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > limit or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)

    def __call__(self):
        if not is_distributed():
            return
        w = world_size()
        work = []
        self.in_place = 0
        for bucket in self.buckets:
            grads = [p.grad for p in bucket]
            flat = _flat_view(grads) if all(g is not None for g in grads) else None
            if flat is not None:
                self.in_place += 1
                work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, None))
                continue
            grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, bucket)]
            flat = torch.cat([g.reshape(-1) for g in grads])
            work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, bucket))
        for handle, flat, bucket in work:
            handle.wait()
            flat.div_(w)
            if bucket is None:
                continue
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[off:off + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p))
                off += n


class EarlyBucketReducer:
    """The generator's gradient exchange started from INSIDE its backward pass.  The engine cuts the batched weight-gradient launch into one
    launch per bucket (RRDBEngine.wgrad_exchange) and calls start(i, bucket) behind launch i: the bucket's in-place all-reduce is queued on the
    collective stream behind that launch and runs under the launches of the following buckets (xGMI is point to point: a 23 MB ring step and
    a 1.7 ms MFMA launch do not compete for anything).  finish() — behind the last launch — makes the compute stream wait for the collectives
    and scales the sums to means, before autograd hands the views on as .grad; only the LAST bucket's collective is exposed (0.5 of 1.5 ms at 8
    GPUs by DESIGN section 6's estimate).  Calling the object where GradBucketAllReducer used to be called does nothing when the exchange has
    already happened in this step, and falls back to the late, bucketed exchange otherwise (gradient accumulation, pixel-shuffle layers, a
    backward that did not run on the engine): same sums, same division, bit-identical gradients either way."""

    # OFF by default.  With the two-stream backward (RRDBEngine.wgrad_overlap: the bf16 / f16 one-plane gradient formats) the groups ARE the buckets
    # and the exchange costs +0.1-0.25 ms of a 25.2 ms configs[2] step at one rank under a launcher (profiles/r05_c3_exchange_ab.log, second block;
    # a one-rank RCCL group blocks the host inside every collective call until the stream it waits for has caught up, which is why the calls are
    # made at the end of the recorded pass) — against ~1 ms of all-reduce it can hide at 8 GPUs by the estimate of DESIGN section 6.  Where the
    # backward is ONE stream (hi+lo gradients, pixel-shuffle layers) the engine cuts the weight-gradient launch into one launch per bucket instead,
    # which costs 1.4-1.7 ms at one rank (each part ends in its own half-empty round of workgroups).  It has never run on more than one GPU: switch it
    # on (train.early_gradient_exchange, bench.py --early-exchange) where a measured scaling run shows the exposed exchange to be the larger number.
    ENABLED = False

    def __init__(self, params, bucket_mb=32.0):
        self.late = GradBucketAllReducer(params, bucket_mb)
        self.bucket_bytes = int(bucket_mb * 2 ** 20)
        self._work, self._done = [], False
        self.early_buckets = 0        # buckets of the last step that were exchanged from inside the backward (diagnostics / tests)

    buckets = property(lambda self: self.late.buckets)
    params = property(lambda self: self.late.params)

    @property
    def in_place(self):
        return self.early_buckets or self.late.in_place

    def start(self, i, t):
        if not is_distributed():
            return
        if i == 0:
            self._work = []
        self._work.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), t))

    def finish(self):
        w = world_size()
        for handle, t in self._work:
            handle.wait()             # (RCCL: the compute stream waits; the host does not)
            t.div_(w)
        self._done, self.early_buckets, self._work = bool(self._work), len(self._work), []

    def __call__(self):
        if self._done:
            self._done = False
            return
        self.early_buckets = 0
        self.late()


def broadcast_tensor(t, src=0):
    """Every rank gets rank `src`'s value of `t` (in place on a contiguous clone); identity when not distributed."""
    if not is_distributed():
        return t
    t = t.contiguous().clone()
    dist.broadcast(t, src)
    return t


def all_reduce_mean_scalar(value, device=None):
    """Mean over ranks of a python float (loss bookkeeping, e.g. the Z-search's batch-mean loss history)."""
    if not is_distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ('cuda' if dist.get_backend() == 'nccl' else 'cpu'))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / world_size()


def gather_scalars(t):
    """Every rank's copy of a small 1-D tensor of device scalars, as a [world, k] tensor on the same device (row r = rank r's values): the
    GAN bookkeeping that gates generator steps must come out the same on every rank, or the ranks disagree about which collectives come
    next.  One all-reduce of a one-row-per-rank buffer (every backend has that for device tensors).  Identity ([1, k]) when not distributed."""
    t = t.detach().reshape(1, -1)
    if not is_distributed():
        return t
    buf = torch.zeros(world_size(), t.size(1), dtype=t.dtype, device=t.device)
    buf[rank()] = t[0]
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf
