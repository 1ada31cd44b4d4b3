#!/usr/bin/env python
"""Headline benchmark: HR pixels / second of the RRDB-23 x4 generator + CEM (eval mode) forward on synthetic
32 x 3 x 128 x 128 fp32 batches per GPU (BASELINE.json configs[1]), weak scaling over N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5]

`--gpus N` with N > 1 launches its own ranks (python -m torch.distributed.run, one process per GPU, RCCL) unless it already runs
under a launcher (WORLD_SIZE set).  A step = one pass of the hot path over one batch already resident in HBM.  Rank 0 prints ONE
JSON line with the throughput, the roofline fraction of the dominant kernel (conv3x3, measured with HIP events on the launch
stream) and a CPU baseline (the oracle's CPU restatement, timed on this box's host cores on a bounded sample).

Arithmetic.  The headline runs the fp32-class precision 'split' (bf16x3): every product of every layer is evaluated from 16-bit
operands (bf16 hi + bf16 lo, three MFMAs, fp32 accumulate) — the scheme SURVEY.md section 7.4 sanctions for the fp32 configs.
'mixed' (fp16 operands, one MFMA per product inside the dense blocks: narrower than fp32-class arithmetic by the letter, 3e-5
from the fp32 oracle on this workload) is timed right after it and reported in the `alt_precision` block with BOTH of its
roofline fractions: against the fp32-equivalent algorithmic bytes and against the bytes it physically moves.

`--workload c3` times the generator + discriminator training step of configs[2] at its per-GPU shape (32 crops of 52x52, latent 3)
with the gradient all-reduce over RCCL when N > 1.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'explorable-super-resolution_amd'))
sys.path.insert(0, ROOT)

# RRDB-23 x4, batch 32, eval-padded 148^2 LR frames
NB, SF, BATCH, LR_SIZE = 23, 4, 32, 128
MARGIN_LR = 10
LR_PX_THROUGH_G = BATCH * (LR_SIZE + 2 * MARGIN_LR) ** 2          # 700,928
ALGO_BYTES_PER_LR_PX = 255436                                      # layer-granular fp32 read+write bytes per LR pixel through G
ALGO_BYTES_PER_FWD = LR_PX_THROUGH_G * ALGO_BYTES_PER_LR_PX        # 179.0 GB
FLOP_PER_LR_PX = 2 * 17926848
FLOP_PER_FWD = LR_PX_THROUGH_G * FLOP_PER_LR_PX                    # 25.13 TFLOP
HBM_PEAK = 8.0e12
N_CONV_LAUNCHES = 3 + NB * 15 + 2 + 1                              # 351
MFMA_TERMS = {'split': 3, 'f16x2': 2, 'mixed': 1.16}              # bf16/f16 MFMA instructions issued per product, network average

# what the `traffic: null` of the c3 / c4 / c5 roofline blocks means: HBM counters need rocprofv3 passes, which this process cannot take of itself
TRAFFIC_NOT_COLLECTED = ('not collected by this run (PMC counters need separate rocprofv3 --pmc passes: tools/pmc_workload.sh %s); committed '
                         'summaries of such passes, where taken: profiles/*_pmc*.json')

DTYPE = {'split': 'bf16x3 (bf16 hi+lo operands = 16 significand bits per operand in every product, three MFMAs, f32 accumulate)',
         'mixed': 'f16 (residual stream stored as hi+lo planes, hi+lo main-path weights, one-plane dense-block products, f32 accumulate)',
         'f16x2': 'f16x2', 'bf16': 'bf16', 'f16': 'f16'}
ARITHMETIC = {'split': 'split-bf16 (bf16x3) MFMA operands, fp32 accumulate, fp32 I/O', 'bf16': 'bf16 MFMA operands, fp32 accumulate',
              'f16': 'f16 MFMA operands, fp32 accumulate', 'f16x2': 'f16 weights x f16 hi+lo activations (2 MFMAs per product), fp32 accumulate',
              'mixed': 'fp32 I/O; f16 MFMA operands, fp32 accumulate: residual stream stored as hi+lo (22-bit) planes; hi+lo weights x hi+lo '
                       'activations (3 MFMAs) in the 6 convs outside the dense blocks; one-plane weights x hi planes (1 MFMA) and '
                       'one-plane intermediates inside the dense blocks'}


# ---------------------------------------------------------------------------------------------------------------- self-launch
def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(gpus, argv, port=None, script=None):
    """The command `bench.py --gpus N` re-executes itself with when it was started as a plain process: one rank per GPU on this node."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus), '--master-addr', '127.0.0.1',
            '--master-port', str(port or free_port()), script or os.path.abspath(__file__)] + list(argv)


def needs_self_launch(gpus, environ):
    return gpus > 1 and 'WORLD_SIZE' not in environ


def select_backend(environ):
    """'nccl' (= RCCL over xGMI) always; 'gloo' only under the test-only switch ESR_BENCH_SHARE_GPU=1 (several ranks on one GPU)."""
    return 'gloo' if environ.get('ESR_BENCH_SHARE_GPU') == '1' else 'nccl'


# ---------------------------------------------------------------------------------------------------------------- workload pieces
def build_model(device, nb=NB):
    import torch
    import CEM.CEMnet as CEMnet
    import models.modules.architecture as arch
    import models.networks as networks
    torch.manual_seed(0)
    cem = CEMnet.CEMnet(CEMnet.Get_CEM_Conf(SF))
    net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=nb, gc=32, upscale=SF, norm_type=None, act_type='leakyrelu', mode='CNA',
                       upsample_mode='upconv', latent_input=None, num_latent_channels=0)
    G = cem.WrapArchitecture_PyTorch(net)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        networks.init_weights(G, init_type='kaiming', scale=0.1)      # the reference's training init (networks.py:119)
    return cem, G.to(device).eval()


def cpu_model_string():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(G, cem):
    """the oracle (CPU restatement of the reference path, torch fp32 on the host cores) on ONE 128x128 image of the
    same workload (RRDB-23 x4 + CEM eval = 1/32 of a step) and on configs[0] (RRDB-3, 1x3x32x32).  Thread counts 8/16/32/64 (capped by the
    cores this process may use) are each tried once after a warm-up (oversubscribed torch CPU convolutions are slower); at the best count
    the figure is the MEDIAN of 3 further runs.  CPU model string and core counts are part of the record."""
    import statistics
    import torch
    from oracle import cem_oracle as co
    from oracle import rrdb_oracle as ro
    avail = len(os.sched_getaffinity(0))     # cores this process may actually use (cgroup/affinity aware)
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    taps = co.CEMTaps(SF)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(1, 3, LR_SIZE, LR_SIZE, generator=g)
    keep = {}

    def run(xin=x, state=sd, nb=NB):
        with torch.no_grad():
            xp = torch.nn.functional.pad(xin, (taps.margins_LR,) * 4, mode='replicate')
            gen = ro.rrdb_forward(state, xp, nb, SF, 0, prefix='generated_image_model.model')
            keep['gen'] = gen
            return co.cem_combine(xp, gen, taps, crop=True)

    def clock(fn, n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return ts, out
    sweep = {}
    for nt in sorted({min(n, avail) for n in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        run()
        sweep[nt] = clock(run, 1)[0][0]
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    ts, y = clock(run, 3)
    t = statistics.median(ts)
    gen = keep['gen']
    # configs[0]: RRDB-3 x4 + CEM on 1x3x32x32 (the reference's own CPU-runnable case), same thread count, random weights of that shape
    _, G3 = build_model('cpu', nb=3)
    sd3 = {k: v.detach() for k, v in G3.state_dict().items()}
    x3 = torch.rand(1, 3, 32, 32, generator=g)
    run(x3, sd3, 3)
    t3 = statistics.median(clock(lambda: run(x3, sd3, 3), 5)[0])
    return {'value': (SF * LR_SIZE) ** 2 / t, 'unit': 'HR pixels/s', 'cores': best, 'kind': 'port',
            'sample': '1 of the 32 images of a step (1x3x128x128 -> 512x512, RRDB-23 x4 + CEM eval): 1 warm-up, median of 3 runs at the best '
                      'thread count, %.2f s each' % t,
            'runs_s': [round(v, 3) for v in ts], 'thread_sweep_s': {str(k): round(v, 3) for k, v in sweep.items()}, 'cores_available': avail,
            'cpu_model': cpu_model_string(), 'logical_cpus': os.cpu_count(),
            'configs0': {'workload': 'configs[0]: RRDB-3 x4 + CEM, 1x3x32x32 -> 128x128', 'ms': t3 * 1e3, 'value': (SF * 32) ** 2 / t3, 'unit': 'HR pixels/s',
                         'sample': 'median of 5 runs, %d threads' % best}}, x, y, gen


def kernel_set_id():
    """Identity of the kernel sources a counter summary was taken on: sha256 over the conv / CEM kernel translation units and the shared
    header (the files whose code the static PMC fields of the headline describe), first 16 hex digits.  tools/summarise_*.py stamp it into
    every summary they write; this file prints a static PMC field only when the stamp equals the current sources'."""
    import hashlib
    h = hashlib.sha256()
    for f in ('esr_conv.hip', 'esr_cem.hip', 'esr_common.h'):
        path = os.path.join(ROOT, 'explorable-super-resolution_amd', 'csrc', f)
        if not os.path.exists(path):
            continue
        with open(path, 'rb') as fh:
            h.update(f.encode() + b'\0' + fh.read())
    return h.hexdigest()[:16]


def git_blob_id(path):
    """git's blob id of a file's current content (sha1 of 'blob <size>\\0' + bytes): lets a reader find the exact committed summary."""
    import hashlib
    data = open(path, 'rb').read()
    return hashlib.sha1(b'blob %d\0' % len(data) + data).hexdigest()


def _static_pmc(path, key):
    """(value, provenance) of a committed counter summary — value None when the summary carries no kernel-set stamp or was taken on other
    kernel sources than the ones in the tree now (the provenance then says so instead of a stale number being printed)."""
    d = json.load(open(path))
    prov = {'file': os.path.basename(path), 'git_blob': git_blob_id(path), 'kernel_set': d.get('kernel_set'), 'kernel_set_now': kernel_set_id()}
    if d.get('kernel_set') != prov['kernel_set_now']:
        prov['stale'] = 'csrc changed since these counters were taken: not printed'
        return None, prov
    return d.get(key), prov


def newest_pmc(precision):
    """HBM bytes per conv launch measured with rocprofv3 PMC passes of this same command on an earlier run (FETCH_SIZE / WRITE_SIZE, separate
    --pmc runs, FETCH_SIZE x2 on gfx950; tools/summarise_profiles.py): the newest summary committed under profiles/ for this precision."""
    import glob
    import re
    pmc = sorted((f for f in glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json'))
                  if ('mixed' in os.path.basename(f)) == (precision == 'mixed')),
                 key=lambda f: [int(n) for n in re.findall(r'\d+', os.path.basename(f))])    # r01_v9 < r01_v11 < r02_v1
    if not pmc or precision not in ('split', 'mixed'):
        return None, None
    return _static_pmc(pmc[-1], 'hbm_bytes_per_launch')


def newest_pmc_mfma(precision):
    """MFMA-pipe occupancy and effective shader clock of the conv launches from the newest committed counter summary of this command
    (tools/pmc_mfma.sh: SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE in separate rocprofv3 --pmc passes) — static, like roofline.traffic."""
    import glob
    import re
    # only summaries of the shipping kernel: r<round>_v<n>_<precision>_pmc_mfma.json (experiment variants carry a suffix after the precision)
    fs = sorted((f for f in glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_mfma.json'))
                 if re.fullmatch(r'r\d+_v\d+_%s_pmc_mfma\.json' % precision, os.path.basename(f))),
                key=lambda f: [int(n) for n in re.findall(r'\d+', os.path.basename(f))])
    if not fs:
        return None, None
    return _static_pmc(fs[-1], 'all_conv_launches')


# ---------------------------------------------------------------------------------------------------------------- main
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c2', choices=['c2', 'c3', 'c4', 'c5'],
                    help="c2 (default, the headline): configs[1] forward; c3: configs[2] G+D training step at its per-GPU shape, gradients all-reduced over RCCL; "
                         "c4: configs[3] Z-search iterations (64 Z samples of 512x512, sharded over the ranks); c5: configs[4] x8 inference with the "
                         "blurry_cubic_2.0 CEM kernel (16 images of 256x256, sharded over the ranks)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt-precision', action='store_true', help="skip the extra timing of the 'mixed' fp16 mode")
    ap.add_argument('--precision', default=None, choices=['mixed', 'split', 'f16x2', 'bf16', 'f16'],
                    help="c2 default 'split' (headline): bf16 hi+lo operands in every product; 'mixed': fp16 one-MFMA dense blocks, reported in the "
                         "alt_precision block.  c3 default 'bf16' (configs[2] names bf16)")
    ap.add_argument('--batch', type=int, default=BATCH, help='experiments only: the headline workload is batch 32')
    ap.add_argument('--early-exchange', action='store_true',
                    help="c3, N > 1 (or one rank under a launcher): exchange the generator's gradients bucket by bucket from INSIDE the backward pass "
                         "(one weight-gradient launch per bucket, each followed by its all-reduce: esr_hip.dist.EarlyBucketReducer) instead of after it")
    ap.add_argument('--no-extra-workloads', action='store_true',
                    help="skip the short configs[2] / configs[4] runs appended (after the headline's timed region) as `extra_workloads`")
    return ap.parse_args(argv)


def main(argv=None):
    global BATCH, LR_PX_THROUGH_G, ALGO_BYTES_PER_FWD, FLOP_PER_FWD
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if needs_self_launch(args.gpus, os.environ):
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus and os.environ.get('ESR_BENCH_SHARE_GPU') != '1':
            raise SystemExit('bench.py --gpus %d: this node exposes %d GPU(s)' % (args.gpus, have))
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # RCCL needs dmabuf IPC on this driver
        env.setdefault('OMP_NUM_THREADS', '8')
        sys.exit(subprocess.call(launch_command(args.gpus, argv), env=env))

    import torch
    if args.batch != BATCH:
        BATCH = args.batch
        LR_PX_THROUGH_G = BATCH * (LR_SIZE + 2 * MARGIN_LR) ** 2
        ALGO_BYTES_PER_FWD = LR_PX_THROUGH_G * ALGO_BYTES_PER_LR_PX
        FLOP_PER_FWD = LR_PX_THROUGH_G * FLOP_PER_LR_PX

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d was started by a launcher with WORLD_SIZE=%d: pass --gpus %d (or start it as a plain '
                         'process, it launches its own ranks)' % (args.gpus, world, world))
    # ESR_BENCH_SHARE_GPU=1 (tests only): the ranks share the GPUs there are and talk through gloo — RCCL refuses two ranks on one device.
    # It exercises this file's multi-rank logic on a one-GPU box; its numbers mean nothing.
    share = os.environ.get('ESR_BENCH_SHARE_GPU') == '1'
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    if world > 1 or 'WORLD_SIZE' in os.environ:        # under a launcher the collectives run whatever the world size is (one rank: RCCL all the same)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        from esr_hip import dist as D
        D.SINGLE_RANK_COLLECTIVES = True                # under a launcher this file runs the collectives whatever N is (N = 1: RCCL all the same)
        if select_backend(os.environ) == 'gloo':
            dist.init_process_group(backend='gloo')
        else:
            dist.init_process_group(backend='nccl', device_id=dev)

    def rank_info():
        """What this rank runs on — lets whoever reads the line confirm N distinct GPUs and the RCCL backend."""
        prop = torch.cuda.get_device_properties(dev)
        info = {'rank': rank, 'device_index': dev_index, 'device': prop.name, 'gcn_arch': getattr(prop, 'gcnArchName', None),
                'pci_bus_id': '%04x:%02x:%02x' % (getattr(prop, 'pci_domain_id', 0), getattr(prop, 'pci_bus_id', 0), getattr(prop, 'pci_device_id', 0)),
                'uuid': str(getattr(prop, 'uuid', '')), 'backend': dist.get_backend() if dist is not None else None}
        if dist is not None and info['backend'] == 'nccl':
            try:
                info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:                     # the version query is informational only
                info['rccl_version'] = 'unavailable: %s' % e
        if dist is None:
            return [info]
        infos = [None] * world
        dist.all_gather_object(infos, info)
        return infos

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        """(max over ranks, per-rank list) of a duration in seconds"""
        if dist is None:
            return dt, [dt]
        t = torch.zeros(world, device=dev, dtype=torch.float64)       # one slot per rank, summed: an all-gather that every backend has for device tensors
        t[rank] = dt
        dist.all_reduce(t)
        per = [float(v) for v in t.tolist()]
        return max(per), per

    if args.workload == 'c3':
        out = run_c3(args, dev, rank, world, dist, sync, max_over_ranks)
    elif args.workload == 'c4':
        out = run_c4(args, dev, rank, world, dist, sync, max_over_ranks)
    elif args.workload == 'c5':
        out = run_c5(args, dev, rank, world, dist, sync, max_over_ranks)
    else:
        out = run_c2(args, dev, rank, world, dist, sync, max_over_ranks)
        if args.batch == BATCH and not args.no_extra_workloads:
            extra = extra_workloads(args, dev, rank, world, dist, sync, max_over_ranks)      # every rank takes part (barriers, collectives)
            if rank == 0:
                out['extra_workloads'] = extra
    infos = rank_info()
    if rank == 0:
        out['ranks'] = infos
        out['distinct_gpus'] = len({(i['pci_bus_id'], i['uuid']) for i in infos})
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_c2(args, dev, rank, world, dist, sync, max_over_ranks):
    import torch
    precision = args.precision or 'split'
    cem, G = build_model(dev)
    net = G.generated_image_model
    net.set_precision(precision)
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.rand(BATCH, 3, LR_SIZE, LR_SIZE, generator=g).to(dev)      # synthetic LR batch, resident in HBM

    eng = net.engine
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None):
        with torch.no_grad():
            if i is None:
                return G(x)
            # bracket the generator's conv launches (same stream as the kernels: torch's current stream)
            eng._ev = (ev0[i], ev1[i])
            y = G(x)
            eng._ev = None
            return y

    def timed(with_events):
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            y = step(i if with_events else None)
        sync()
        return max_over_ranks(time.perf_counter() - t0) + (y,)

    for _ in range(args.warmup):
        step()
    dt, per_rank, y = timed(True)
    ms_per_step = dt / args.steps * 1e3
    hr_px = BATCH * (SF * LR_SIZE) ** 2
    value = world * hr_px * args.steps / dt
    if rank != 0:
        if precision == 'split' and not args.no_alt_precision:      # every rank takes part in the alt timing's barriers
            net.set_precision('mixed')
            for _ in range(2):
                step()
            timed(False)
        return None

    traffic, traffic_src = newest_pmc(precision) if BATCH == 32 else (None, None)
    conv_ms = sorted(a.elapsed_time(b) for a, b in zip(ev0, ev1))[len(ev0) // 2]      # generator span per step (ms)
    t_launch = conv_ms * 1e-3 / N_CONV_LAUNCHES
    achieved = (ALGO_BYTES_PER_FWD / N_CONV_LAUNCHES) / t_launch
    # CEM downsample-consistency of the timed output (interior) with the HIP downsampler
    with torch.no_grad():
        d = G.DownscaleOP(y)
    m = int(cem.invalidity_margins_LR)
    cons = float(((d - x)[:, :, m:-m, m:-m] ** 2).mean().sqrt())
    terms = MFMA_TERMS.get(precision, 1)
    out = {
        'metric': 'HR pixels/sec (RRDB-23 x4, 128->512, bs32 per GPU, fwd + CEM)', 'value': value, 'unit': 'HR pixels/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE[precision], 'data': 'synthetic',
        'config': {'workload': 'configs[1]: RRDB-23 x4 SR forward, batch 32 of 128x128 per GPU, fp32 I/O, CEM wrap (eval: G runs on 148x148)',
                   'arithmetic': ARITHMETIC[precision], 'global_batch': BATCH * world,
                   'parallelism': 'dp%d (independent image shards, no data-path collective)' % world},
        'world_size_seen': world, 'ms_per_step_per_rank': [p / args.steps * 1e3 for p in per_rank],
        'roofline': {'bound': 'hbm', 'kernel': 'conv3x3_tile_kernel (351 launches per forward, all instantiations)', 'achieved': achieved / 1e9, 'peak': HBM_PEAK / 1e9,
                     'unit': 'GB/s', 'frac': achieved / HBM_PEAK,
                     # PMC traffic is NOT re-measured by this run (counters need rocprofv3): it is the committed summary of the same command
                     'traffic': traffic, 'traffic_static': True, 'traffic_source': traffic_src,
                     'algorithmic_bytes_per_launch': ALGO_BYTES_PER_FWD / N_CONV_LAUNCHES, 'avg_launch_ms': t_launch * 1e3,
                     'generator_ms_per_step': conv_ms,
                     'mfma_fp32_equiv_tflops': FLOP_PER_FWD / (conv_ms * 1e-3) / 1e12,
                     'mfma_bf16_issue_frac': terms * FLOP_PER_FWD / (conv_ms * 1e-3) / 2.5e15,
                     # sustained dense bf16 MFMA rate measured on this part with random operands (power-limited clock, profiles/microbench/mfma_peak.hip)
                     'mfma_bf16_issue_frac_of_measured_1.79PF': terms * FLOP_PER_FWD / (conv_ms * 1e-3) / 1.79e15},
        'cem_consistency_rmse_interior': cons,
    }
    if BATCH == 32:
        # The wrapping CEM projection alone (A6-A8: out = g + U(K(x - D(g))), cropped), timed live with events over 20 back-to-back projections of the
        # generator's output, against SURVEY section 8(a) A8's algorithmic minimum (read g 134.6 MB + x 8.4 MB, write out 100.7 MB)
        from esr_hip import cem_ops
        with torch.no_grad():
            gimg = net(x, pad=m)
            proj = lambda: cem_ops.project(x, gimg, G.DownscaleOP.taps(), G.Conv_LR_with_Inv_hTh_OP.taps(), G.Upscale_OP.taps(), SF, G.pre_stride,
                                           lr_pad=m, crop=SF * m)
            for _ in range(3):
                proj()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(20):
                proj()
            c1.record()
            torch.cuda.synchronize()
            del gimg
        cem_s = c0.elapsed_time(c1) / 20 * 1e-3
        cem_bytes = 134.6e6 + 8.4e6 + 100.7e6
        out['cem_projection'] = {'kernels': 'cem_downscale_wave_kernel, cem_lrfilter_wave_kernel, cem_upscale_wave_kernel (3 launches)', 'us': cem_s * 1e6,
                                 'bound': 'hbm', 'algorithmic_bytes': cem_bytes, 'achieved': cem_bytes / cem_s / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                                 'frac': cem_bytes / cem_s / HBM_PEAK, 'traffic': None,
                                 'traffic_source': 'not collected by this run; committed counters of 20 such projections: profiles/r06_cem_wave_pmc.json (0.389 GB per projection)'}
    if traffic:
        out['roofline']['frac_physical_static'] = traffic / t_launch / HBM_PEAK      # bytes the mode really moves (static PMC figure) / measured time
    pm, pm_src = newest_pmc_mfma(precision) if BATCH == 32 else (None, None)
    if pm_src and not pm:
        out['roofline']['mfma_counters_source'] = pm_src         # stale summary: the provenance says why nothing is printed
    if pm:
        # counters of an earlier rocprofv3 run of this command (a profiled run clocks ~2 % lower): fraction of cycles the MFMA pipes were busy
        # and GRBM_GUI_ACTIVE / wall time; 32 busy cycles per v_mfma_f32_32x32x16
        out['roofline'].update({'mfma_busy_frac': pm.get('mfma_busy_frac'), 'effective_clock_ghz': pm.get('effective_clock_ghz'),
                                'mfma_issue_pflops_bf16_pmc': pm.get('mfma_issue_pflops_bf16'), 'mfma_counters_static': True, 'mfma_counters_source': pm_src})
    gen_err = None
    if not args.no_cpu_baseline and world == 1:       # the CPU baseline is timed on rank 0 of the single-GPU run only
        cb, xs, ys, gen_ref = cpu_baseline(G, cem)      # gen_ref: the oracle's generator output on the padded frame
        out['cpu_baseline'] = cb
        with torch.no_grad():
            yg = G(xs.to(dev)).cpu()
        out['rel_l2_vs_cpu_oracle'] = float((yg - ys).norm() / ys.norm())
        out['rel_max_vs_cpu_oracle'] = float((yg - ys).abs().max() / ys.abs().max())

        def gen_err(rel_max=False):
            with torch.no_grad():
                gg = net(xs.to(dev), pad=MARGIN_LR).cpu()
            if rel_max:
                return float((gg - gen_ref).abs().max() / gen_ref.abs().max())
            return float((gg - gen_ref).norm() / gen_ref.norm())
        out['generator_rel_l2_vs_cpu_oracle'] = gen_err()       # the generator alone (the CEM output above is dominated by the LR content)
        out['generator_rel_max_vs_cpu_oracle'] = gen_err(True)
    if precision == 'split' and not args.no_alt_precision:
        # beside the headline: the same workload in 'mixed' (fp16, one MFMA per product inside the dense blocks — narrower than fp32-class
        # by the letter), with its own parity figure and BOTH roofline fractions
        net.set_precision('mixed')
        for _ in range(2):
            step()
        dta, _, _ = timed(False)
        tm, tm_src = newest_pmc('mixed') if BATCH == 32 else (None, None)
        alt = {'mode': 'mixed', 'dtype': DTYPE['mixed'], 'ms_per_step': dta / args.steps * 1e3, 'value': world * hr_px * args.steps / dta, 'unit': 'HR pixels/s',
               'roofline_frac_fp32_equiv_bytes': ALGO_BYTES_PER_FWD / (dta / args.steps) / HBM_PEAK,
               'roofline_frac_physical_static': (tm * N_CONV_LAUNCHES / (dta / args.steps) / HBM_PEAK) if tm else None,
               'physical_bytes_per_forward_static': tm * N_CONV_LAUNCHES if tm else None, 'traffic_source': tm_src,
               'note': ARITHMETIC['mixed'] + '; whole-step time (the headline roofline uses the generator span)'}
        if gen_err is not None:
            alt['generator_rel_l2_vs_cpu_oracle'] = gen_err()
            alt['generator_rel_max_vs_cpu_oracle'] = gen_err(True)
        out['alt_precision'] = alt
        net.set_precision(precision)
    return out


def extra_workloads(args, dev, rank, world, dist, sync, max_over_ranks):
    """The default run only, AFTER the headline's timed region and its alt-precision / CPU-baseline legs, each block through the same function
    its own `--workload` run uses — so the line the driver records carries driver-observed numbers for the configs the headline does not time.
    N = 1: a short run (3 warm-up + 5 steps) of configs[2] (`c3`: G+D training step at the per-GPU shape, bf16), of configs[4] (`c5`: x8
    inference, f16, blurry_cubic_2.0 CEM kernel) and (1 warm-up + 2 timed iterations) of configs[3] (`c4`: the Z search, 64 samples of 512x512,
    'split').  N > 1: configs[2] only — the one workload with data-path collectives — with its communication diagnosis (`comm`: the step with the
    gradient exchange after the backward, from inside it, and with no exchange at all; run_c3), so that the first multi-GPU record answers what
    the exchange costs and which form to default to.  Nothing here touches a headline key; a failure is recorded as {'error': ...} for that
    block instead of costing the line (ranks fail together or not at all: the blocks' collectives are symmetric)."""
    import copy
    import gc
    import torch
    res = {}
    blocks = (('c3', run_c3, 5, 3), ('c5', run_c5, 5, 3), ('c4', run_c4, 2, 1)) if world == 1 else (('c3', run_c3, 5, 3),)
    for name, fn, steps, warmup in blocks:
        a = copy.copy(args)
        a.workload, a.steps, a.warmup, a.precision, a.comm_diag = name, steps, warmup, None, world > 1
        gc.collect()
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        try:
            r = fn(a, dev, rank, world, dist, sync, max_over_ranks)
            if r is not None:
                r = {k: r[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'dtype', 'phases_ms', 'roofline',
                                       'cem_consistency_rmse_interior', 'peak_memory_GB', 'generator_backward', 'gradient_exchange', 'comm',
                                       'seconds_for_200_iterations', 'activation_stash') if k in r} | {'workload': r['config']['workload']}
        except Exception as e:                      # noqa: BLE001 — the headline line must survive whatever an appended block does
            r = {'error': '%s: %s' % (type(e).__name__, e)}
        if r is not None:
            r['wall_s'] = time.perf_counter() - t0
        res[name] = r
    return res


def run_c3(args, dev, rank, world, dist, sync, max_over_ranks):
    """configs[2] at its per-GPU shape: SRRaGANModel.optimize_parameters() (G forward/backward, Discriminator_VGG_128 forward / backward / the
    WGAN-GP penalty's double backward and both Adam steps on the library's kernels) on 32 crops of 52x52 (HR 208x208, latent 3) per GPU; G and D
    gradients are all-reduced over RCCL when N > 1."""
    import contextlib
    import io
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import bench_paths
    import models
    from esr_hip import dist as D
    precision = args.precision or 'bf16'
    if getattr(args, 'early_exchange', False):
        D.EarlyBucketReducer.ENABLED = True
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = models.create_model(bench_paths.make_opt(True, with_D=True))
    if precision != 'split':
        model.netG.generated_image_model.set_precision(precision)
    d_bf16 = precision == 'bf16'
    if d_bf16:
        model.D_dtype = torch.bfloat16          # configs[2] names bf16: the critic's convolutions run under bf16 autocast (fp32 parameters and losses)
    B = 32 if args.batch == BATCH else args.batch
    g = torch.Generator().manual_seed(2000 + rank)
    data = {'LR': torch.rand(B, 3, 52, 52, generator=g).to(dev), 'HR': torch.rand(B, 3, 208, 208, generator=g).to(dev),
            'Z': (torch.rand(B, 3, 208, 208, generator=g) * 2 - 1).to(dev)}
    for _ in range(args.warmup):
        model.feed_data(data); model.optimize_parameters()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):                  # the timed region: free-running steps (no host synchronisation inside a step)
        model.feed_data(data); model.optimize_parameters()
    sync()
    dt, per_rank = max_over_ranks(time.perf_counter() - t0)
    # per-phase GPU times from a few extra steps OUTSIDE the timed region: the phase timers synchronise at the end of every step, which
    # costs the overlap of the host's launch work with the GPU (49-50 instead of 45-46 ms per step)
    nphase = min(args.steps, 5)
    model.timing = {}
    for _ in range(nphase):
        model.feed_data(data); model.optimize_parameters()
    sync()
    timing_kept = dict(model.timing)
    comm = None
    if getattr(args, 'comm_diag', False) and dist is not None:
        # What the gradient exchange costs and which form to default to, answered by ONE record: the same step (a) with
        # the exchange after the backward (the timed region above), (b) started from inside the backward (EarlyBucketReducer), (c) with no
        # exchange at all — the ranks' replicas then drift apart, which is why this comes last, on a model nobody uses afterwards.
        # exposed_comm_ms = (a) - (c): the part of the collectives that the step's own work does not hide.
        model.timing = None

        # (ESR_BENCH_SHARE_GPU, the tests' ranks-on-one-GPU switch: gloo stages every bucket through the host, ~1 s per step — two steps per arm)
        ARM_STEPS, ARM_WARM = (2, 1) if os.environ.get('ESR_BENCH_SHARE_GPU') == '1' else (5, 2)

        def arm(nsteps=ARM_STEPS, nwarm=ARM_WARM):
            for _ in range(nwarm):
                model.feed_data(data); model.optimize_parameters()
            sync()
            t = time.perf_counter()
            for _ in range(nsteps):
                model.feed_data(data); model.optimize_parameters()
            sync()
            return max_over_ranks(time.perf_counter() - t)[0] / nsteps * 1e3
        late_ms = dt / args.steps * 1e3
        was = model.grad_reducer.ENABLED
        model.grad_reducer.ENABLED = not was
        other_ms = arm()
        other_buckets = model.grad_reducer.early_buckets
        model.grad_reducer.ENABLED = was
        early_ms, after_ms = (late_ms, other_ms) if was else (other_ms, late_ms)
        real_G, real_D = model.grad_reducer, getattr(model, 'grad_reducer_D', None)

        class NoExchange:                          # what SRRaGANModel asks of a reducer (see _G_backward), doing nothing
            ENABLED, params, buckets, early_buckets, in_place = False, [], [], 0, 0

            def __call__(self):
                pass
        model.grad_reducer = NoExchange()
        if real_D is not None:
            model.grad_reducer_D = NoExchange()
        none_ms = arm()
        model.grad_reducer, model.grad_reducer_D = real_G, real_D
        comm = {'ms_per_step_exchange_after_backward': after_ms, 'ms_per_step_exchange_inside_backward': early_ms, 'ms_per_step_no_exchange': none_ms,
                'exposed_comm_ms': (early_ms if was else after_ms) - none_ms, 'exposed_comm_ms_after_backward': after_ms - none_ms,
                'exposed_comm_ms_inside_backward': early_ms - none_ms, 'early_buckets': other_buckets if not was else model.grad_reducer.early_buckets,
                'timed_region_used': 'inside the backward' if was else 'after the backward',
                'G_gradient_bytes': int(sum(p.numel() * 4 for p in real_G.params)), 'D_gradient_bytes': int(sum(p.numel() * 4 for p in real_D.params)) if real_D is not None else 0,
                'backend': dist.get_backend(), 'steps_per_arm': ARM_STEPS,
                'note': 'same model, same data, arms back to back: (the timed region), the other exchange form (%d warm-up + %d steps), no exchange '
                        '(the same; replicas drift: diagnosis only).  exposed_comm_ms = timed form minus no exchange' % (ARM_WARM, ARM_STEPS)}
    if rank != 0:
        return None
    log = model.get_current_log()
    split_ms = {k: v / nphase for k, v in (getattr(model, 'timing', None) or timing_kept).items()}
    flop_g = 3 * 32 * 52 * 52 * 2 * 18316944 * (B / 32)        # fwd + dgrad + wgrad of G (lat 3), SURVEY §8(d)
    # layer-granular bytes of the same three passes (49,268 + 15,728 elements per LR pixel for RRDB-23 x4 lat 3), at the bytes
    # per element the precision stores (one 16-bit plane, or hi + lo)
    eng_g = getattr(model.netG.generated_image_model, 'engine', None)
    bytes_g = 3 * B * 52 * 52 * (49268 + 15728) * (4 if precision in ('split', 'mixed') else 2)
    return {'metric': 'LR crops/sec (RRDB-23 x4 lat 3 G + Discriminator_VGG_128 WGAN-GP step, 32 x 52x52 per GPU)', 'value': world * B * args.steps / dt, 'unit': 'LR crops/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': DTYPE.get(precision, precision) + ' (generator); discriminator ' + (('%s on the HIP kernels (esr_hip/critic.py)' % ('bf16' if d_bf16 else 'bf16x3'))
                                                                              if model.D_engine is not None else ('%s on MIOpen' % ('bf16 autocast' if d_bf16 else 'fp32'))), 'data': 'synthetic',
            'config': {'workload': 'configs[2] per-GPU shape: SRRaGANModel.optimize_parameters(), G+D step, %d crops of 52x52 (HR 208x208, latent 3) per GPU' % B,
                       'global_batch': B * world, 'parallelism': 'dp%d (G and D gradients all-reduced over RCCL)' % world},
            'world_size_seen': world, 'ms_per_step_per_rank': [p / args.steps * 1e3 for p in per_rank],
            'gradient_exchange': ('inside the backward, %d buckets' % model.grad_reducer.early_buckets) if model.grad_reducer.early_buckets else
                                 ('after the backward, %d buckets' % len(model.grad_reducer.buckets) if dist is not None else 'none (one process)'),
            'generator_backward': ('weight gradients in %d groups, %d of them on a second stream under the data-gradient chain' % (eng_g.wgrad_overlap, eng_g.wgrad_overlap - 1))
                                  if (eng_g is not None and isinstance(eng_g.wgrad_overlap, int) and eng_g.wgrad_overlap >= 2 and precision != 'split') else 'one weight-gradient launch behind the data-gradient chain',
            'phases_ms': split_ms, 'comm': comm, 'losses': {k: float(v) for k, v in log.items() if isinstance(v, (int, float))},
            'roofline': {'bound': 'mfma', 'kernel': 'generator convs (forward + data gradient + weight gradient)', 'achieved': MFMA_TERMS.get(precision, 1) * flop_g / (dt / args.steps) / 1e12,
                         'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': MFMA_TERMS.get(precision, 1) * flop_g / (dt / args.steps) / 2.5e15, 'traffic': None,
                         'traffic_source': TRAFFIC_NOT_COLLECTED % 'c3',
                         'generator_algorithmic_bytes_per_step': bytes_g, 'generator_hbm_frac_of_step': bytes_g / (dt / args.steps) / 8.0e12,
                         'note': 'MFMA issue rate (and, beside it, the layer-granular HBM bytes) of the generator over the WHOLE step time: D, optimizers '
                                 'and all-reduce are in the denominator; per-owner GPU time of a step: profiles/*_c3_*_step_kernels.csv'}}


def run_c4(args, dev, rank, world, dist, sync, max_over_ranks):
    """configs[3]: latent search — Z_optimizer.optimize() iterations on 64 Z samples of 512x512 for one 128x128 LR image (RRDB-23 x4 lat 3 +
    CEM eval: G on 148x148), objective STD_increase, Adam on Z.  One step = one iteration = forward with graph, CEM, objective, data gradient
    through the frozen generator, Adam.  The Z samples are independent: sharded over the ranks, no data-path collective (one scalar
    all-reduce per iteration for the loss history)."""
    import contextlib
    import io
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import bench_paths
    import models
    from esr_hip import dist as D
    from Z_optimization import Z_optimizer
    precision = args.precision or 'split'
    B = 64 if args.batch == BATCH else args.batch
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = models.create_model(bench_paths.make_opt(False))
    net = model.netG.generated_image_model
    net.set_precision(precision)
    lr = torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(3000)).to(dev)
    lo, hi = D.shard_range(B)
    model.feed_data({'LR': lr.expand(hi - lo, -1, -1, -1), 'Z': torch.zeros(hi - lo, 3, 512, 512, device=dev)}, need_GT=False)
    model.test()
    with contextlib.redirect_stdout(io.StringIO()):
        zo = Z_optimizer(objective='STD_increase', Z_size=[512, 512], model=model, Z_range=1, max_iters=max(args.warmup, 1), data={'LR': lr, 'STD_increment': 0.01},
                         initial_LR=0.1, batch_size=B)
        zo.optimize()                           # warm-up iterations (records the launch lists)
        zo.max_iters = args.steps
        torch.cuda.reset_peak_memory_stats()
        sync()
        t0 = time.perf_counter()
        zo.optimize()
        sync()
    dt, per_rank = max_over_ranks(time.perf_counter() - t0)
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    if rank != 0:
        return None
    t_it = dt / args.steps
    lr_px = B * 148 * 148                                    # LR pixels through G per iteration (1,401,856 at B = 64)
    flop = 2 * 2 * 18316944 * lr_px                          # forward + data gradient, RRDB-23 x4 lat 3
    algo_bytes = 2 * 259984 * lr_px                          # layer-granular fp32-equivalent bytes of the two passes
    terms = MFMA_TERMS.get(precision, 1)
    return {'metric': 'HR pixels/sec of Z-search iterations (RRDB-23 x4 lat 3 + CEM, 64 Z samples of 512x512, forward + data gradient + Adam on Z)',
            'value': B * 512 * 512 * args.steps / dt, 'unit': 'HR pixels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': t_it * 1e3,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': DTYPE.get(precision, precision), 'data': 'synthetic',
            'config': {'workload': 'configs[3]: Z_optimizer.optimize(), STD_increase, %d Z samples of 512x512 on one 128x128 LR image (G on 148x148), Adam lr 0.1; '
                                   'the reference config runs 200 such iterations' % B,
                       'global_batch': B, 'parallelism': 'dp%d (Z samples sharded, no data-path collective)' % world},
            'world_size_seen': world, 'ms_per_step_per_rank': [p / args.steps * 1e3 for p in per_rank],
            'seconds_for_200_iterations': 200 * t_it, 'loss_first_last': [float(zo.loss_values[0]), float(zo.loss_values[-1])], 'iterations_kept': len(zo.loss_values),
            'peak_memory_GB': peak, 'activation_stash': net.engine.stash,
            'roofline': {'bound': 'mfma', 'kernel': 'conv3x3_tile_kernel (forward and data gradient of the frozen generator)', 'achieved': terms * flop / t_it / 1e12 / world,
                         'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': terms * flop / t_it / 2.5e15 / world, 'traffic': None,
                         'traffic_source': TRAFFIC_NOT_COLLECTED % 'c4',
                         'fp32_equiv_tflops': flop / t_it / 1e12 / world, 'algorithmic_bytes_per_iteration': algo_bytes,
                         'hbm_frac': algo_bytes / t_it / HBM_PEAK / world,
                         'note': 'MFMA issue rate over the WHOLE iteration (CEM, objective, Adam and the host in the denominator); per GPU'}}


def run_c5(args, dev, rank, world, dist, sync, max_over_ranks):
    """configs[4]: RRDB-23 x8 inference with a non-bicubic CEM kernel ('blurry_cubic_2.0': 45^2 / 35^2 taps, margin 12), 16 images of 256x256 ->
    2048x2048, fp16 operands as the config names (--precision split for the fp32-class mode); images sharded over the ranks."""
    import contextlib
    import io
    import torch
    import CEM.CEMnet as CEMnet
    import models.modules.architecture as arch
    import models.networks as networks
    from esr_hip import dist as D
    precision = args.precision or 'f16'
    B = 16 if args.batch == BATCH else args.batch
    torch.manual_seed(0)
    cem = CEMnet.CEMnet(CEMnet.Get_CEM_Conf(8), upscale_kernel='blurry_cubic_2.0')
    net = arch.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=NB, gc=32, upscale=8, norm_type=None, act_type='leakyrelu', mode='CNA', upsample_mode='upconv',
                       latent_input=None, num_latent_channels=0)
    G = cem.WrapArchitecture_PyTorch(net)
    with contextlib.redirect_stdout(io.StringIO()):
        networks.init_weights(G, init_type='kaiming', scale=0.1)
    G = G.to(dev).eval()
    net.set_precision(precision)
    lo, hi = D.shard_range(B)
    x = torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(4000))[lo:hi].to(dev)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    with torch.no_grad():
        for _ in range(args.warmup):
            y = G(x)
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            net.engine._ev = (ev0[i], ev1[i])
            y = G(x)
        net.engine._ev = None
        sync()
        dt, per_rank = max_over_ranks(time.perf_counter() - t0)
        d = G.DownscaleOP(y)
    if rank != 0:
        return None
    m = int(cem.invalidity_margins_LR)
    cons = float(((d - x)[..., m:-m, m:-m] ** 2).mean().sqrt())
    t = dt / args.steps
    conv_ms = sorted(a.elapsed_time(b) for a, b in zip(ev0, ev1))[len(ev0) // 2]
    side = 256 + 2 * m
    lr_px = (hi - lo) * side * side                                    # per rank
    bytes_per_px = 313356 // (1 if precision in ('split', 'mixed') else 2)       # RRDB-23 x8, fp32 / fp16 elements
    algo = lr_px * bytes_per_px
    flop = 2 * 22138560 * lr_px
    terms = MFMA_TERMS.get(precision, 1)
    n_launch = 3 + NB * 15 + 3 + 1
    return {'metric': 'HR pixels/sec (RRDB-23 x8, 256->2048, 16 images, fwd + CEM with the blurry_cubic_2.0 kernel)', 'value': B * 2048 * 2048 * args.steps / dt,
            'unit': 'HR pixels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': t * 1e3, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': DTYPE.get(precision, precision), 'data': 'synthetic',
            'config': {'workload': "configs[4]: RRDB-23 x8 SR forward, %d images of 256x256 -> 2048x2048, CEM with upscale_kernel='blurry_cubic_2.0' (eval: G on %dx%d)" % (B, side, side),
                       'arithmetic': ARITHMETIC.get(precision, precision), 'global_batch': B, 'parallelism': 'dp%d (independent image shards, no data-path collective)' % world},
            'world_size_seen': world, 'ms_per_step_per_rank': [p / args.steps * 1e3 for p in per_rank], 'cem_consistency_rmse_interior': cons,
            'cem_kernel_sizes': [int(cem.ds_kernel.shape[0]), int(cem.inv_hTh.shape[0])], 'peak_memory_GB': torch.cuda.max_memory_allocated() / 2 ** 30,
            'roofline': {'bound': 'hbm', 'kernel': 'conv3x3_tile_kernel (%d launches per forward)' % n_launch, 'achieved': algo / (conv_ms * 1e-3) / 1e9, 'peak': HBM_PEAK / 1e9,
                         'unit': 'GB/s', 'frac': algo / (conv_ms * 1e-3) / HBM_PEAK, 'traffic': None, 'traffic_source': TRAFFIC_NOT_COLLECTED % 'c5',
                         'algorithmic_bytes_per_forward_per_gpu': algo,
                         'generator_ms_per_step': conv_ms, 'avg_launch_ms': conv_ms / n_launch,
                         'mfma_issue_frac': terms * flop / (conv_ms * 1e-3) / 2.5e15,
                         'note': 'layer-granular bytes at the element size the precision stores (SURVEY 8(d): 313,356 B per LR pixel in fp32, half in fp16); rank 0'}}


if __name__ == '__main__':
    main()

