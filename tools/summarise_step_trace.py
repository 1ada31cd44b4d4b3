#!/usr/bin/env python
"""Steady-state per-step kernel summary of a training-step profile (rocprofv3 --kernel-trace of `bench.py --workload c3`).

rocprofv3's own --stats table covers the whole process, including MIOpen's one-off kernel search / naive fallbacks during the first steps.
This script cuts the trace at a kernel launched exactly once per step (`--marker`, default the CEM's down-scaling pass of the generator
forward; the batched weight-gradient kernel no longer qualifies since the critic uses it too) and aggregates the LAST `--steps` complete steps: per kernel name, launches per step, average duration, milliseconds per step.

    python tools/summarise_step_trace.py --trace <dir with *_kernel_trace.csv> --tag r02_c3_bf16 [--steps 4]
"""
import argparse
import collections
import csv
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:150]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--trace', required=True)
    ap.add_argument('--tag', required=True)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--marker', default='cem_downscale_sep_kernel')
    a = ap.parse_args()
    f = glob.glob(os.path.join(a.trace, '**', '*_kernel_trace.csv'), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if a.marker in r['Kernel_Name']]
    assert len(marks) > a.steps, 'not enough steps in the trace'
    lo, hi = marks[-a.steps - 1], marks[-1]
    window = rows[lo:hi]
    span = (int(window[-1]['End_Timestamp']) - int(window[0]['Start_Timestamp'])) / 1e6 / a.steps
    per = collections.OrderedDict()
    for r in window:
        k = short(r['Kernel_Name'])
        d = per.setdefault(k, [0, 0])
        d[0] += 1
        d[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    busy = sum(d[1] for d in per.values()) / 1e6 / a.steps
    out = os.path.join(ROOT, 'profiles', a.tag + '_step_kernels.csv')
    with open(out, 'w') as fo:
        fo.write('# last %d steps of the trace: %.2f ms wall per step, %.2f ms of kernel time per step\n' % (a.steps, span, busy))
        fo.write('kernel,launches_per_step,avg_us,ms_per_step,percent_of_kernel_time\n')
        for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            fo.write('"%s",%.1f,%.1f,%.3f,%.1f\n' % (k, n / a.steps, t / n / 1e3, t / 1e6 / a.steps, 100 * t / 1e6 / a.steps / busy))
        # who owns the time: the library's kernels, the critic's convolutions (MIOpen / CK), and torch's element-wise / reduction / copy glue
        groups = collections.OrderedDict((g, 0) for g in ('esr_hip conv / wgrad / pack / CEM', 'MIOpen + CK convolutions', 'MIOpen batch norm',
                                                            'layout transposes', 'torch element-wise / reductions / copies', 'other'))
        for k, (n, t) in per.items():
            if re.match(r'(conv3x3_|pack_|cem_|act_|unpack_|grad_|pixel_|soft_hist|zero_|bn_|adam_|wgrad_|img_|tv_|struct)', k): g = 'esr_hip conv / wgrad / pack / CEM'
            elif 'igemm' in k or 'grouped_conv' in k or 'Conv' in k or 'gemm' in k.lower() or 'Cijk' in k: g = 'MIOpen + CK convolutions'
            elif 'BatchNorm' in k: g = 'MIOpen batch norm'
            elif 'transpose' in k: g = 'layout transposes'
            elif k.startswith('at::') or 'elementwise' in k or 'reduce' in k or 'SubTensor' in k: g = 'torch element-wise / reductions / copies'
            else: g = 'other'
            groups[g] += t
        fo.write('# by owner (ms per step):' + '; '.join(' %s %.2f' % (g, t / 1e6 / a.steps) for g, t in groups.items()) + '\n')
    print(open(out).read()[:6000])


if __name__ == '__main__':
    main()
