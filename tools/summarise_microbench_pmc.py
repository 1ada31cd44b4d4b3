#!/usr/bin/env python
"""Condense the rocprofv3 passes of profiles/microbench/pingpong_mix (tools/experiments/r04_item1_ab.sh: kernel trace, SQ_VALU_MFMA_BUSY_CYCLES,
GRBM_GUI_ACTIVE, SQ wait counters — one pass each) into profiles/<tag>_pmc.json: per variant (dispatch order of the binary, last of its three
repetitions) wall time, MFMA pipe busy fraction, effective clock, bf16 issue rate and the wave-state split."""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 1024
NAMES = ['2wg', '2wg, setprio while multiplying', '2wg, buffer_load lds', 'pingpong, shared weights', 'pingpong, shared, prio multiply',
         'pingpong, shared, prio waves 4-7', 'pingpong, shared, buffer_load lds', 'pingpong, weights per half']


def one(d, pat):
    f = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return f[0] if f else None


def dispatches(d):
    """dispatch id -> {counter: value summed over the chip}, in dispatch order, benchmark kernels only"""
    per = collections.OrderedDict()
    f = one(d, '*_counter_collection.csv')
    for r in csv.DictReader(open(f)):
        if not r['Kernel_Name'].startswith('void k_'):
            continue
        per.setdefault(int(r['Dispatch_Id']), collections.defaultdict(float))[r['Counter_Name']] += float(r['Counter_Value'])
    return [per[k] for k in sorted(per)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dir', required=True)
    ap.add_argument('--tag', required=True)
    ap.add_argument('--iters', type=int, default=4000)
    a = ap.parse_args()
    walls = []
    for r in csv.DictReader(open(one(os.path.join(a.dir, 'mb_trace'), '*_kernel_trace.csv'))):
        if r['Kernel_Name'].startswith('void k_'):
            walls.append((int(r['Dispatch_Id']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9))
    walls = [w for _, w in sorted(walls)]
    sq, grbm, wait = (dispatches(os.path.join(a.dir, d)) for d in ('mb_sq', 'mb_grbm', 'mb_wait'))
    n = len(NAMES)
    out = {'binary': 'profiles/microbench/pingpong_mix (3 repetitions per variant, the last one reported)', 'iterations_per_launch': a.iters, 'sources': {}}
    for s, src in enumerate(('48 MB (Infinity Cache)', '2 GB (HBM stream)')):
        rows = {}
        for v, name in enumerate(NAMES):
            i = (s * n + v) * 3 + 2
            t = walls[i]
            gui = grbm[i]['GRBM_GUI_ACTIVE']
            per_xcd = gui / 8 if gui / t > 3.0e9 else gui
            busy = sq[i]['SQ_VALU_MFMA_BUSY_CYCLES']
            w = wait[i]
            wc = w.get('SQ_WAVE_CYCLES') or 1
            rows[name] = {'wall_us_per_iteration': t / a.iters * 1e6, 'effective_clock_ghz': per_xcd / t / 1e9, 'mfma_busy_frac': busy / N_SIMD / per_xcd,
                          'mfma_issue_pflops_bf16': busy / 32 * 32768 / t / 1e15,
                          'wave_state_frac': {k: w.get(k, 0) / wc for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY')}}
        out['sources'][src] = rows
    json.dump(out, open(os.path.join(ROOT, 'profiles', a.tag + '_pmc.json'), 'w'), indent=1)
    for src, rows in out['sources'].items():
        print(src)
        for k, r in rows.items():
            print('  %-34s %.3f us  %.2f GHz  busy %.3f  %.3f PF  wait %.2f' % (k, r['wall_us_per_iteration'], r['effective_clock_ghz'], r['mfma_busy_frac'],
                                                                              r['mfma_issue_pflops_bf16'], r['wave_state_frac']['SQ_WAIT_ANY']))


if __name__ == '__main__':
    main()
