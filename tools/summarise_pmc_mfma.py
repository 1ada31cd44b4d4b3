#!/usr/bin/env python
"""Condense the passes of tools/pmc_mfma.sh into profiles/<tag>_pmc_mfma.json: per conv3x3 instantiation of the configs[1] forward
  * launches, mean wall duration (kernel trace)
  * SQ_VALU_MFMA_BUSY_CYCLES per launch (MI355X_MICROARCH.md: counts cycles, = 32 x number of v_mfma_f32_32x32x16 issued, summed over the
    chip's SIMDs) and the MFMA count it implies, next to the analytic count of the launch
  * GRBM_GUI_ACTIVE per launch and the effective shader clock = GRBM_GUI_ACTIVE / wall duration (rocprofv3 reports the sum over the 8
    XCDs' GRBMs when the value exceeds what one clock domain can count in the launch's duration: both readings are recorded)
  * mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD)
bench.py carries the chip-level figures (static, like roofline.traffic)."""
import argparse
import collections
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 256 * 4


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None


def inst(name):
    m = re.search(r'conv3x3_tile_kernel<[^>]*>', name)
    return m.group(0) if m else None


def counters(d):
    f = one(os.path.join(d, '**', '*_counter_collection.csv'))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return per
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = inst(r['Kernel_Name'])
        if k:
            acc[(k, r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
    for (k, _, c), v in acc.items():
        per[k][c].append(v)
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tag', required=True)
    ap.add_argument('--dir', required=True)
    ap.add_argument('--precision', default='split')
    a = ap.parse_args()
    wall = collections.defaultdict(list)
    grid = {}
    f = one(os.path.join(a.dir, 'trace', '**', '*_kernel_trace.csv'))
    for r in csv.DictReader(open(f)):
        k = inst(r['Kernel_Name'])
        if k:
            wall[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9)
            grid[k] = int(r['Grid_Size_X']) // 256
    sq, grbm, mops = counters(os.path.join(a.dir, 'sq')), counters(os.path.join(a.dir, 'grbm')), counters(os.path.join(a.dir, 'mops'))
    mean = lambda v: sum(v) / len(v) if v else None
    out = {'precision': a.precision, 'command': 'bench.py --steps 3 --warmup 1 --no-alt-precision --no-cpu-baseline --precision ' + a.precision, 'kernels': {}}
    tot = collections.defaultdict(float)
    for k in sorted(wall):
        t = mean(wall[k])
        e = {'launches': len(wall[k]), 'workgroups': grid[k], 'wall_us': t * 1e6}
        busy, sqb, gui = mean(sq[k].get('SQ_VALU_MFMA_BUSY_CYCLES', [])), mean(sq[k].get('SQ_BUSY_CYCLES', [])), mean(grbm[k].get('GRBM_GUI_ACTIVE', []))
        if busy is not None:
            e['SQ_VALU_MFMA_BUSY_CYCLES'] = busy
            e['mfma_32x32x16_issued'] = busy / 32
            e['mfma_busy_cycles_per_simd'] = busy / N_SIMD
        if sqb is not None:
            e['SQ_BUSY_CYCLES'] = sqb
        if gui is not None:
            e['GRBM_GUI_ACTIVE'] = gui
            # one GRBM per XCD: a value of ~8x what a <=2.4 GHz clock can count in the wall time is the sum over the XCDs
            per_xcd = gui / 8 if gui / t > 3.0e9 else gui
            e['GRBM_GUI_ACTIVE_per_xcd'] = per_xcd
            e['effective_clock_ghz'] = per_xcd / t / 1e9
            if busy is not None:
                e['mfma_busy_frac'] = busy / N_SIMD / per_xcd
        for c, v in mops[k].items():
            e[c] = mean(v)
        out['kernels'][k] = e
        n = len(wall[k])
        tot['wall'] += t * n
        for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE_per_xcd'):
            if c in e:
                tot[c] += e[c] * n
    if tot['GRBM_GUI_ACTIVE_per_xcd']:
        out['all_conv_launches'] = {'wall_s_sum': tot['wall'], 'effective_clock_ghz': tot['GRBM_GUI_ACTIVE_per_xcd'] / tot['wall'] / 1e9,
                                    'mfma_busy_frac': tot['SQ_VALU_MFMA_BUSY_CYCLES'] / N_SIMD / tot['GRBM_GUI_ACTIVE_per_xcd'] if tot['SQ_VALU_MFMA_BUSY_CYCLES'] else None,
                                    'mfma_issue_pflops_bf16': tot['SQ_VALU_MFMA_BUSY_CYCLES'] / 32 * 32768 / tot['wall'] / 1e15 if tot['SQ_VALU_MFMA_BUSY_CYCLES'] else None}
    import sys
    sys.path.insert(0, ROOT)
    import bench
    out['kernel_set'] = bench.kernel_set_id()        # the sources these counters describe (bench.py prints them only while it matches)
    out['library'] = os.environ.get('ESR_HIP_LIBRARY') or 'libesr_hip.so'
    path = os.path.join(ROOT, 'profiles', a.tag + '_pmc_mfma.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out.get('all_conv_launches'), indent=1))
    print('wrote', path)


if __name__ == '__main__':
    main()
