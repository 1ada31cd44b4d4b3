#!/usr/bin/env python
"""Condense the passes of tools/pmc_workload.sh into profiles/<tag>_pmc.json: per kernel (name matched by --keep, template arguments kept),
over the second half of its launches (steady state): launches, mean wall duration (kernel-trace pass) and the mean of every counter per launch,
plus the derived figures the DESIGN quotes:
  mfma_busy_frac      SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD)
  effective_clock_ghz GRBM_GUI_ACTIVE per XCD / wall duration   (rocprofv3 reports the sum over the 8 XCDs)
  hbm_read_gb / hbm_write_gb   FETCH_SIZE x 1 KiB x 2 (gfx950: the counter reads half of a wide coalesced stream, MI355X_MICROARCH.md HBM section) and
                      WRITE_SIZE x 1 KiB — A/B ratios are exact, absolutes carry that correction
  l2_hit_rate         TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  wait_frac           SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
Counter passes are separate runs of the same command: per-launch means are comparable, individual launches are not paired."""
import argparse
import collections
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """kernel name without its namespace and argument list (template arguments kept)"""
    n = name.replace('(anonymous namespace)::', '').replace('void ', '')
    depth = 0
    for i, ch in enumerate(n):
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            return n[:i].strip()
    return n.strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tag', required=True)
    ap.add_argument('--dir', required=True)
    ap.add_argument('--keep', default='.')
    ap.add_argument('--command', default='')
    a = ap.parse_args()
    keep = re.compile(a.keep)
    wall, grid = collections.defaultdict(list), {}
    for f in glob.glob(os.path.join(a.dir, 'trace', '**', '*_kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if keep.search(k):
                wall[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9)
                grid[k] = int(r['Grid_Size_X']) // max(int(r.get('Workgroup_Size_X', 256) or 256), 1)
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(a.dir, 'g*', '**', '*_counter_collection.csv'), recursive=True):
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if keep.search(k):
                acc[(k, r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
        for (k, did, c), v in sorted(acc.items(), key=lambda kv: int(kv[0][1])):
            cnt[k][c].append(v)
    steady = lambda v: v[len(v) // 2:] if len(v) > 1 else v
    mean = lambda v: sum(steady(v)) / len(steady(v)) if v else None
    out = {'command': a.command, 'kernels': {}}
    for k in sorted(wall, key=lambda k: -sum(wall[k])):
        t = mean(wall[k])
        e = {'launches': len(wall[k]), 'workgroups': grid.get(k), 'wall_us': t * 1e6, 'counters': {c: mean(v) for c, v in sorted(cnt[k].items())}}
        c = e['counters']
        gui = c.get('GRBM_GUI_ACTIVE')
        if gui:
            per_xcd = gui / 8 if gui / t > 3.0e9 else gui
            e['effective_clock_ghz'] = per_xcd / t / 1e9
            if c.get('SQ_VALU_MFMA_BUSY_CYCLES') is not None:
                e['mfma_busy_frac'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * per_xcd)
                e['mfma_32x32x16_issued'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / 32
        if c.get('FETCH_SIZE') is not None:
            e['hbm_read_gb'] = c['FETCH_SIZE'] * 1024 * 2 / 1e9
        if c.get('WRITE_SIZE') is not None:
            e['hbm_write_gb'] = c['WRITE_SIZE'] * 1024 / 1e9
        if c.get('TCC_HIT_sum') is not None and c.get('TCC_MISS_sum') is not None and c['TCC_HIT_sum'] + c['TCC_MISS_sum'] > 0:
            e['l2_hit_rate'] = c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum'])
        if c.get('SQ_WAVE_CYCLES'):
            e['wait_frac'] = (c.get('SQ_WAIT_INST_ANY') or 0.0) / c['SQ_WAVE_CYCLES']
        if c.get('SQ_INSTS_MFMA'):
            for n in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM'):
                if c.get(n) is not None:
                    e[n.lower().replace('sq_insts_', '') + '_per_mfma'] = c[n] / c['SQ_INSTS_MFMA']
        out['kernels'][k] = e
    p = os.path.join(ROOT, 'profiles', a.tag + '_pmc.json')
    json.dump(out, open(p, 'w'), indent=1)
    for k, e in list(out['kernels'].items())[:12]:
        print('%-90s %5d x %8.1f us  busy %s  clk %s  L2 hit %s  wait %s' % (k[:90], e['launches'], e['wall_us'],
              '%.3f' % e['mfma_busy_frac'] if 'mfma_busy_frac' in e else '-', '%.2f' % e['effective_clock_ghz'] if 'effective_clock_ghz' in e else '-',
              '%.2f' % e['l2_hit_rate'] if 'l2_hit_rate' in e else '-', '%.2f' % e['wait_frac'] if 'wait_frac' in e else '-'))
    print('wrote', p)


if __name__ == '__main__':
    main()
