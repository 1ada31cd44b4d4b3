#!/usr/bin/env python
"""Condense rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

    python tools/summarise_profiles.py --tag r01_v4 --stats gpurun_out/prof_v4/runc --fetch gpurun_out/pmc_fetch4/runc \
        --write gpurun_out/pmc_write4/runc [--sq gpurun_out/pmc_sq4/runc]

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3's --stats table), profiles/<tag>_conv_launches.csv (one forward's conv
launches in order with their durations) and profiles/<tag>_pmc_traffic.json, which bench.py reads for `roofline.traffic`:
HBM bytes per conv launch = (2 x FETCH_SIZE + WRITE_SIZE) / launches, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
wide coalesced reads on gfx950 (it tallies 128-byte requests at 64 bytes); both counters are reported in KiB by rocprofv3.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_CONV = 351


def one(pattern):
    f = glob.glob(pattern)
    assert f, pattern
    return f[0]


def counter_sum(d, kernel_substr='conv3x3'):
    rows = [r for r in csv.DictReader(open(one(os.path.join(d, '*_counter_collection.csv')))) if kernel_substr in r['Kernel_Name']]
    per = collections.defaultdict(float)
    for r in rows:
        per[r['Counter_Name']] += float(r['Counter_Value'])
    ndisp = len({r['Dispatch_Id'] for r in rows})
    return per, ndisp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tag', required=True)
    ap.add_argument('--stats')
    ap.add_argument('--fetch')
    ap.add_argument('--write')
    ap.add_argument('--sq')
    a = ap.parse_args()
    out = os.path.join(ROOT, 'profiles')
    if a.stats:
        shutil.copy(one(os.path.join(a.stats, '*_kernel_stats.csv')), os.path.join(out, a.tag + '_kernel_stats.csv'))
        rows = sorted(csv.DictReader(open(one(os.path.join(a.stats, '*_kernel_trace.csv')))), key=lambda r: int(r['Start_Timestamp']))
        conv = [r for r in rows if 'conv3x3' in r['Kernel_Name']][-N_CONV:]
        with open(os.path.join(out, a.tag + '_conv_launches.csv'), 'w') as f:
            f.write('index,kernel,workgroups,lds_bytes,vgpr,sgpr,start_us,duration_us\n')
            t0 = int(conv[0]['Start_Timestamp'])
            for i, r in enumerate(conv):
                k = re.search(r'conv3x3_tile_kernel<[^>]*>', r['Kernel_Name']).group(0)
                f.write('%d,"%s",%d,%s,%s,%s,%.1f,%.1f\n' % (i, k, int(r['Grid_Size_X']) // 256, r['LDS_Block_Size'], r['VGPR_Count'], r['SGPR_Count'],
                                                         (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
        tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in conv)
        print('one forward: %d conv launches, sum of durations %.2f ms, average %.1f us' % (len(conv), tot / 1e6, tot / 1e3 / len(conv)))
    if a.fetch and a.write:
        fe, nf = counter_sum(a.fetch)
        wr, nw = counter_sum(a.write)
        fetch_b = fe['FETCH_SIZE'] * 1024 / nf
        write_b = wr['WRITE_SIZE'] * 1024 / nw
        d = {'kernel': 'conv3x3_tile_kernel (all instantiations)', 'launches_measured': nf,
             'FETCH_SIZE_bytes_per_launch_raw': fetch_b, 'WRITE_SIZE_bytes_per_launch': write_b,
             'fetch_correction': 'x2 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md, HBM section)',
             'hbm_bytes_per_launch': 2 * fetch_b + write_b, 'hbm_bytes_per_forward': (2 * fetch_b + write_b) * N_CONV}
        # the CEM kernels of the same forwards (downscale / lrfilter / upscale, 2-D or separable): HBM bytes per forward against the fused ideal
        # of SURVEY.md 8(d) (read g + read x + write out = 243.7 MB for configs[1])
        fc, nfc = counter_sum(a.fetch, 'cem_')
        wc, nwc = counter_sum(a.write, 'cem_')
        if nfc and nwc:
            nfwd = max(nf // N_CONV, 1)
            d['cem_kernels'] = {'launches_measured': nfc, 'FETCH_SIZE_bytes_per_forward_raw': fc['FETCH_SIZE'] * 1024 / nfwd,
                                'WRITE_SIZE_bytes_per_forward': wc['WRITE_SIZE'] * 1024 / nfwd,
                                'hbm_bytes_per_forward': (2 * fc['FETCH_SIZE'] + wc['WRITE_SIZE']) * 1024 / nfwd, 'fused_ideal_bytes_per_forward': 243.7e6}
        if a.sq:
            sq, ns = counter_sum(a.sq)
            d['sq_counters_sum_over_%d_launches' % ns] = dict(sq)
        import sys
        sys.path.insert(0, ROOT)
        import bench
        d['kernel_set'] = bench.kernel_set_id()      # the sources these counters describe (bench.py prints them only while it matches)
        json.dump(d, open(os.path.join(out, a.tag + '_pmc_traffic.json'), 'w'), indent=1)
        print(json.dumps(d, indent=1))


if __name__ == '__main__':
    main()
