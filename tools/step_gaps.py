#!/usr/bin/env python
"""Where a training step's GPU sits idle: from a rocprofv3 --kernel-trace CSV of `bench.py --workload c3`, the gaps between consecutive
kernels of the last full step (a step ends with the generator's Adam launch: the second adam_multi_kernel of a pair), by size class, and
the largest ones with the kernels on either side.
    python tools/step_gaps.py <kernel_trace.csv> [top n]"""
import csv
import re
import sys


def main():
    f = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '')) for r in csv.DictReader(open(f)))
    adam = [i for i, r in enumerate(rows) if 'adam_multi' in r[2]]
    ends = adam[1::2]                      # D's Adam, then G's, per step
    if len(ends) < 3:
        sys.exit('fewer than three steps in the trace')
    seg = rows[ends[-3] + 1:ends[-2] + 1]
    t0, t1 = seg[0][0], seg[-1][1]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [(s1 - e0, n0, n1, (e0 - t0) / 1e6) for (s0, e0, n0), (s1, e1, n1) in zip(seg, seg[1:])]
    print('step: %.2f ms from the first kernel to the last, %.2f ms inside kernels, %d launches, %.2f ms of gaps' %
          ((t1 - t0) / 1e6, busy / 1e6, len(seg), sum(g for g, *_ in gaps if g > 0) / 1e6))
    for lo, hi in ((0, 2000), (2000, 5000), (5000, 20000), (20000, 100000), (100000, 10 ** 12)):
        sel = [g for g, *_ in gaps if lo <= g < hi]
        print('  gaps of %6.0f .. %-8s us: %4d, %.2f ms' % (lo / 1e3, '%.0f' % (hi / 1e3) if hi < 10 ** 12 else 'inf', len(sel), sum(sel) / 1e6))
    short = lambda n: re.sub(r'\(.*', '', n)[:60]
    for g, n0, n1, t in sorted(gaps, reverse=True)[:top]:
        print('  %7.1f us at %6.2f ms: %s -> %s' % (g / 1e3, t, short(n0), short(n1)))


if __name__ == '__main__':
    main()
