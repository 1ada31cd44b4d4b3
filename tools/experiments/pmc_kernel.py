"""Counters of ONE kernel out of rocprofv3 --pmc runs: python tools/experiments/pmc_kernel.py <name substring> <dir> [<dir> ...]
prints, per counter, the mean per launch over the launches whose name contains the substring (last half of them: steady state)."""
import collections, csv, glob, sys
pat = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                vals[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in vals.items():
            v = v[len(v) // 2:]
            print('%-32s %14.0f   (%d launches)' % (k, sum(v) / len(v), len(v)))
