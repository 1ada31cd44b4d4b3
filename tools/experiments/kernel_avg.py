"""Average duration of the launches whose name contains a substring, from rocprofv3 --kernel-trace --stats output dirs:
python tools/experiments/kernel_avg.py <substring> <dir> [<dir> ...]"""
import csv, glob, sys
pat = sys.argv[1]
for d in sys.argv[2:]:
    f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if pat in r['Name']:
            print(d, r['Name'][:90].replace('(anonymous namespace)::', ''), 'calls', r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3))
