"""Per-kernel GPU time of a rocprofv3 --kernel-trace run, per iteration: python tools/experiments/kernel_table.py <trace dir> <iterations>"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
n = float(sys.argv[2])
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    k = re.sub(r'void |\(anonymous namespace\)::|at::native::', '', r['Kernel_Name'])
    k = re.sub(r'\(.*', '', k)[:80]
    tot[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
    cnt[k] += 1
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) * 1e-6
print('%d launches, %.2f ms of kernels per iteration' % (len(rows) / n, sum(tot.values()) / n))
for k in sorted(tot, key=tot.get, reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print('%8.3f ms  %6.1f launches  %7.1f us  %s' % (tot[k] / n, cnt[k] / n, tot[k] / cnt[k] * 1e3, k))
