"""Every launch of the LAST iteration of a rocprofv3 --kernel-trace run, in order: python tools/experiments/launch_list.py <trace dir> <launches per iteration> [name filter]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ''
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))[-n:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    k = re.sub(r'void |\(anonymous namespace\)::|at::native::', '', r['Kernel_Name'])
    k = re.sub(r'\(.*', '', k)[:70]
    if flt and flt not in k:
        continue
    print('%9.1f us  +%7.1f us  grid %8s wg %4s lds %6s  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                                           r.get('Grid_Size', '?'), r.get('Workgroup_Size', '?'), r.get('LDS_Block_Size', '?'), k))
