#!/bin/bash
# Extra SQ counter passes over the configs[1] forward (one rocprofv3 --pmc pass per group, never combined with a trace domain):
# usage: tools/pmc_extra.sh <tag> [precision]   ->  gpurun_out/<tag>/<group>/..., summary printed per conv instantiation
set -e
TAG=${1:?tag}; PREC=${2:-split}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
B="python bench.py --steps 2 --warmup 1 --no-alt-precision --no-cpu-baseline --no-extra-workloads --precision $PREC"
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d "$OUT/g$i" -- $B > "$OUT/g$i.log" 2>&1 || echo "group $i ($G) failed: $(tail -2 $OUT/g$i.log)"
done
python - "$OUT" <<'PY'
import collections, csv, glob, re, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/g*/**/*_counter_collection.csv', recursive=True):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        m = re.search(r'conv3x3_tile_kernel<[^>]*>', r['Kernel_Name'])
        if m:
            acc[(m.group(0), r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
    for (k, _, c), v in acc.items():
        per[k][c].append(v)
for k in sorted(per):
    n = max(len(v) for v in per[k].values())
    if n < 20:
        continue
    print(k, n, 'launches')
    for c in sorted(per[k]):
        v = per[k][c]
        print('   %-32s %14.0f' % (c, sum(v) / len(v)))
PY
