#!/bin/bash
# Same-box A/B of two builds of libesr_hip.so (the shipping one and e.g. a saved copy of an earlier round's), runs alternating:
#   tools/experiments/ab_libs.sh <other.so> [repeats]      prints ms per step of configs[1] (split, mixed), configs[2] and configs[4]
OTHER=$(readlink -f "${1:?other library}"); REP=${2:-2}
cd "$(dirname "$0")/../.."
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline())
alt=d.get('alt_precision')
print('$1 $2: %.2f ms' % d['ms_per_step'] + ('   mixed %.2f ms' % alt['ms_per_step'] if alt else ''))"; }
for i in $(seq $REP); do
  for w in c2 c3 c5; do
    extra=""; [ $w = c2 ] && extra="--no-extra-workloads --no-cpu-baseline --steps 20"
    python bench.py --workload $w --warmup 3 $extra | line this $w
    ESR_HIP_LIBRARY=$OTHER python bench.py --workload $w --warmup 3 $extra | line other $w
  done
done
