#!/bin/bash
# Same-box A/B of the shipping library against another build on the one-MFMA workloads: configs[4] (f16) and configs[1] in 'mixed', alternating.
#   tools/experiments/ab_lib_lines.sh <other.so> [repeats]
OTHER=$(readlink -f "${1:?other library}"); REP=${2:-2}
cd "$(dirname "$0")/../.."
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1 $2: %.2f ms per step, generator span %.2f ms' % (d['ms_per_step'], d['roofline'].get('generator_ms_per_step', float('nan'))))"; }
for i in $(seq $REP); do
  python bench.py --workload c5 --warmup 3 --steps 10 2>/dev/null | line this c5
  ESR_HIP_LIBRARY=$OTHER python bench.py --workload c5 --warmup 3 --steps 10 2>/dev/null | line other c5
  python bench.py --workload c2 --precision mixed --warmup 3 --steps 20 --no-extra-workloads --no-cpu-baseline 2>/dev/null | line this c2-mixed
  ESR_HIP_LIBRARY=$OTHER python bench.py --workload c2 --precision mixed --warmup 3 --steps 20 --no-extra-workloads --no-cpu-baseline 2>/dev/null | line other c2-mixed
done
