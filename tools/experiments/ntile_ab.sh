#!/bin/bash
# (Runs against the tree with tools/experiments/patches/r06_three_experiments.patch applied = commit bc221d6.)
# Same-box A/B of the tiles-per-workgroup constant of the one-MFMA conv kernels (csrc/esr_conv.hip: NTILE_ONE_MFMA):
#   tools/experiments/ntile_ab.sh <other.so> [repeats]
# prints ms per step of configs[4] (f16) and of configs[1] in 'mixed' for the shipping library and for <other.so> (a build of the same sources
# with NTILE_ONE_MFMA = 1: sed the constant in a scratch copy of esr_conv.hip, link with the other objects), runs alternating.
OTHER=$(readlink -f "${1:?other library}"); REP=${2:-2}
cd "$(dirname "$0")/../.."
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('$1 $2: %.2f ms per step, generator span %.2f ms' % (d['ms_per_step'], r.get('generator_ms_per_step', float('nan'))))"; }
for i in $(seq $REP); do
  python bench.py --workload c5 --warmup 3 --steps 10 | line this c5
  ESR_HIP_LIBRARY=$OTHER python bench.py --workload c5 --warmup 3 --steps 10 | line other c5
  python bench.py --workload c2 --precision mixed --warmup 3 --steps 20 --no-extra-workloads --no-cpu-baseline | line this c2-mixed
  ESR_HIP_LIBRARY=$OTHER python bench.py --workload c2 --precision mixed --warmup 3 --steps 20 --no-extra-workloads --no-cpu-baseline | line other c2-mixed
done
