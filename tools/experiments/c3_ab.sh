#!/bin/bash
# same-box A/B of the configs[2] step under an environment switch: tools/experiments/c3_ab.sh VAR valueA valueB [repeats]
VAR=$1; A=$2; B=$3; N=${4:-2}
cd "$(dirname "$0")/../.."
for r in $(seq $N); do
  for v in $A $B; do
    env $VAR=$v python bench.py --workload c3 --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/c3_ab.json
    python - "$VAR=$v" <<'PY'
import json, sys
l = json.loads(open('/tmp/c3_ab.json').read())
print(sys.argv[1], round(l['ms_per_step'], 2), {k: round(v, 2) for k, v in l['phases_ms'].items()})
PY
  done
done
