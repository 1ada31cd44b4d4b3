#!/bin/bash
# configs[2] step with the generator's weight gradients behind the data-gradient chain (0) and under it (RRDBEngine.wgrad_overlap groups), same box, alternating.
# usage: tools/experiments/c3_overlap_ab.sh [reps]
cd "$(dirname "$0")/../.."
for r in $(seq 1 ${1:-2}); do
  for g in 0 3 0 4; do
    python - $g <<'PY' 2>/dev/null | grep '^groups'
import sys, json, io, contextlib
sys.path.insert(0, 'explorable-super-resolution_amd'); sys.path.insert(0, '.')
import esr_hip.engine as E
E.WGRAD_OVERLAP = int(sys.argv[1])
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main(['--workload', 'c3', '--steps', '20', '--warmup', '4'])
d = [json.loads(l) for l in buf.getvalue().splitlines() if l.startswith('{')][-1]
print('groups %d: %.2f ms per step  %s' % (E.WGRAD_OVERLAP, d['ms_per_step'], {k: round(v, 2) for k, v in d['phases_ms'].items()}))
PY
  done
done
