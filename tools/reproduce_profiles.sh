#!/bin/bash
# Regenerates the measurements committed under profiles/ for the current kernel set, on a machine with one MI355X:
#   the bench line, the rocprofv3 kernel summary of the same command, and the HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in
#   SEPARATE passes, never combined with a trace domain), then condenses them with tools/summarise_profiles.py.
# usage: tools/reproduce_profiles.sh <tag> [precision]       e.g.  tools/reproduce_profiles.sh r02_v1_split split
#        tools/reproduce_profiles.sh <tag> c3                 kernel summary of the configs[2] G+D step (bench.py --workload c3)
set -e
TAG=${1:?tag}; PREC=${2:-split}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
if [ "$PREC" = c3 ]; then
  python bench.py --workload c3 --steps 2 --warmup 2 > /dev/null 2>&1       # lets MIOpen's find mode fill its cache before the profiled run
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/stats" -- python bench.py --workload c3 --steps 6 --warmup 4 > "$OUT/stats.log" 2>&1
  python tools/summarise_step_trace.py --trace "$OUT/stats" --tag "$TAG" --steps 4
  python bench.py --workload c3 --steps 10 --warmup 3 | tee "profiles/${TAG}_bench.json.log"
  cp profiles/${TAG}_* gpurun_out/        # (gpurun merges only gpurun_out/ back: copy the summaries there too)
  exit 0
fi
B="python bench.py --steps 3 --warmup 1 --no-alt-precision --no-cpu-baseline --no-extra-workloads --precision $PREC"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $B > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $B > /dev/null 2>&1
python tools/summarise_profiles.py --tag "$TAG" --stats "$(dirname "$(find "$OUT/stats" -name '*_kernel_stats.csv' | head -1)")" \
    --fetch "$(dirname "$(find "$OUT/fetch" -name '*_counter_collection.csv' | head -1)")" \
    --write "$(dirname "$(find "$OUT/write" -name '*_counter_collection.csv' | head -1)")"
python bench.py --steps 10 --warmup 3 --no-extra-workloads --precision "$PREC" | tee "profiles/${TAG}_bench.json.log"
cp profiles/${TAG}_* gpurun_out/
