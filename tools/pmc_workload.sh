#!/bin/bash
# Counters of the kernels of ANY bench.py workload: one rocprofv3 pass per counter group — never combined with a trace
# domain — plus one kernel-trace pass for the wall durations; condensed per kernel into profiles/<tag>_pmc.json by tools/summarise_pmc_workload.py.
# usage: tools/pmc_workload.sh <tag> "<python script and its arguments>" [kernel-name regex to keep]
#   tools/pmc_workload.sh r04_wgrad "bench.py --workload c3 --steps 4 --warmup 3" 'wgrad_batch'
#   tools/pmc_workload.sh r04_c5_f16 "bench.py --workload c5 --precision f16 --steps 2 --warmup 1 --no-cpu-baseline" 'conv3x3_tile'
#   tools/pmc_workload.sh r04_cem_chunk8 "tools/experiments/cem_project_loop.py 8" 'cem_'
set -e
TAG=${1:?tag}; ARGS=${2:?bench arguments}; KEEP=${3:-.}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
B="python $ARGS"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- $B > "$OUT/trace.log" 2>&1
i=0
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_INSTS_VMEM" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d "$OUT/g$i" -- $B > "$OUT/g$i.log" 2>&1 || echo "group $i ($G) failed: $(tail -2 $OUT/g$i.log)"
done
python tools/summarise_pmc_workload.py --tag "$TAG" --dir "$OUT" --keep "$KEEP" --command "$ARGS"
cp profiles/${TAG}_pmc.json gpurun_out/ 2>/dev/null || true
rm -rf "$OUT"/g[0-9]* "$OUT"/trace      # the raw counter tables are hundreds of MB: gpurun merges at most 64 MiB back
