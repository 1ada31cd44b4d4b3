#!/bin/bash
# MFMA utilisation and effective shader clock of the configs[1] forward, per conv instantiation:
#   pass 1  --kernel-trace            wall duration of every launch
#   pass 2  --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES        (SQ block)
#   pass 3  --pmc GRBM_GUI_ACTIVE GRBM_COUNT                     (effective clock = GRBM_GUI_ACTIVE / wall)
#   pass 4  --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA     (cross-check of the MFMA count, when the counters exist)
# Counter passes are separate runs and never combined with a trace domain.  usage: tools/pmc_mfma.sh <tag> [precision]
set -e
TAG=${1:?tag}; PREC=${2:-split}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
B="python bench.py --steps 3 --warmup 1 --no-alt-precision --no-cpu-baseline --no-extra-workloads --precision $PREC"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- $B > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq" -- $B > "$OUT/sq.log" 2>&1 || echo "sq pass failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$OUT/grbm" -- $B > "$OUT/grbm.log" 2>&1 || echo "grbm pass failed"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA --output-format csv -d "$OUT/mops" -- $B > "$OUT/mops.log" 2>&1 || echo "mops pass failed"
python tools/summarise_pmc_mfma.py --tag "$TAG" --dir "$OUT" --precision "$PREC"
cp profiles/${TAG}_* gpurun_out/ 2>/dev/null || true
