#!/bin/bash
# Round 4, VERDICT r3 item 1: same-box A/B of the configs[1] forward (split) on the shipping library and the two experiment builds of the conv
# kernel (make -C explorable-super-resolution_amd/csrc experiments): s_setprio around the MFMA phase, buffer_load ... lds copies; then the
# counter passes of tools/pmc_mfma.sh for each, and the instruction-mix microbenchmark of the ping-pong arrangement with its own counter passes.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04_item1
mkdir -p $O
L=$PWD/explorable-super-resolution_amd/esr_hip
B="python bench.py --steps 20 --warmup 5 --no-alt-precision --no-cpu-baseline"
for rep in 1 2 3; do
  for v in base prio buflds; do
    if [ $v = base ]; then unset ESR_HIP_LIBRARY; else export ESR_HIP_LIBRARY=$L/libesr_hip_exp_$v.so; fi
    $B 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v rep$rep', round(d['ms_per_step'],2), 'ms  frac', round(d['roofline']['frac'],4))" | tee -a $O/ab.txt
  done
done
for v in base prio buflds; do
  if [ $v = base ]; then unset ESR_HIP_LIBRARY; else export ESR_HIP_LIBRARY=$L/libesr_hip_exp_$v.so; fi
  bash tools/pmc_mfma.sh r04_v0_split_$v split > $O/pmc_$v.log 2>&1
done
unset ESR_HIP_LIBRARY
cd profiles/microbench
bin/pingpong_mix | tee ../../$O/pingpong_mix.log
rocprofv3 --kernel-trace --output-format csv -d ../../$O/mb_trace -- bin/pingpong_mix > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d ../../$O/mb_sq -- bin/pingpong_mix > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d ../../$O/mb_grbm -- bin/pingpong_mix > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d ../../$O/mb_wait -- bin/pingpong_mix > /dev/null 2>&1
cd ../..
ls -R $O | head -50
cp profiles/r04_v0_* $O/ 2>/dev/null
