#!/bin/bash
# configs[2] step at ONE rank: plain process (one weight-gradient launch, no collective) vs one rank under a launcher with the gradient exchange
# after the backward (default) vs from inside it, bucket by bucket (--early-exchange) — what the bucketed launches cost where nothing can overlap.
cd "$(dirname "$0")/../.."
line() { grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1: %.2f ms  [%s]  G backward %.2f ms' % (d['ms_per_step'], d['gradient_exchange'], d['phases_ms']['G_losses_and_backward']))"; }
for i in 1 2 3; do
  python bench.py --workload c3 --steps 10 --warmup 3 | line "plain process"
  for f in "" "--early-exchange"; do
    HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29520 + i)) bench.py --gpus 1 --workload c3 --steps 10 --warmup 3 $f 2>/dev/null | line "one rank under a launcher $f"
  done
done
