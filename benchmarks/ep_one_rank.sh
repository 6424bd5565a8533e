#!/bin/bash
# Expert parallelism with ONE rank (dispatch / return exchanges really issued over RCCL, nothing on the wire)
# against the local grouped-expert path, on one box: the cost of the dispatcher itself.
# usage: bash benchmarks/ep_one_rank.sh [micro-batch]
set -e
B=${1:-64}
export MASTER_ADDR=127.0.0.1
echo "== local experts (single engine), micro-batch $B"
timeout -k 10 200 python bench.py --model gpt-moe-8x --batch $B --steps 15 --warmup 4 2>/dev/null | cut -c1-330
port=29740
for extra in "--mode ep" "--mode ep --ep_capacity 1.0" "--mode ep_fsdp"; do
  port=$((port + 1))
  echo "== $extra"
  DPA_FORCE_DIST=1 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port $port bench.py --gpus 1 --model gpt-moe-8x --batch $B --steps 15 --warmup 4 $extra 2>/dev/null | cut -c1-330
done
