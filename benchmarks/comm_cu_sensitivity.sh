#!/bin/bash
# How much does the DDP step slow down when a collective's kernels hold some CUs during backward?
# One GPU, one-rank RCCL group; DPA_EMULATE_COMM occupies <CUs> for <microseconds> at every bucket launch.
set -e
export DPA_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
port=29750
for spec in "" "8,500" "16,500" "32,500" "32,1000" "64,500"; do
  port=$((port + 1))
  echo "== emulate '$spec'"
  DPA_EMULATE_COMM=$spec timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --steps 20 --warmup 6 --mode ddp 2>/dev/null
done
