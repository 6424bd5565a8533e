#!/bin/bash
# Every engine through RCCL with ONE rank (collectives really issued, nothing on the wire): the
# bookkeeping / launch overhead of each mode relative to the plain single-GPU engine, on one box.
set -e
export DPA_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
port=29710
echo "== single"
env -u DPA_FORCE_DIST timeout -k 10 200 python bench.py --steps 20 --warmup 6 2>/dev/null
for mode in "$@"; do
  port=$((port + 1))
  echo "== $mode"
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port $port bench.py --gpus 1 --steps 20 --warmup 6 --mode $mode 2>/dev/null
done
