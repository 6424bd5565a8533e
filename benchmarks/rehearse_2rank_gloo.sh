#!/bin/bash
# Two ranks sharing ONE GPU over gloo: rehearses the multi-rank GPU code paths of every engine
# (bucket hooks, side streams, shard bookkeeping) on a single-GPU box. RCCL refuses two ranks on
# one device, so transport is gloo; numbers are meaningless, only completion and the loss matter.
set -e
export DPA_BACKEND=gloo MASTER_ADDR=127.0.0.1
port=29610
for mode in ddp zero1 zero2 fsdp fsdp_keep ep ep_fsdp; do
  port=$((port + 1))
  echo "== $mode"
  DPA_HANG_DUMP_S=100 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port $port bench.py --gpus 2 --steps 3 --warmup 2 --mode $mode --batch 8 $([[ $mode == ep* ]] && echo --model gpt-moe-8x)
done
# the rotary / grouped-query / SwiGLU block (projection with the rotary epilogue, fused SwiGLU) under the sharding engines
for mode in ddp zero2 fsdp; do
  port=$((port + 1))
  echo "== $mode gqa-rope-swiglu-124m"
  DPA_HANG_DUMP_S=100 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port $port bench.py --gpus 2 --steps 3 --warmup 2 --mode $mode --batch 8 --model gqa-rope-swiglu-124m
done
