#!/bin/bash
# EIGHT ranks sharing ONE GPU over gloo: the shard shapes, bucket layouts, vocabulary slices and the
# one-expert-per-rank layout of an 8-GPU node, run through the GPU kernels on a single-GPU box. RCCL refuses
# several ranks on one device, so transport is host-staged gloo; only completion, the loss and the byte counts
# matter. usage: bash benchmarks/rehearse_8rank_gloo.sh > gpurun_out/rehearse8.txt
set -e
export DPA_BACKEND=gloo MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=2
port=29710
run() {   # label, then bench.py arguments
  local label=$1; shift
  port=$((port + 1))
  echo "== $label"
  DPA_HANG_DUMP_S=150 timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
    --master-port $port bench.py --gpus 8 --steps 2 --warmup 1 --batch 2 "$@" 2>gpurun_out/rehearse8_err.log | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)
        print('ranks', d['n_gpus'], 'transport', d['transport'], 'ms/step %.0f' % d['ms_per_step'], 'final_loss %.4f' % d['final_loss'],
              'buckets', d['n_buckets'], 'bytes', d['comm_bytes_per_step'])
"
}
mkdir -p gpurun_out
run "ddp gpt2-124m" --mode ddp
run "ddp gpt2-124m vocabulary-parallel head (uneven 128-aligned slices at 8 ranks)" --mode ddp --vocab_parallel
run "zero2 gpt2-124m" --mode zero2
run "fsdp gpt2-124m" --mode fsdp
run "fsdp gqa-rope-swiglu-124m" --mode fsdp --model gqa-rope-swiglu-124m
run "ep gpt-moe-8x (one expert per rank)" --mode ep --model gpt-moe-8x
run "ep gpt-moe-8x static capacity 2.0" --mode ep --model gpt-moe-8x --ep_capacity 2.0
run "ep_fsdp gpt-moe-8x" --mode ep_fsdp --model gpt-moe-8x
