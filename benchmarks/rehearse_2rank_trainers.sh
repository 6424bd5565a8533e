#!/bin/bash
# The multi-GPU training entry points end to end with two ranks sharing ONE GPU (host-staged gloo):
# engine build, training steps with gradient accumulation, evaluation, checkpoint save, shutdown.
set -e
export DPA_BACKEND=gloo MASTER_ADDR=127.0.0.1 DPA_HANG_DUMP_S=100
out=${1:-/tmp/dpa_rehearse}
mkdir -p "$out"
port=29650
common="--preset char-10m --dataset synthetic --synthetic_tokens 262144 --batch_size 4 --total_batch_size_str 4*256*2*2
        --max_iters 6 --eval --eval_interval 3 --eval_iters 2 --warmup_steps 2 --save_model"
for entry in multi-gpu/ddp/train.py "multi-gpu/zero1,2/train_zero1.py" "multi-gpu/zero1,2/train_zero2.py" multi-gpu/fsdp/train.py; do
  port=$((port + 1))
  name=$(basename "$entry" .py)
  echo "== $entry"
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port $port "$entry" $common --file_name "$out/$name" --data_dir "$out/data"
  ls -la "$out/${name}_ckpt.pt"
done
