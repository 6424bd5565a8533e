#!/bin/bash
# DDP launch example: one process per MI355X over RCCL (same flags as the reference's
# multi-gpu/ddp/train.sh). For ZeRO-1 / ZeRO-2 / FSDP launch multi-gpu/zero1,2/train_zero1.py,
# multi-gpu/zero1,2/train_zero2.py or multi-gpu/fsdp/train.py with the same arguments.

NGPU=${NGPU:-8}
cd "$(dirname "$0")"
export HSA_ENABLE_IPC_MODE_LEGACY=0
# --max-restarts with "--resume auto": after a failure torchrun relaunches the same command line and every
# rank continues from its last periodic resume file (--save_interval), or from scratch if there is none
torchrun --standalone --local-addr 127.0.0.1 --max-restarts 3 --nproc_per_node=$NGPU train.py \
  --preset gpt2-124m --dataset synthetic --synthetic_tokens 16777216 \
  --batch_size 32 --total_batch_size_str "32*1024*$NGPU" --max_iters 2000 \
  --learning_rate 6e-4 --warmup_steps 100 --grad_clip 1.0 --bucket_mb 48 \
  --eval --eval_interval 200 --eval_iters 20 --save_model --file_name gpt2_124m_ddp --save_interval 200 --resume auto "$@"
