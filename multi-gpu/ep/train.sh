#!/bin/bash
# Expert-parallel MoE launch example: one process per MI355X over RCCL; 8 routed experts on 8 GPUs
# = one expert per GPU, tokens dispatched with two all-to-alls per MoE layer over xGMI. The dense
# layers are fully sharded (train_ep_fsdp.py) or data-parallel with bucketed all-reduce (train_ep.py).

NGPU=${NGPU:-8}
cd "$(dirname "$0")"
export HSA_ENABLE_IPC_MODE_LEGACY=0
# --max-restarts with "--resume auto": after a failure torchrun relaunches the same command line and every
# rank continues from its last periodic resume file (--save_interval), or from scratch if there is none
torchrun --standalone --local-addr 127.0.0.1 --max-restarts 3 --nproc_per_node=$NGPU train_ep_fsdp.py \
  --preset gpt-moe-8x --dataset synthetic --synthetic_tokens 16777216 \
  --batch_size 16 --total_batch_size_str "16*1024*$NGPU" --max_iters 2000 \
  --learning_rate 6e-4 --warmup_steps 100 --grad_clip 1.0 \
  --eval --eval_interval 200 --eval_iters 20 --save_model --file_name gpt_moe8x_ep --save_interval 200 --resume auto "$@"
