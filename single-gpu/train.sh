#!/bin/bash
# Single-GPU launch example (same flags as the reference's single-gpu/train.sh).
# GPT-2-124M shape, T=1024, micro-batch 32, 32*1024*2 tokens per optimizer step.
# Use --dataset synthetic --synthetic_tokens 16777216 when no tokenised corpus is present.

DATASET='synthetic'
TOTAL_BATCH_SIZE_STR="32*1024*2"
BATCH_SIZE=32
MAX_ITERS=2000
LEARNING_RATE=6e-4
WARMUP_STEPS=100
GRAD_CLIP=1.0
EVAL_INTERVAL=200
EVAL_ITERS=20

VOCAB_SIZE=50304
BLOCK_SIZE=1024
N_EMBD=768
N_LAYER=12
N_HEAD=12
UP_DIM=3072
POS_EMB='learn'
ATTN='mha'
NON_LINEARITY='gelu'
DROPOUT=0.0

cd "$(dirname "$0")"
python train.py \
  --dataset $DATASET --synthetic_tokens 16777216 \
  --total_batch_size_str $TOTAL_BATCH_SIZE_STR --batch_size $BATCH_SIZE --max_iters $MAX_ITERS \
  --learning_rate $LEARNING_RATE --warmup_steps $WARMUP_STEPS --grad_clip $GRAD_CLIP \
  --eval --eval_interval $EVAL_INTERVAL --eval_iters $EVAL_ITERS \
  --vocab_size $VOCAB_SIZE --block_size $BLOCK_SIZE --n_embd $N_EMBD --n_layer $N_LAYER --n_head $N_HEAD \
  --up_dim $UP_DIM --pos_emb $POS_EMB --attn $ATTN --non_linearity $NON_LINEARITY --dropout $DROPOUT \
  --save_model --file_name gpt2_124m "$@"
