#!/bin/bash
# The user-facing flow on one GPU: train a small model on synthetic tokens with --save_model, then sample from the
# checkpoint with single-gpu/sample.py (bf16, graphed decode).
set -e
D=$(mktemp -d)
timeout -k 10 300 python single-gpu/train.py --dataset synthetic --synthetic_tokens 2097152 --vocab_size 257 --block_size 256 \
  --n_embd 384 --n_layer 6 --n_head 6 --up_dim 1536 --attn mha --pos_emb rope --non_linearity swiglu --batch_size 16 \
  --total_batch_size_str "16*256" --max_iters 60 --data_dir $D/data --file_name $D/run --save_model 2>&1 | tail -4
timeout -k 10 200 python single-gpu/sample.py --ckpt $D/run_ckpt.pt --prompt "To be, or not" --max_new_tokens 120 --num_samples 4 \
  --tokenizer bytes --token_ids 2>&1 | grep -E "^\[sample\]|^--- sample 3" 
rm -rf $D
