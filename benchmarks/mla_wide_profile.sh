#!/bin/bash
# Kernel mix of a training run with the reference's launch configuration of latent attention
# (single-gpu/train.sh:34-36 of the reference: --attn mla --pos_emb rope, latent 256, rotary head 128, 1024 / 8 heads).
out=${1:-gpurun_out/mla_wide}
mkdir -p $out
cd "$(dirname "$0")/.."
timeout -k 10 300 python single-gpu/train.py --dataset synthetic --synthetic_tokens 4000000 --data_dir /tmp/dpa_mla_wide \
  --vocab_size 50304 --block_size 1024 --n_embd 1024 --n_head 8 --n_layer 8 --up_dim 4096 --attn mla --pos_emb rope \
  --q_latent_dim 256 --kv_latent_dim 256 --rope_head_dim 128 --non_linearity swiglu \
  --batch_size 16 --total_batch_size_str 16*1024 --max_iters 12 --warmup_steps 2 --log_interval 4 \
  --profile kernels --hip_graph off --file_name /tmp/dpa_mla_wide_run > $out/train_profile_kernels.log 2>&1 || { tail -20 $out/train_profile_kernels.log; exit 1; }
grep -A40 "total GPU kernel time" $out/train_profile_kernels.log | cut -c1-160 > $out/kernel_table.txt
head -45 $out/kernel_table.txt
echo "stock attention / batched-matmul kernels in the table:"
grep -c -i "scaled_dot\|repeat_interleave\|bmm\|attention_forward\|fmha" $out/kernel_table.txt || true
grep "iter\|tok/s" $out/train_profile_kernels.log | tail -3
