#!/bin/bash
# Which flag makes the allocator's peak grow with the step count (300 steps, peak at 100 against 300).
D=$(mktemp -d)
export MASTER_ADDR=127.0.0.1
common="--dataset synthetic --synthetic_tokens 2097152 --vocab_size 1024 --max_iters 300 --log_interval 100 --data_dir $D/data
 --n_embd 256 --n_layer 2 --n_head 4 --up_dim 512 --batch_size 8 --block_size 256 --total_batch_size_str 8*256 --hip_graph off"
n=0
show() { a=$(echo "$2" | grep "^step: 100 " | sed -n 's/.*GPU RAM: \([0-9.]*\)GB.*/\1/p'); b=$(echo "$2" | grep "^step: 300 " | sed -n 's/.*GPU RAM: \([0-9.]*\)GB.*/\1/p'); echo "$1 | peak $a -> $b GB"; }
single() { n=$((n + 1)); out=$(timeout -k 10 300 python single-gpu/train.py $common --file_name $D/r$n "$@" 2>&1); show "single $*" "$out"; }
ddp1() { n=$((n + 1)); out=$(DPA_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29950 + n)) multi-gpu/ddp/train.py $common --file_name $D/r$n "$@" 2>&1); show "ddp(1) $*" "$out"; }
gqa="--attn gqa --n_kv_heads 2 --pos_emb rope --non_linearity swiglu"
single $gqa
single $gqa --dropout 0.1
single $gqa --act_recomp
single $gqa --act_recomp --act_recomp_scope block
single $gqa --dropout 0.1 --act_recomp --act_recomp_scope block
ddp1 $gqa
ddp1 $gqa --vocab_parallel
ddp1 $gqa --tied_tail split
ddp1 $gqa --grad_dtype fp32
rm -rf $D
