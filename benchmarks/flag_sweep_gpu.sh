#!/bin/bash
# single-gpu/train.py and the one-rank DDP script under the training flags that change the step's structure
# (evaluation passes, recomputation scopes, dropout, gradient accumulation, graph modes, gradient dtype, resume).
D=$(mktemp -d)
export MASTER_ADDR=127.0.0.1
common="--dataset synthetic --synthetic_tokens 1048576 --vocab_size 1024 --max_iters 6 --data_dir $D/data
 --n_embd 256 --n_layer 2 --n_head 4 --up_dim 512 --batch_size 8 --block_size 256"
fail=0
n=0
check() {
  name=$1; rc=$2; out=$3
  last=$(echo "$out" | grep "^step: 6 " | tail -1)
  if [ $rc -ne 0 ] || [ -z "$last" ] || echo "$last" | grep -qi "nan"; then echo "FAIL: $name"; echo "$out" | tail -6; fail=1; else echo "ok: $name | $(echo $last | cut -c1-40)"; fi
}
single() { n=$((n + 1)); out=$(timeout -k 10 240 python single-gpu/train.py $common --file_name $D/r$n "$@" 2>&1); check "single $*" $? "$out"; }
ddp1() { n=$((n + 1)); out=$(DPA_FORCE_DIST=1 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29850 + n)) multi-gpu/ddp/train.py $common --file_name $D/r$n "$@" 2>&1); check "ddp(1) $*" $? "$out"; }
gqa="--attn gqa --n_kv_heads 2 --pos_emb rope --non_linearity swiglu"
moe="--attn mha --pos_emb learn --non_linearity swiglu --moe --n_exp 4 --n_shared 1 --n_act 2"
single $gqa --total_batch_size_str "8*256" --eval --eval_interval 2 --eval_iters 2
single $gqa --total_batch_size_str "8*256*4" --dropout 0.1
single $gqa --total_batch_size_str "8*256" --act_recomp --act_recomp_scope block
single $moe --total_batch_size_str "8*256*2" --act_recomp --hip_graph off
single $moe --total_batch_size_str "8*256" --hip_graph on --dropout 0.1
single $gqa --total_batch_size_str "8*256" --grad_dtype fp32 --hip_graph on
single $gqa --total_batch_size_str "8*256" --save_interval 3 --resume auto
single $gqa --total_batch_size_str "8*256" --save_interval 3 --resume auto --max_iters 6     # second run: resumes at the end
ddp1 $gqa --total_batch_size_str "8*256*2" --grad_dtype fp32 --tied_tail split
ddp1 $gqa --total_batch_size_str "8*256" --vocab_parallel --hip_graph on
ddp1 $moe --total_batch_size_str "8*256" --comm_backend own
rm -rf $D
exit $fail
