#!/bin/bash
# The structure-changing training flags held for 600 steps each through the real trainers (small model, one GPU):
# what a few-step sweep cannot see is growth. A configuration passes if it reaches step 600 with a finite loss and
# the allocator's peak ("GPU RAM" of the log line) at step 600 is no more than 2 % above that at step 100.
# -> one line per configuration on stdout
D=$(mktemp -d)
export MASTER_ADDR=127.0.0.1
common="--dataset synthetic --synthetic_tokens 2097152 --vocab_size 1024 --max_iters 600 --log_interval 100 --data_dir $D/data
 --n_embd 256 --n_layer 2 --n_head 4 --up_dim 512 --batch_size 8 --block_size 256 --total_batch_size_str 8*256"
fail=0
n=0
check() {
  name=$1; rc=$2; out=$3
  a=$(echo "$out" | grep "^step: 100 " | tail -1 | sed -n 's/.*GPU RAM: \([0-9.]*\)GB.*/\1/p')
  b=$(echo "$out" | grep "^step: 600 " | tail -1 | sed -n 's/.*GPU RAM: \([0-9.]*\)GB.*/\1/p')
  last=$(echo "$out" | grep "^step: 600 " | tail -1 | cut -c1-44)
  if [ $rc -ne 0 ] || [ -z "$b" ] || echo "$last" | grep -qi "nan" || ! python3 -c "import sys; sys.exit(0 if float('$b') <= 1.02 * float('$a') + 0.01 else 1)"; then
    echo "FAIL: $name | peak $a -> $b GB"; echo "$out" | tail -6; fail=1
  else echo "ok: $name | $last | peak $a -> $b GB"; fi
}
single() { n=$((n + 1)); out=$(timeout -k 10 300 python single-gpu/train.py $common --file_name $D/r$n "$@" 2>&1); check "single $*" $? "$out"; }
dist1() { script=$1; shift; n=$((n + 1)); out=$(DPA_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29900 + n)) "$script" $common --file_name $D/r$n "$@" 2>&1); check "$(basename $script)(1) $*" $? "$out"; }
gqa="--attn gqa --n_kv_heads 2 --pos_emb rope --non_linearity swiglu"
moe="--attn mha --pos_emb learn --non_linearity swiglu --moe --n_exp 4 --n_shared 1 --n_act 2"
mla="--attn mla --pos_emb rope --q_latent_dim 64 --kv_latent_dim 64 --rope_head_dim 32 --non_linearity swiglu"
single $gqa --hip_graph on
single $gqa --hip_graph off --dropout 0.1 --act_recomp --act_recomp_scope block
single $moe --hip_graph on --dropout 0.1 --aux_free
single $moe --hip_graph off --act_recomp
single $mla --eval --eval_interval 200 --eval_iters 2
dist1 multi-gpu/ddp/train.py $gqa --vocab_parallel --tied_tail split --grad_dtype fp32
dist1 multi-gpu/ddp/train.py $gqa --hip_graph on --comm_backend own
dist1 "multi-gpu/zero1,2/train_zero1.py" $gqa --comm_backend own
dist1 multi-gpu/fsdp/train.py $mla --act_recomp --fsdp_prefetch 2 --fsdp_inflight 3
dist1 multi-gpu/ep/train_ep.py $moe --ep_capacity 1.5 --dropout 0.1
dist1 multi-gpu/ep/train_ep_fsdp.py $moe --comm_backend own --aux_free
rm -rf $D
exit $fail
