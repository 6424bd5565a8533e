#!/bin/bash
# MoE options through single-gpu/train.py (auxiliary-loss-free balancing, shared / active expert counts, activations,
# attention families under the experts): each has to step and report a finite loss.
D=$(mktemp -d)
common="--dataset synthetic --synthetic_tokens 1048576 --vocab_size 1024 --max_iters 6 --data_dir $D/data
 --n_embd 256 --n_layer 2 --n_head 4 --up_dim 512 --batch_size 8 --block_size 256 --total_batch_size_str 8*256 --moe"
fail=0
n=0
run() {
  n=$((n + 1))
  out=$(timeout -k 10 240 python single-gpu/train.py $common --file_name $D/r$n "$@" 2>&1); rc=$?
  last=$(echo "$out" | grep "^step: 6 " | tail -1)
  if [ $rc -ne 0 ] || [ -z "$last" ] || echo "$last" | grep -qi "nan"; then echo "FAIL: $*"; echo "$out" | tail -6; fail=1; else echo "ok: $* | $(echo $last | cut -c1-40)"; fi
}
run --n_exp 8 --n_shared 0 --n_act 2 --non_linearity gelu --attn mha --pos_emb learn
run --n_exp 8 --n_shared 2 --n_act 4 --non_linearity swiglu --attn gqa --n_kv_heads 2 --pos_emb rope
run --n_exp 4 --n_shared 1 --n_act 2 --non_linearity swiglu --aux_free --attn mha --pos_emb learn
run --n_exp 4 --n_shared 1 --n_act 2 --non_linearity gelu --aux_free --gamma 0.01 --attn mla --pos_emb rope --q_latent_dim 32 --kv_latent_dim 32 --rope_head_dim 16
run --n_exp 16 --n_shared 1 --n_act 3 --non_linearity relu --attn mqa --pos_emb sin
run --n_exp 4 --n_shared 1 --n_act 2 --non_linearity swiglu --coeff 0.05 --eval --eval_interval 3 --eval_iters 2
rm -rf $D
exit $fail
