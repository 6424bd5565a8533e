#!/bin/bash
# single-gpu/train.py for a few steps over the model families at small shapes (the dispatch sends their products to
# the library) and at shapes the own kernels take: every configuration has to start, step and report a finite loss.
D=$(mktemp -d)
common="--dataset synthetic --synthetic_tokens 1048576 --vocab_size 1024 --max_iters 4 --data_dir $D/data --file_name $D/run"
fail=0
run() {
  out=$(timeout -k 10 200 python single-gpu/train.py $common "$@" 2>&1); rc=$?
  last=$(echo "$out" | grep "^step: 4 " | tail -1)
  if [ $rc -ne 0 ] || [ -z "$last" ] || echo "$last" | grep -qi "nan"; then echo "FAIL: $*"; echo "$out" | tail -4; fail=1; else echo "ok: $* | $(echo $last | cut -c1-60)"; fi
}
run    # the reference's no-flag model: MLA + RoPE, SwiGLU, 6 layers of 256
for bs in "--batch_size 2 --block_size 256 --total_batch_size_str 2*256" "--batch_size 16 --block_size 512 --total_batch_size_str 16*512"; do
  for m in "--attn mha --pos_emb learn --non_linearity gelu" "--attn gqa --n_kv_heads 2 --pos_emb rope --non_linearity swiglu" \
           "--attn mqa --pos_emb sin --non_linearity relu" "--attn mla --pos_emb rope --q_latent_dim 32 --kv_latent_dim 32 --rope_head_dim 16 --non_linearity swiglu" \
           "--attn mha --pos_emb learn --non_linearity swiglu --moe --n_exp 4 --n_shared 1 --n_act 2" "--attn gqa --n_kv_heads 2 --pos_emb rope --non_linearity gelu --act_recomp"; do
    run --n_embd 256 --n_layer 2 --n_head 4 --up_dim 512 $bs $m
  done
done
rm -rf $D
exit $fail
