#!/bin/bash
# Train a few steps, save, sample (graphed decode where the family allows it) for each model family.
D=$(mktemp -d)
fail=0
i=0
while read -r m; do
  i=$((i + 1))
  timeout -k 10 200 python single-gpu/train.py --dataset synthetic --synthetic_tokens 262144 --vocab_size 257 --block_size 128 --n_embd 256 \
    --n_layer 2 --n_head 4 --up_dim 512 --batch_size 8 --total_batch_size_str "8*128" --max_iters 3 --data_dir $D/data --file_name $D/run$i \
    --save_model $m > $D/train$i.log 2>&1 || { echo "FAIL train: $m"; tail -4 $D/train$i.log; fail=1; continue; }
  out=$(timeout -k 10 200 python single-gpu/sample.py --ckpt $D/run${i}_ckpt.pt --prompt "abc" --max_new_tokens 48 --num_samples 3 --tokenizer bytes --token_ids 2>&1); rc=$?
  if [ $rc -ne 0 ]; then echo "FAIL sample: $m"; echo "$out" | tail -5; fail=1; else echo "ok: $m | $(echo "$out" | grep '^\[sample\]' | cut -c1-110)"; fi
done <<'LIST'
--attn mha --pos_emb learn --non_linearity gelu
--attn gqa --n_kv_heads 2 --pos_emb rope --non_linearity swiglu
--attn mqa --pos_emb sin --non_linearity relu
--attn mla --pos_emb rope --q_latent_dim 32 --kv_latent_dim 32 --rope_head_dim 16 --non_linearity swiglu
--attn mla --pos_emb learn --q_latent_dim 32 --kv_latent_dim 32 --non_linearity gelu
--attn mha --pos_emb learn --non_linearity swiglu --moe --n_exp 4 --n_shared 1 --n_act 2
LIST
rm -rf $D
exit $fail
