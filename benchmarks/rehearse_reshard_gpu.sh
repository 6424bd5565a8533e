#!/bin/bash
# Restart on another number of ranks, on the GPU (ranks share one MI355X over host-staged gloo): an 8-rank run writes
# sharded resume files, the same command line with 4 ranks (then 2, then 8 again) re-cuts the optimizer shards and trains on.
export DPA_BACKEND=gloo MASTER_ADDR=127.0.0.1 DPA_HANG_DUMP_S=150 OMP_NUM_THREADS=2
out=$(mktemp -d /tmp/dpa_reshard.XXXX)
port=29740
common="--dataset synthetic --synthetic_tokens 524288 --batch_size 2 --total_batch_size_str 2*256*8
        --warmup_steps 2 --save_interval 4 --resume auto --data_dir $out/data"
fail=0
leg() {   # entry, name, ranks, max_iters
  port=$((port + 1))
  log="$out/$2.$3.log"
  timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $3 --master-addr 127.0.0.1 --master-port $port \
    "$1" ${model:---preset char-10m} $common --file_name "$out/$2" --max_iters $4 > "$log" 2>&1
  rc=$?
  res=$(grep -h "resumed from\|starting fresh" "$log" | head -1 | sed 's/.*\(at iteration [0-9]*\|starting fresh\).*/\1/')
  last=$(grep "^step: $4 " "$log" | tail -1 | cut -c1-75)
  if [ $rc -eq 0 ] && [ -n "$last" ]; then echo "ok: $2 with $3 ranks | $res | $last"; else echo "FAIL: $2 with $3 ranks (exit $rc)"; grep -B2 -A25 -m1 "Traceback" "$log" | cut -c1-200; fail=1; fi
}
for e in "multi-gpu/zero1,2/train_zero2.py zero2" "multi-gpu/fsdp/train.py fsdp" "multi-gpu/ddp/train.py ddp" "multi-gpu/ep/train_ep.py ep" "multi-gpu/ep/train_ep_fsdp.py ep_fsdp"; do
  set -- $e
  model=""
  case $2 in ep*) model="--vocab_size 1024 --block_size 256 --n_embd 256 --n_layer 2 --n_head 4 --up_dim 512 --attn mha --pos_emb learn --non_linearity swiglu --moe --n_exp 9 --n_shared 1 --n_act 2";; esac
  leg "$1" $2 8 9
  leg "$1" $2 4 14
  leg "$1" $2 2 18
  leg "$1" $2 8 22
done
rm -rf "$out"
exit $fail
