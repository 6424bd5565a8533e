#!/bin/bash
# The multi-GPU training entry points end to end with EIGHT ranks sharing one GPU (host-staged gloo), at the world size
# of a node: training with gradient accumulation, evaluation, periodic SHARDED resume files (one per rank, committed by
# the marker protocol), a rank that dies mid-run (DPA_FAULT_AT_ITER: the last rank exits), the relaunch by torchrun and
# the continuation from the last committed generation, and the final gathered checkpoint in the reference's format.
# usage: bash benchmarks/rehearse_8rank_trainers.sh [scratch dir]
export DPA_BACKEND=gloo MASTER_ADDR=127.0.0.1 DPA_HANG_DUMP_S=150 OMP_NUM_THREADS=2 DPA_FAULT_AT_ITER=9
out=${1:-$(mktemp -d /tmp/dpa_rehearse8.XXXX)}
mkdir -p "$out"
port=29680
dense="--preset char-10m --dataset synthetic --synthetic_tokens 524288 --batch_size 2 --total_batch_size_str 2*256*8*2"
moe="--dataset synthetic --synthetic_tokens 524288 --vocab_size 1024 --block_size 256 --n_embd 256 --n_layer 2 --n_head 4 --up_dim 512
     --attn mha --pos_emb learn --non_linearity swiglu --moe --n_exp 9 --n_shared 1 --n_act 2 --batch_size 2 --total_batch_size_str 2*256*8"
common="--max_iters 12 --eval --eval_interval 6 --eval_iters 2 --warmup_steps 2 --save_model --save_interval 4 --resume auto ${SAVE_ASYNC:+--save_async}"
fail=0
run() {   # entry, model flags...
  entry=$1; shift
  port=$((port + 1))
  name=$(basename "$(dirname "$entry")" | tr -d ',')_$(basename "$entry" .py)   # (ddp/train.py and fsdp/train.py must not share resume files)
  log="$out/$name.log"
  timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --max-restarts 1 --master-addr 127.0.0.1 \
    --master-port $port "$entry" "$@" $common --file_name "$out/$name" --data_dir "$out/data" > "$log" 2>&1
  rc=$?
  died=$(grep -c "fault injection" "$log"); resumed=$(grep -h "resumed from" "$log" | head -1 | sed 's/.*at iteration/at iteration/')
  last=$(grep "^step: 12 " "$log" | tail -1 | cut -c1-40)
  ck=$(ls "$out/${name}_ckpt.pt" 2>/dev/null | wc -l)
  if [ $rc -eq 0 ] && [ "$died" -ge 1 ] && [ -n "$resumed" ] && [ -n "$last" ] && [ "$ck" = 1 ]; then echo "ok: $entry | rank died: $died | resumed $resumed | $last | checkpoint written"
  else echo "FAIL: $entry (exit $rc, died $died, resumed '$resumed', last '$last', checkpoint $ck)"; grep -B2 -A28 -m1 "Traceback" "$log" | cut -c1-220; fail=1; fi
}
if [ -z "$ONLY_SHARDED" ]; then
run multi-gpu/ddp/train.py $dense
run "multi-gpu/zero1,2/train_zero1.py" $dense
run "multi-gpu/zero1,2/train_zero2.py" $dense
fi
run multi-gpu/fsdp/train.py $dense
run multi-gpu/ep/train_ep.py $moe
run multi-gpu/ep/train_ep_fsdp.py $moe
rm -rf "$out"
exit $fail
