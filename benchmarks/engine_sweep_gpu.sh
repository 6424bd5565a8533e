#!/bin/bash
# The multi-GPU entry scripts with ONE rank over RCCL (DPA_FORCE_DIST=1: every collective issued) on a rotary / SwiGLU
# model and, for expert parallelism, SwiGLU experts: every script has to step and report a finite loss.
D=$(mktemp -d)
export DPA_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
common="--dataset synthetic --synthetic_tokens 1048576 --vocab_size 1024 --max_iters 4 --data_dir $D/data --file_name $D/run
 --n_embd 256 --n_layer 2 --n_head 4 --up_dim 512 --batch_size 8 --block_size 512 --total_batch_size_str 8*512"
fail=0
port=29820
run() {
  script=$1; shift
  port=$((port + 1))
  out=$(timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port "$script" $common "$@" 2>&1); rc=$?
  last=$(echo "$out" | grep "^step: 4 " | tail -1)
  if [ $rc -ne 0 ] || [ -z "$last" ] || echo "$last" | grep -qi "nan"; then echo "FAIL: $script $*"; echo "$out" | tail -6; fail=1; else echo "ok: $script $* | $(echo $last | cut -c1-44)"; fi
}
dense="--attn gqa --n_kv_heads 2 --pos_emb rope --non_linearity swiglu"
run multi-gpu/ddp/train.py $dense
run "multi-gpu/zero1,2/train_zero1.py" $dense
run "multi-gpu/zero1,2/train_zero2.py" $dense
run multi-gpu/fsdp/train.py $dense
moe="--attn mha --pos_emb learn --non_linearity swiglu --moe --n_exp 4 --n_shared 1 --n_act 2"
run multi-gpu/ep/train_ep.py $moe
run multi-gpu/ep/train_ep_fsdp.py $moe
run multi-gpu/ddp/train.py $moe
rm -rf $D
exit $fail
