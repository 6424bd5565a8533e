#!/bin/bash
# Sustained runs of the other benchmark configurations through the REAL trainers with one rank over RCCL
# (DPA_FORCE_DIST=1: every collective of the engine is issued), evaluation and periodic resume files included, and a
# restart drill on the GPU: the rank dies once mid-run (DPA_FAULT_AT_ITER), torchrun relaunches the command line and
# --resume auto continues from the last committed resume generation.
#   1. GPT-2-350M, ZeRO-2, micro-batch 32, 300 steps, dies at 150
#   2. GPT MoE 8x top-2, expert parallel + FSDP dense layers, micro-batch 64, 200 steps, dies at 110
#   3. GPT-2-1.3B, FSDP, micro-batch 16, 200 steps (no resume files: 18 GB each)
# -> gpurun_out/sustained_engines/*.log and a summary on stdout
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/sustained_engines
mkdir -p $O
D=$(mktemp -d /tmp/dpa_sus.XXXX)
export DPA_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
common="--dataset synthetic --synthetic_tokens 33554432 --data_dir $D/data --learning_rate 3e-4 --warmup_steps 20 --grad_clip 1.0 --log_interval 25"
port=29810
drill() {   # name, fault iteration, script, flags...
  local name=$1 fault=$2 script=$3; shift 3
  port=$((port + 1))
  DPA_FAULT_AT_ITER=$fault timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port $port --max-restarts 1 "$script" $common --file_name $D/$name "$@" > $O/$name.log 2>&1
  local rc=$?
  echo "== $name: exit $rc"
  grep -h "fault injection\|resumed from" $O/$name.log | cut -c1-200
  grep -h "^step: " $O/$name.log | tail -3 | cut -c1-220
  rm -f $D/$name*
  return $rc
}
drill zero2_350m 150 "multi-gpu/zero1,2/train_zero2.py" --preset gpt2-350m --batch_size 32 --total_batch_size_str "32*1024" \
  --max_iters 300 --eval --eval_interval 100 --eval_iters 5 --save_interval 100 --resume auto || exit 1
drill ep_fsdp_moe8x 110 multi-gpu/ep/train_ep_fsdp.py --preset gpt-moe-8x --batch_size 64 --total_batch_size_str "64*1024" \
  --max_iters 200 --eval --eval_interval 100 --eval_iters 5 --save_interval 80 --resume auto || exit 1
drill fsdp_1p3b -1 multi-gpu/fsdp/train.py --preset gpt2-1.3b --batch_size 16 --total_batch_size_str "16*1024" \
  --max_iters 200 --eval --eval_interval 100 --eval_iters 5 || exit 1
rm -rf $D
