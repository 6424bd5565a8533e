#!/bin/bash
# End-to-end CLI smoke on one GPU: the profile modes of the single-GPU entry point and a one-rank DDP job over
# each RCCL transport (collectives forced with DPA_FORCE_DIST=1). Usage: benchmarks/cli_smoke_gpu.sh <outdir>
set -e
out=${1:-gpurun_out/cli_smoke}
mkdir -p $out
cd "$(dirname "$0")/.."
common="--preset gpt2-124m --n_layer 4 --dataset synthetic --synthetic_tokens 2000000 --data_dir /tmp/dpa_cli_smoke --batch_size 8 --total_batch_size_str 8*1024 --max_iters 12 --warmup_steps 2 --log_interval 4"
timeout -k 10 200 python single-gpu/train.py $common --profile --file_name /tmp/dpa_smoke_a > $out/single_profile_events.log 2>&1
timeout -k 10 200 python single-gpu/train.py $common --profile kernels --hip_graph off --file_name /tmp/dpa_smoke_b > $out/single_profile_kernels.log 2>&1
for be in c10d own; do
  DPA_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29571 multi-gpu/ddp/train.py $common --profile --comm_backend $be --tied_tail split \
    --grad_dtype fp32 --file_name /tmp/dpa_smoke_$be > $out/ddp_${be}_profile.log 2>&1
done
tail -n 14 $out/single_profile_events.log
grep -m1 -A6 "total GPU kernel time" $out/single_profile_kernels.log | cut -c1-150
for be in c10d own; do echo "== ddp over $be"; grep -A12 "^phase (" $out/ddp_${be}_profile.log | cut -c1-150; done
