#!/bin/bash
# Kernel mix of the expert-parallel step with ONE rank (RCCL exchanges issued, nothing on the wire) under rocprofv3.
# The rendezvous variables are exported here: the profiled program is bench.py itself, not a launcher.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_ep1
mkdir -p $OUT
export DPA_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29791
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $REPO/bench.py --gpus 1 --model gpt-moe-8x --batch 64 --mode ep --steps 5 --warmup 3 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cd $REPO
python benchmarks/summarize_prof.py $OUT 8 45 > gpurun_out/ep1_kernel_summary.txt
rm -f $OUT/*trace.csv $OUT/*.db
head -48 gpurun_out/ep1_kernel_summary.txt | cut -c1-170
