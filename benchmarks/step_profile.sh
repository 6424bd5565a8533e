#!/bin/bash
# Per-kernel time of a bench.py configuration under rocprofv3 (kernel trace + stats), summarised per training step.
# usage: bash benchmarks/step_profile.sh <name> [bench.py arguments...]      -> gpurun_out/<name>_kernel_summary.txt
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1
shift
OUT=$REPO/gpurun_out/prof_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=5
WARM=3
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $REPO/bench.py --steps $STEPS --warmup $WARM "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cd $REPO
python benchmarks/summarize_prof.py $OUT $((STEPS + WARM)) 40 > gpurun_out/${NAME}_kernel_summary.txt
rm -f $OUT/*trace.csv $OUT/*.db
head -30 gpurun_out/${NAME}_kernel_summary.txt | cut -c1-170
