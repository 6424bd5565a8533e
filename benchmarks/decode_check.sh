#!/bin/bash
# Decode path on the GPU box: tests that touch it, kernel mix of the graphed step under rocprofv3, throughput.
# usage (from the repo root, through gpurun): bash benchmarks/decode_check.sh [preset] [batch]
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
PRESET=${1:-gpt2-124m}
BATCH=${2:-8}
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof_dec
cd $REPO
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q --timeout 120 --timeout-method thread -k "decode" > $OUT/t_decode.log 2>&1
rc=$?
tail -4 $OUT/t_decode.log
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dec -o dec -- python3 $REPO/benchmarks/decode_profile.py $PRESET $BATCH 512 > $OUT/prof_dec/run.log 2>&1 || { tail -5 $OUT/prof_dec/run.log; exit 1; }
grep "new tokens" $OUT/prof_dec/run.log
cd $REPO
python benchmarks/summarize_prof.py $OUT/prof_dec 1024 ${3:-8} > $OUT/decode_kernel_mix.txt
cat $OUT/decode_kernel_mix.txt
rm -f $OUT/prof_dec/*trace.csv $OUT/prof_dec/*.db
timeout -k 10 200 python benchmarks/generate_bench.py $PRESET $BATCH 128 256 2>&1 | grep hipGraph | tee -a $OUT/decode_kernel_mix.txt
