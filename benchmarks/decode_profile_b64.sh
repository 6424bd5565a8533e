REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_decode_b64
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $REPO/benchmarks/decode_profile.py gpt2-124m 64 256 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cd $REPO
python benchmarks/summarize_prof.py $OUT 512 16 > gpurun_out/decode_b64_kernel_summary.txt
rm -f $OUT/*trace.csv $OUT/*.db
cat gpurun_out/decode_b64_kernel_summary.txt | cut -c1-160
