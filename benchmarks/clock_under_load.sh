mkdir -p gpurun_out
out=gpurun_out/clock_under_load.txt
: > $out
for w in idle wgrad gemm attn; do
  timeout -k 10 60 python benchmarks/clock_under_load.py $w 5 >> $out 2>/dev/null || exit 1
done
timeout -k 10 120 python benchmarks/wgrad3_ablate.py >> $out 2>/dev/null || exit 1
cat $out
