# bf16 flat gradients (split slabs + finish kernels) against fp32 flat gradients (weight-gradient kernels add their
# partial sums straight into the buffer) at micro-batch 16 / 32 / 256
set -o pipefail
mkdir -p gpurun_out/g32
for b in 16 32 256; do
  for g in bf16 fp32; do
    name=b${b}_${g}
    steps=30; [ $b = 256 ] && steps=20
    timeout -k 10 300 python bench.py --steps $steps --warmup 6 --batch $b --grad_dtype $g > gpurun_out/g32/$name.log 2>&1 || { tail -5 gpurun_out/g32/$name.log; exit 1; }
    tail -1 gpurun_out/g32/$name.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', 'tok/s %.0f' % r['value'], 'ms %.2f' % r['ms_per_step'], 'mfu %.3f' % r['mfu'])"
  done
done
