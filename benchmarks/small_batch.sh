# small micro-batches: eager launches against the step replayed from one hipGraph
set -o pipefail
mkdir -p gpurun_out/sb
for b in 16 32; do
  for g in "" "--hip_graph"; do
    name=b${b}; [ -n "$g" ] && name=b${b}_graph
    timeout -k 10 300 python bench.py --steps 30 --warmup 6 --batch $b $g > gpurun_out/sb/$name.log 2>&1 || { tail -5 gpurun_out/sb/$name.log; exit 1; }
    tail -1 gpurun_out/sb/$name.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', 'tok/s %.0f' % r['value'], 'ms %.2f' % r['ms_per_step'], 'mfu %.3f' % r['mfu'])"
  done
done
bash benchmarks/step_profile.sh r3_b16 --batch 16 > /dev/null && head -45 gpurun_out/r3_b16_kernel_summary.txt | cut -c1-150
