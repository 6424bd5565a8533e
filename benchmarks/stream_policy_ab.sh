# LayerNorm kernels with and without the streaming cache policy (DPA_LN_STREAM_MB: activations from that size on are streamed),
# micro-batch 256 (402 MB per activation) and 64 (101 MB)
set -o pipefail
mkdir -p gpurun_out/stream
for b in 256 64; do
  for mb in 1000000 256 0; do
    name=b${b}_mb${mb}
    DPA_LN_STREAM_MB=$mb timeout -k 10 300 python bench.py --steps 24 --warmup 6 --batch $b > gpurun_out/stream/$name.log 2>&1 || { tail -5 gpurun_out/stream/$name.log; exit 1; }
    tail -1 gpurun_out/stream/$name.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', 'tok/s %.0f' % r['value'], 'ms %.2f' % r['ms_per_step'], 'mfu %.4f' % r['mfu'])"
  done
done
