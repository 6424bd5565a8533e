# one-rank RCCL rehearsal: replicated head against the vocabulary-parallel head (same FLOPs at one rank; the split head's
# collectives are issued with one participant)
set -o pipefail
mkdir -p gpurun_out/vp
for vp in "" "--vocab_parallel"; do
  name=rep; [ -n "$vp" ] && name=vp
  DPA_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --steps 10 --warmup 3 --mode ddp --batch 64 $vp > gpurun_out/vp/bench_$name.log 2>&1 || { tail -5 gpurun_out/vp/bench_$name.log; exit 1; }
  tail -1 gpurun_out/vp/bench_$name.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', 'tok/s %.0f' % r['value'], 'ms %.2f' % r['ms_per_step'], 'bytes', r['comm_bytes_per_step'])"
done
