set -o pipefail
mkdir -p gpurun_out/v1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/v1/pytest_gpu.log 2>&1; rc=$?
tail -5 gpurun_out/v1/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
for tr in c10d own; do
  for mode in ddp zero2 fsdp; do
    DPA_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --mode $mode --batch 64 --comm_backend $tr > gpurun_out/v1/bench_${mode}_${tr}.log 2>&1 || { tail -5 gpurun_out/v1/bench_${mode}_${tr}.log; exit 1; }
    tail -1 gpurun_out/v1/bench_${mode}_${tr}.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['config']['parallelism'], r['transport'], 'rccl_ranks', r['rccl_ranks'], 'tok/s %.0f' % r['value'], 'ms %.2f' % r['ms_per_step'], 'exposed %.3f' % (r['exposed_comm_ms'] or 0), 'busy', r['comm_busy_ms'], 'reg', r['registered_buffers'])"
  done
done
for mode in ep ep_fsdp; do
  DPA_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --mode $mode --model gpt-moe-8x --batch 16 > gpurun_out/v1/bench_${mode}.log 2>&1 || { tail -5 gpurun_out/v1/bench_${mode}.log; exit 1; }
  tail -1 gpurun_out/v1/bench_${mode}.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['config']['parallelism'], r['transport'], 'tok/s %.0f' % r['value'], 'ms %.2f' % r['ms_per_step'], 'exposed %.3f' % (r['exposed_comm_ms'] or 0))"
done
timeout -k 10 300 python bench.py --model gpt-moe-8x --batch 16 --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('local experts', 'tok/s %.0f' % r['value'], 'ms %.2f' % r['ms_per_step'])"
