set -o pipefail
mkdir -p gpurun_out/final
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/final/pytest_gpu.log 2>&1; rc=$?
tail -3 gpurun_out/final/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; rc=$?
tail -2 gpurun_out/final/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/final/bench.log 2>&1; rc=$?
tail -1 gpurun_out/final/bench.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_torchrun.log 2>&1; rc=$?
tail -1 gpurun_out/final/bench_torchrun.log
exit $rc
