# kernels touched by the weight-gradient / head-loss work: their tests, then the headline bench and the 1.3B bench
set -o pipefail
mkdir -p gpurun_out/v2
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 240 --timeout-method thread -k "wgrad or linear_cross_entropy or gemm_nt or grouped or mlp or linear" > gpurun_out/v2/pytest_kernels.log 2>&1; rc=$?
tail -5 gpurun_out/v2/pytest_kernels.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/v2/bench_b256.log 2>&1 || { tail -5 gpurun_out/v2/bench_b256.log; exit 1; }
tail -1 gpurun_out/v2/bench_b256.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --batch 64 > gpurun_out/v2/bench_b64.log 2>&1 || { tail -5 gpurun_out/v2/bench_b64.log; exit 1; }
tail -1 gpurun_out/v2/bench_b64.log
timeout -k 10 400 python bench.py --model gpt2-1.3b --mode single --batch 48 --steps 5 --warmup 2 > gpurun_out/v2/bench_1p3b.log 2>&1 || { tail -5 gpurun_out/v2/bench_1p3b.log; exit 1; }
tail -1 gpurun_out/v2/bench_1p3b.log
