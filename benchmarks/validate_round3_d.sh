# whole GPU suite, then headline bench
set -o pipefail
mkdir -p gpurun_out/v4
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/v4/pytest_gpu.log 2>&1; rc=$?
tail -12 gpurun_out/v4/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/v4/bench_b256.log 2>&1 || { tail -5 gpurun_out/v4/bench_b256.log; exit 1; }
tail -1 gpurun_out/v4/bench_b256.log | cut -c1-400
