# attention: wide (key, value) widths, benchmarked shapes, underflow cases; latent-attention model variants
set -o pipefail
mkdir -p gpurun_out/v3
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q --timeout 300 --timeout-method thread -k "attention or mla or rope or gqa" > gpurun_out/v3/pytest_attn.log 2>&1; rc=$?
tail -15 gpurun_out/v3/pytest_attn.log
exit $rc
