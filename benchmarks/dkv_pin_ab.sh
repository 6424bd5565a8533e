# Build the variants first: python benchmarks/ab_build.py nopin attention.hip -DDPA_ATTN_DKV_PIN_V=0 (at the commit before the
# prologue reorder) and python benchmarks/ab_build.py noloop attention.hip -DDPA_ATTN_DKV_ABLATE=6.
# Same-box A/B of the dK/dV prologue: default library vs benchmarks/ab/_C_nopin.so (V fragments waited for inside the
# loop, first tile issued after the K/V loads) and _C_noloop.so (no tile loop at all: prologue + epilogue only).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== default"; timeout -k 10 120 python benchmarks/attn_bench.py 4 30
echo "== old prologue";  DPA_EXT_LIB=benchmarks/ab/_C_nopin.so timeout -k 10 120 python benchmarks/attn_bench.py 4 30
echo "== no tile loop (timing only)";  DPA_EXT_LIB=benchmarks/ab/_C_noloop.so timeout -k 10 120 python benchmarks/attn_bench.py 4 30
echo "== default"; timeout -k 10 120 python benchmarks/attn_bench.py 4 30
} > gpurun_out/dkv_pin.txt 2>&1
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "attn or attention or dropout or rope or gqa" > gpurun_out/dkv_pin_tests.txt 2>&1
