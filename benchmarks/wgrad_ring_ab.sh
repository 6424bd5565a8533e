# wgrad3_kernel builds side by side on one box: default build, then benchmarks/ab/_C_<name>.so for each name in WG3_BUILDS
# (python benchmarks/ab_build.py <name> gemm_wgrad.hip -D...), on the 124M shapes and the 1.3B shapes against the library
mkdir -p gpurun_out
out=gpurun_out/wgrad_ring_ab.txt
: > $out
for v in default ${WG3_BUILDS}; do
  lib=; [ $v = default ] || lib=benchmarks/ab/_C_$v.so
  echo "## build: $v" >> $out
  DPA_EXT_LIB=$lib timeout -k 10 120 python benchmarks/wgrad3_ablate.py 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  DPA_EXT_LIB=$lib WGRAD_CASES="1.3b;1.3b, micro-batch 48" timeout -k 10 120 python benchmarks/wgrad_shapes_bench.py 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
cat $out
