# same-box timing of wgrad3_kernel and its three ablation builds (benchmarks/ab_build.py wg<v> gemm_wgrad.hip -DDPA_WG3_ABLATE=<v>)
mkdir -p gpurun_out
out=gpurun_out/wgrad3_ablate.txt
: > $out
timeout -k 10 120 python benchmarks/wgrad3_ablate.py >> $out 2>&1 || exit 1
for v in ${WG3_BUILDS:-wg1 wg2 wg3}; do
  DPA_EXT_LIB=benchmarks/ab/_C_$v.so timeout -k 10 120 python benchmarks/wgrad3_ablate.py >> $out 2>&1 || exit 1
done
cat $out
