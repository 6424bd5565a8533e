# GELU epilogue forms of the NT GEMM on the c_fc shape (build: hipcc ... -DDPA_GELU_VARIANT=n benchmarks/gemm_nt_driver.hip csrc/gemm_nt.hip):
# 1 packed pairs with compare + select, 2 sign by bit copy, 3 = 2 with one reciprocal per pair; timing-only ablations 10 (no
# transcendental instructions) and 11 (no arithmetic)
set -o pipefail
for v in ${GELU_VARIANTS:-1 2 3 10 11}; do
  echo "==== DPA_GELU_VARIANT=$v"
  chk=1; [ $v -ge 10 ] && chk=0
  timeout -k 10 120 benchmarks/ab/gemm_gelu_v$v.bin gelu $chk || exit 1
done
