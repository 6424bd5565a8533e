#!/bin/bash
# Counter passes over the NT GEMM driver (one shape, stores on / dropped). Usage: gemm_nt_pmc.sh <outdir>
set -e
out=${1:-gpurun_out/gemm_pmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/benchmarks/gemm_nt_driver.bin
O=$GRAFT_REPO_ROOT/$out
run() {  # name variant counters...
  name=$1; var=$2; shift 2
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- $B prof $var 3072 768 0 > $O/$name.log 2>&1
}
for var in 0 2; do
  run sq_v$var $var SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU
  run tcc_v$var $var TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
  run grbm_v$var $var GRBM_GUI_ACTIVE TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum
done
ls -R $O | head -50
