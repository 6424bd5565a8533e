#!/bin/bash
# Counter passes over the NT GEMM driver for both kernel forms (one / two workgroups per CU) and two epilogues:
# MFMA-busy share, wave wait shares and the clock the chip holds (GRBM_GUI_ACTIVE / 8 / kernel time).
# Usage: gemm_nt2_pmc.sh <outdir> [N K]
set -e
out=${1:-gpurun_out/gemm2_pmc}
N=${2:-3072}; K=${3:-768}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/benchmarks/gemm_nt_driver.bin
O=$GRAFT_REPO_ROOT/$out
for form in 1 2; do
  for epi in 0 1 2; do
    export DPA_FORM=$form
    timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
      --output-format csv -d $O/f${form}_e${epi} -- $B prof 0 $N $K $epi > $O/f${form}_e${epi}.log 2>&1
  done
done
python3 $GRAFT_REPO_ROOT/benchmarks/summarize_pmc_clock.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
