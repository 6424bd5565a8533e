#!/bin/bash
# Counter passes over the 256x256 weight-gradient kernel on one shape (default: the GPT-2-124M c_fc gradient at
# micro-batch 256): MFMA-busy share, wave wait shares, clock, LDS activity and conflicts.
# Usage: wgrad_pmc.sh <outdir> [M N K]; DPA_EXT_LIB selects another build (benchmarks/ab_build.py)
set -e
out=${1:-gpurun_out/wgrad_pmc}
M=${2:-262144}; N=${3:-3072}; K=${4:-768}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/$out
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $REPO/$out/p$i -- python3 $REPO/benchmarks/wgrad_one.py $M $N $K 4 > $REPO/$out/p$i.log 2>&1
done
python3 $REPO/benchmarks/summarize_pmc_clock.py $REPO/$out > $REPO/$out/summary.txt 2>&1
python3 $REPO/benchmarks/summarize_pmc.py $REPO/$out/p2 4 >> $REPO/$out/summary.txt 2>&1 || true
cat $REPO/$out/summary.txt
