#!/bin/bash
# timing of the wide attention kernels against the absorbed form on stock SDPA, then one counter pass over them
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out/attn_wide
timeout -k 10 300 python3 $REPO/benchmarks/attn_wide_bench.py 2>&1 | grep -v amdgpu.ids > $REPO/gpurun_out/attn_wide/bench.txt
cat $REPO/gpurun_out/attn_wide/bench.txt
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d $REPO/gpurun_out/attn_wide/p1 -- python3 $REPO/benchmarks/attn_wide_bench.py > $REPO/gpurun_out/attn_wide/p1.log 2>&1
python3 $REPO/benchmarks/summarize_pmc_clock.py $REPO/gpurun_out/attn_wide | grep -i "attn_\|==" > $REPO/gpurun_out/attn_wide/pmc.txt
cat $REPO/gpurun_out/attn_wide/pmc.txt
