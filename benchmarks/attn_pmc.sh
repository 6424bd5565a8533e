#!/bin/bash
# Counter passes over the attention micro-benchmark (case 4: B64 T1024 nh12 hs64): how busy the vector
# and matrix pipes are, how often both run in the same cycle, and what the waves wait for. One pass per
# rocprofv3 run (at most 8 SQ counters each). Output: gpurun_out/attn_pmc_*.txt
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CASE=${1:-4}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_BUSY_CU_CYCLES"
P3="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rm -rf /tmp/apmc$i
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d /tmp/apmc$i -o p -- python3 $REPO/benchmarks/attn_bench.py $CASE 3 > $OUT/attn_pmc_run$i.log 2>&1
  python3 $REPO/benchmarks/summarize_pmc.py /tmp/apmc$i 6 > $OUT/attn_pmc_$i.txt
done
cat $OUT/attn_pmc_1.txt $OUT/attn_pmc_2.txt $OUT/attn_pmc_3.txt
