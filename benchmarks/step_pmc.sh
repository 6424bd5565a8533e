#!/bin/bash
# Hardware counters per kernel of a bench.py configuration (rocprofv3 --pmc, one pass, six SQ counters), summarised
# per kernel name: MFMA-busy share and LDS bank conflicts per LDS instruction.
# usage: bash benchmarks/step_pmc.sh <name> [bench.py arguments...]      -> gpurun_out/<name>_pmc_summary.txt
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1
shift
OUT=$REPO/gpurun_out/pmc_$NAME
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 500 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES \
  --output-format csv -d $OUT -o p -- python3 $REPO/bench.py --steps 3 --warmup 2 "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cd $REPO
(echo "# rocprofv3 --pmc over bench.py $* (5 steps): mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)"; python benchmarks/summarize_pmc.py $OUT 24) > gpurun_out/${NAME}_pmc_summary.txt
rm -rf $OUT
grep "mfma_util" gpurun_out/${NAME}_pmc_summary.txt | cut -c1-140 | head -16
