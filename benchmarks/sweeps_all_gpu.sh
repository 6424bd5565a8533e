#!/bin/bash
# All end-to-end sweeps in one call (model families, trainer flags, engines through one-rank RCCL, MoE flags, sampling),
# each section headed by its script's name: -> gpurun_out/sweeps_all.txt
mkdir -p gpurun_out
out=gpurun_out/sweeps_all.txt
: > $out
fail=0
for s in config_sweep_gpu flag_sweep_gpu engine_sweep_gpu moe_flag_sweep_gpu sample_sweep_gpu; do
  echo "## $s" >> $out
  timeout -k 10 900 bash benchmarks/$s.sh >> $out 2>&1 || { fail=1; echo "(section failed)" >> $out; break; }
done
grep -c "^ok" $out
grep -n "FAIL\|failed" $out | head
exit $fail
