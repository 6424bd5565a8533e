# The library as built before this session's epilogue / cross-entropy / LayerNorm / transposes changes (benchmarks/ab/_C_start_of_session.so,
# with the one-by-one transposes) against the current build, interleaved on one box: bench.py at micro-batch 256, 64, 16
set -o pipefail
mkdir -p gpurun_out/sab
for rep in 1 2; do
  for b in 256 64 16; do
    for which in old new; do
      name=${which}_b${b}_r${rep}
      steps=20; [ $b != 256 ] && steps=30
      if [ $which = old ]; then export DPA_EXT_LIB=$GRAFT_REPO_ROOT/benchmarks/ab/_C_start_of_session.so DPA_BATCH_TRANSPOSES=0; else unset DPA_EXT_LIB DPA_BATCH_TRANSPOSES; fi
      timeout -k 10 300 python bench.py --steps $steps --warmup 6 --batch $b > gpurun_out/sab/$name.log 2>&1 || { tail -5 gpurun_out/sab/$name.log; exit 1; }
      tail -1 gpurun_out/sab/$name.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', 'tok/s %.0f' % r['value'], 'ms %.2f' % r['ms_per_step'], 'mfu %.4f' % r['mfu'])"
    done
  done
done
