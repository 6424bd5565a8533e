# NT GEMM epilogue stores with the default cache policy (build -DDPA_GEMM_STORE_AUX=0) against non-temporal (the default build)
# at small micro-batches, where an output fits the 256 MiB memory-side cache and its reader follows at once
set -o pipefail
mkdir -p gpurun_out/stp
for rep in 1 2; do
  for b in 8 16 32 64; do
    for which in st0 nt; do
      name=${which}_b${b}_r${rep}
      if [ $which = st0 ]; then export DPA_EXT_LIB=$GRAFT_REPO_ROOT/benchmarks/ab/_C_st0.so; else unset DPA_EXT_LIB; fi
      timeout -k 10 300 python bench.py --steps 30 --warmup 6 --batch $b > gpurun_out/stp/$name.log 2>&1 || { tail -5 gpurun_out/stp/$name.log; exit 1; }
      tail -1 gpurun_out/stp/$name.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', 'tok/s %.0f' % r['value'], 'ms %.2f' % r['ms_per_step'], 'mfu %.4f' % r['mfu'])"
    done
  done
done
