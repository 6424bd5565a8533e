# NT GEMM LDS-DMA pieces with (m0keep) and without (default build) the M0 save / restore pair, interleaved: bench.py at micro-batch 256 and 64
set -o pipefail
mkdir -p gpurun_out/m0
for rep in 1 2; do
  for b in 256 64; do
    for which in m0keep new; do
      name=${which}_b${b}_r${rep}
      if [ $which = m0keep ]; then export DPA_EXT_LIB=$GRAFT_REPO_ROOT/benchmarks/ab/_C_m0keep.so; else unset DPA_EXT_LIB; fi
      timeout -k 10 300 python bench.py --steps 20 --warmup 6 --batch $b > gpurun_out/m0/$name.log 2>&1 || { tail -5 gpurun_out/m0/$name.log; exit 1; }
      tail -1 gpurun_out/m0/$name.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', 'tok/s %.0f' % r['value'], 'ms %.2f' % r['ms_per_step'], 'mfu %.4f' % r['mfu'])"
    done
  done
done
