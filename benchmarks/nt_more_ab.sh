# non-temporal stores of the attention outputs / non-temporal accesses of the optimizer pass against the default build, interleaved
set -o pipefail
mkdir -p gpurun_out/ntm
for rep in 1 2; do
  for which in base attnnt optnt; do
    name=${which}_r${rep}
    if [ $which = base ]; then unset DPA_EXT_LIB; else export DPA_EXT_LIB=$GRAFT_REPO_ROOT/benchmarks/ab/_C_$which.so; fi
    timeout -k 10 300 python bench.py --steps 24 --warmup 6 > gpurun_out/ntm/$name.log 2>&1 || { tail -5 gpurun_out/ntm/$name.log; exit 1; }
    tail -1 gpurun_out/ntm/$name.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', 'tok/s %.0f' % r['value'], 'ms %.2f' % r['ms_per_step'], 'mfu %.4f' % r['mfu'])"
  done
done
