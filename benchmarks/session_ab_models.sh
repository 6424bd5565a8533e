# session-start library against the current build on the other presets, interleaved on one box
set -o pipefail
mkdir -p gpurun_out/sabm
run() {  # name, bench args...
  local name=$1; shift
  for which in old new; do
    if [ $which = old ]; then export DPA_EXT_LIB=$GRAFT_REPO_ROOT/benchmarks/ab/_C_start_of_session.so DPA_BATCH_TRANSPOSES=0; else unset DPA_EXT_LIB DPA_BATCH_TRANSPOSES; fi
    timeout -k 10 400 python bench.py --steps 20 --warmup 5 "$@" > gpurun_out/sabm/${which}_$name.log 2>&1 || { tail -5 gpurun_out/sabm/${which}_$name.log; exit 1; }
    tail -1 gpurun_out/sabm/${which}_$name.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('${which}_$name', 'tok/s %.0f' % r['value'], 'ms %.2f' % r['ms_per_step'], 'mfu %.4f' % r['mfu'])"
  done
}
# (the MoE presets need a library with the padded-buffer arguments of the scatter / combine-backward ops: an older build
# cannot serve the current Python side; their A/B of record is in profiles/r3_session_ab_other_presets.txt)
run gqa_rope_swiglu_b128 --model gqa-rope-swiglu-124m --batch 128
run gpt2_350m_b128 --model gpt2-350m --batch 128
run gpt2_1p3b_b48 --model gpt2-1.3b --batch 48
