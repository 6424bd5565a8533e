# the benchmark configurations on one box: headline (GPT-2-124M), MoE 8x, 350M ZeRO-2, 1.3B (single at 48, FSDP at 16),
# GQA + RoPE + SwiGLU, MoE at micro-batch 256, the wide latent-attention configuration
set -o pipefail
O=gpurun_out/many_r3.txt
: > $O
run() { echo "### $*" >> $O; timeout -k 10 280 "$@" 2>gpurun_out/many_err.log | python -c "
import json,sys
for line in sys.stdin:
    try: r=json.loads(line)
    except Exception: continue
    print('tok/s %.0f  ms %.2f  mfu %.3f  mem %.1f GB  %s %s' % (r['value'], r['ms_per_step'], r['mfu'], r['mem_gb'], r['config']['model'], r['config']['parallelism']))
" >> $O || { echo "FAILED rc=$?" >> $O; tail -5 gpurun_out/many_err.log >> $O; return 1; }; }
run python bench.py --steps 20 --warmup 5 &&
run python bench.py --batch 64 --steps 20 --warmup 5 &&
run python bench.py --model gpt-moe-8x --batch 64 --steps 15 --warmup 4 &&
run python bench.py --model gpt-moe-8x --batch 256 --steps 8 --warmup 3 &&
run python bench.py --model gpt-moe-8x-swiglu --batch 64 --steps 15 --warmup 4 &&
run python bench.py --model gpt2-350m --mode zero2 --batch 32 --steps 15 --warmup 4 &&
run python bench.py --model gpt2-350m --mode single --batch 128 --steps 8 --warmup 3 &&
run python bench.py --model gpt2-1.3b --mode fsdp --batch 16 --steps 10 --warmup 3 &&
run python bench.py --model gpt2-1.3b --mode single --batch 48 --steps 6 --warmup 2 &&
run python bench.py --model gqa-rope-swiglu-124m --batch 64 --steps 15 --warmup 4 &&
run python bench.py --model gqa-rope-swiglu-124m --batch 256 --steps 8 --warmup 3
cat $O
