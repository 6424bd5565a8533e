set -o pipefail
R=$PWD
O=gpurun_out/many_s6.txt
: > $O
run() { echo "### $*" >> $O; timeout -k 10 280 "$@" 2>gpurun_out/many_err.log | cut -c1-420 >> $O || { echo "FAILED rc=$?" >> $O; tail -5 gpurun_out/many_err.log >> $O; return 1; }; }
run python bench.py &&
run python bench.py --model gpt-moe-8x --batch 64 --steps 15 --warmup 4 &&
run python bench.py --model gpt2-350m --mode zero2 --batch 32 --steps 15 --warmup 4 &&
run python bench.py --model gpt2-1.3b --mode fsdp --batch 16 --steps 10 --warmup 3 &&
run python bench.py --model gqa-rope-swiglu-124m --batch 64 --steps 15 --warmup 4
cat $O
