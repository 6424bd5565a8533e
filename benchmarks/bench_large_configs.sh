# larger GPT-2 presets at micro-batches that use the 288 GB of one MI355X (single process, no sharding needed)
mkdir -p gpurun_out
out=gpurun_out/bench_large_configs.txt
: > $out
run() { echo "### python bench.py $*" >> $out; timeout -k 10 280 python bench.py "$@" >> $out 2>gpurun_out/bench_large_err.log || { tail -5 gpurun_out/bench_large_err.log >> $out; return 1; }; }
run --model gpt2-350m --mode single --batch 128 --steps 10 --warmup 3 || exit 1
run --model gpt2-1.3b --mode single --batch 48 --steps 8 --warmup 3 || exit 1
run --model gqa-rope-swiglu-124m --batch 256 --steps 10 --warmup 3 || exit 1
python - <<'PY'
import json
for ln in open("gpurun_out/bench_large_configs.txt"):
    if ln.startswith("###"): print(ln.strip())
    elif ln.startswith("{"):
        d = json.loads(ln); print(f'   {d["value"]:.0f} tok/s  {d["ms_per_step"]:.1f} ms/step  MFU {d["mfu"]:.3f}  mem {d["mem_gb"]:.1f} GB')
PY
