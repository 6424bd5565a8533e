#!/bin/bash
# End-to-end numerical A/B on one GPU: the same short training run of a GQA + RoPE + SwiGLU model (a corpus the model
# can learn: one random 3000-token passage repeated) with the rotary embedding inside the qkv GEMM's epilogue / the
# attention backward's stores, and with separate rotary passes (DPA_FUSED_ROPE=0). Prints the two loss trajectories
# side by side. Usage: benchmarks/fused_rope_training_ab.sh <outdir>
set -e
out=${1:-gpurun_out/rope_ab}
mkdir -p $out /tmp/dpa_ab_data
cd "$(dirname "$0")/.."
python - <<PY
import numpy as np
rng = np.random.default_rng(7)
passage = rng.integers(0, 50000, size=3000, dtype=np.uint16)
np.tile(passage, 700).tofile("/tmp/dpa_ab_data/train.bin")
np.tile(passage, 40).tofile("/tmp/dpa_ab_data/val.bin")
PY
common="--preset gqa-rope-swiglu-124m --n_layer 4 --dataset custom --data_dir /tmp/dpa_ab_data --batch_size 8 --total_batch_size_str 8*1024 --max_iters 120 --warmup_steps 10 --learning_rate 6e-4 --log_interval 10 --hip_graph off"
timeout -k 10 300 python single-gpu/train.py $common --file_name /tmp/dpa_ab_fused > $out/fused.log 2>&1
DPA_FUSED_ROPE=0 timeout -k 10 300 python single-gpu/train.py $common --file_name /tmp/dpa_ab_sep > $out/separate.log 2>&1
echo "step | rotary in the GEMM epilogue / attention backward stores | separate rotary passes"
paste <(grep "^step" $out/fused.log | cut -d'|' -f1,2) <(grep "^step" $out/separate.log | cut -d'|' -f2)
