#!/bin/bash
# End-to-end numerical A/B on one GPU: the same short training run (a corpus the model can learn: one random
# 3000-token passage repeated) with the own GEMMs / fused MLP and with the library GEMMs (DPA_LIBRARY_GEMM=1).
# Prints the two loss trajectories side by side. Usage: benchmarks/own_vs_library_training.sh <outdir>
set -e
out=${1:-gpurun_out/own_vs_lib}
mkdir -p $out /tmp/dpa_ab_data
cd "$(dirname "$0")/.."
python - <<PY
import numpy as np
rng = np.random.default_rng(7)
passage = rng.integers(0, 50000, size=3000, dtype=np.uint16)
np.tile(passage, 700).tofile("/tmp/dpa_ab_data/train.bin")
np.tile(passage, 40).tofile("/tmp/dpa_ab_data/val.bin")
PY
common="--preset gpt2-124m --n_layer 4 --dataset custom --data_dir /tmp/dpa_ab_data --batch_size 8 --total_batch_size_str 8*1024 --max_iters 120 --warmup_steps 10 --learning_rate 6e-4 --log_interval 10 --hip_graph off"
timeout -k 10 300 python single-gpu/train.py $common --file_name /tmp/dpa_ab_own > $out/own.log 2>&1
DPA_LIBRARY_GEMM=1 timeout -k 10 300 python single-gpu/train.py $common --file_name /tmp/dpa_ab_lib > $out/library.log 2>&1
paste <(grep "^step" $out/own.log | cut -d'|' -f1,2) <(grep "^step" $out/library.log | cut -d'|' -f2)
