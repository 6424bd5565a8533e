#!/bin/bash
# Wall time of a training run with periodic resume files, written synchronously against --save_async (snapshot to host
# memory, background writer, commit by the loop's poll): GPT-2-350M, micro-batch 32, 60 steps, a 4.3 GB file every 10 steps.
D=$(mktemp -d)
C="--preset gpt2-350m --dataset synthetic --synthetic_tokens 8388608 --data_dir $D/data --batch_size 32 --total_batch_size_str 32*1024 --max_iters 60 --log_interval 20 --warmup_steps 5"
for mode in "none" "sync" "async"; do
  case $mode in none) F="";; sync) F="--save_interval 10";; async) F="--save_interval 10 --save_async";; esac
  t0=$(date +%s.%N)
  timeout -k 10 400 python single-gpu/train.py $C $F --file_name $D/$mode > $D/$mode.log 2>&1 || { tail -5 $D/$mode.log; exit 1; }
  t1=$(date +%s.%N)
  echo "$mode: wall $(python3 -c "print('%.1f' % ($t1 - $t0))") s | $(grep '^step: 60 ' $D/$mode.log | cut -c1-60) | files: $(ls $D | grep -c "${mode}_resume.it")"
done
rm -rf $D
