# Sustained run through the real trainer (single-gpu/train.py): GPT-2-124M, micro-batch 256 x 1024, 1000 optimizer steps
# with evaluation every 250 steps and the checkpoint written at the end; the log (loss, tokens/s and MFU per 25 steps)
# goes to gpurun_out/sustained_trainer_run.log. Shows the throughput the bench reports holding over minutes at the power
# cap, data loader, LR schedule, logging and evaluation included.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u single-gpu/train.py --preset gpt2-124m --dataset synthetic --synthetic_tokens 33554432 \
  --data_dir /tmp/dpa_sustained_data --batch_size 256 --total_batch_size_str "256*1024" --max_iters 1000 \
  --learning_rate 6e-4 --warmup_steps 100 --grad_clip 1.0 --eval --eval_interval 250 --eval_iters 10 \
  --log_interval 25 --save_model --file_name /tmp/dpa_sustained > gpurun_out/sustained_trainer_run.log 2>&1
rc=$?
ls -la /tmp/dpa_sustained* >> gpurun_out/sustained_trainer_run.log 2>&1
exit $rc
