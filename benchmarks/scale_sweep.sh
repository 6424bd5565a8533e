#!/bin/bash
# Weak-scaling sweep on ONE node with N in {1, 2, 4, 8} GPUs: every engine, both RCCL transports, the launch line the
# round-end driver uses. Prints the command lines (default), or runs them with RUN=1 and collects the JSON lines.
#
#   bash benchmarks/scale_sweep.sh                      # print the 32 command lines
#   RUN=1 bash benchmarks/scale_sweep.sh                # run all of them, one after the other -> $OUT
#   RUN=1 GPUS="1 8" MODES="ddp fsdp" TRANSPORTS=c10d bash benchmarks/scale_sweep.sh
#
# Per run bench.py prints one JSON line with tokens/s (whole job), MFU, rccl_ranks (null unless the count travelled
# through RCCL), transport, comm_bytes_per_step, exposed_comm_ms (compute stream waiting for collectives, per step,
# max over ranks) and, on the own communicator, comm_busy_ms (device time of the collectives): scaling efficiency
# = value(N) / (N * value(1)), and what is missing from 100 % is exposed_comm_ms / ms_per_step unless the per-GPU
# compute slowed down (compare ms_per_step - exposed_comm_ms across N).
set -u
cd "$(dirname "$0")/.."
GPUS=${GPUS:-"1 2 4 8"}
TRANSPORTS=${TRANSPORTS:-"c10d own"}
MODES=${MODES:-"ddp zero2 fsdp ep_fsdp"}
STEPS=${STEPS:-50}
WARMUP=${WARMUP:-10}
OUT=${OUT:-gpurun_out/scale_sweep.jsonl}
PORT=${PORT:-29600}
export HSA_ENABLE_IPC_MODE_LEGACY=0
model_of() {   # the benchmark configuration of each engine: model preset and per-GPU micro-batch
  case $1 in
    ddp) echo "--model gpt2-124m --batch 256" ;;
    zero1|zero2) echo "--model gpt2-350m --batch 64" ;;
    fsdp|fsdp_keep) echo "--model gpt2-1.3b --batch 32" ;;
    ep|ep_fsdp) echo "--model gpt-moe-8x --batch 64" ;;
  esac
}
[ "${RUN:-0}" = 1 ] && mkdir -p "$(dirname "$OUT")" && : > "$OUT"
for mode in $MODES; do
  for tr in $TRANSPORTS; do
    for n in $GPUS; do
      args="bench.py --gpus $n --steps $STEPS --warmup $WARMUP --mode $mode $(model_of $mode) --comm_backend $tr"
      if [ "$n" = 1 ]; then
        # one rank: DPA_FORCE_DIST=1 issues every collective through RCCL anyway (the engine's own overhead)
        cmd="DPA_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $PORT $args"
      else
        cmd="python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT $args"
      fi
      echo "$cmd"
      if [ "${RUN:-0}" = 1 ]; then
        if ! timeout -k 10 900 bash -c "$cmd" 2> "$OUT.$mode.$tr.$n.err" | tail -1 >> "$OUT"; then
          echo "FAILED (see $OUT.$mode.$tr.$n.err); stopping" >&2
          exit 1
        fi
      fi
      PORT=$((PORT + 1))
    done
  done
done
[ "${RUN:-0}" = 1 ] && python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
base = {}
print(f"{'mode':10s} {'transport':10s} {'N':>2s} {'tokens/s':>12s} {'MFU':>6s} {'ms/step':>8s} {'exposed ms':>10s} {'eff.':>6s}")
for r in rows:
    par, n = r["config"]["parallelism"], r["n_gpus"]
    mode = par.rstrip("0123456789") if n > 1 else "n=1"
    key = (r["config"]["model"], r.get("transport"))
    if n == 1:
        base[key] = r["value"]
    eff = r["value"] / (n * base[key]) if key in base else float("nan")
    print(f"{par:10s} {str(r.get('transport')):10s} {n:2d} {r['value']:12.0f} {100 * r['mfu']:5.1f}% {r['ms_per_step']:8.2f} "
          f"{(r.get('exposed_comm_ms') or 0):10.2f} {100 * eff:5.1f}%")
PY
exit 0
