"""Bytes on the wire per rank and optimizer step: replicated tied head against the vocabulary-parallel head of a
data-parallel job (ring collectives: an all-reduce of B bytes sends 2 (W - 1) / W B per rank, an all-gather or
reduce-scatter of a result of B bytes (W - 1) / W B). Pure arithmetic; `bench.py --vocab_parallel` reports the bytes
its collectives were actually issued with in `comm_bytes_per_step`."""
V, C = 50304, 768
print(f"{'W':>2} {'tokens/rank':>11} | {'replicated head (MB)':>20} | {'vocab-parallel: rows + stats (MB)':>34} | ratio")
for W in (2, 4, 8):
    for tokens in (4096, V // W, 16384, 65536, 262144):
        head_allreduce = 2 * (W - 1) / W * V * C * 2
        rows = 2 * (W - 1) * tokens * C * 2          # all-gather of hidden rows + reduce-scatter of their gradient
        stats = (W - 1) * tokens * 3 * 4 * W / W      # one all-gather of (max, sum-exp, target logit) per gathered row
        tgts = (W - 1) * tokens * 8
        vp = rows + stats + tgts
        print(f"{W:>2} {tokens:>11} | {head_allreduce / 1e6:>20.1f} | {vp / 1e6:>34.1f} | {vp / head_allreduce:5.2f}x")
print("break-even: tokens per rank = V / W; with tied embeddings the [V, C] gradient of the embedding is reduced anyway,")
print("so the replicated column is not actually saved by the split head.")
