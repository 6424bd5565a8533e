"""Entry point: sample text from a checkpoint written by ``--save_model``.

    python single-gpu/sample.py --ckpt llm_model_ckpt.pt --prompt "Once upon a time" --max_new_tokens 200 \\
        --temperature 0.8 --topk 50 --num_samples 2

The checkpoint holds the model configuration next to the weights, so no model flags are needed. On a GPU the
model runs in bf16 and decodes from one captured hipGraph per token over a preallocated KV cache
(`LLM.generate`, reference: single-gpu/model.py:699-747 — the reference ships the method without a script
around it). The prompt is encoded with the tokeniser the data scripts use (GPT-2 BPE when available, bytes
otherwise); ``--token_ids`` prints raw ids instead of text, for vocabularies the tokeniser does not cover.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))

import torch  # noqa: E402

from distributed_pytorch_amd.config import LLMconfig, Trainconfig  # noqa: E402
from distributed_pytorch_amd.models.llm import LLM  # noqa: E402
from distributed_pytorch_amd.utils.tokenizer import get_encoder  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--ckpt", required=True, help="<file_name>_ckpt.pt written by a training script with --save_model")
    ap.add_argument("--prompt", type=str, default="\n")
    ap.add_argument("--max_new_tokens", type=int, default=200)
    ap.add_argument("--temperature", type=float, default=0.8)
    ap.add_argument("--topk", type=int, default=50)
    ap.add_argument("--num_samples", type=int, default=1, help="sequences sampled together (one batch)")
    ap.add_argument("--seed", type=int, default=1729)
    ap.add_argument("--device", type=str, default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--tokenizer", type=str, default="gpt2", choices=["gpt2", "bytes"])
    ap.add_argument("--token_ids", action="store_true", help="print token ids instead of decoded text")
    args = ap.parse_args(argv)

    # the checkpoint stores the two configuration objects next to the tensors: allow exactly those classes
    with torch.serialization.safe_globals([LLMconfig, Trainconfig]):
        ckpt = torch.load(args.ckpt, map_location="cpu", weights_only=True)
    cfg = LLMconfig(**ckpt["model_config_dict"]).finalize()
    model = LLM(cfg)
    model.load_state_dict(ckpt["model_state"])
    model.to(args.device)
    if args.device.startswith("cuda"):
        model.to(torch.bfloat16)

    enc = get_encoder(args.tokenizer)
    ids = [t for t in enc.encode_ordinary(args.prompt) if t < cfg.vocab_size] or [0]
    ids = ids[-(cfg.block_size - 1):]
    idx = torch.tensor([ids] * args.num_samples, dtype=torch.long, device=args.device)
    new = max(1, args.max_new_tokens)      # past the context the cache slides (eager decode)

    torch.manual_seed(args.seed)
    t0 = time.perf_counter()
    out = model.generate(idx, new, temperature=args.temperature, topk=min(args.topk, cfg.vocab_size) or None)
    if args.device.startswith("cuda"):
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    graphed = getattr(model, "last_decode_graphed", False)
    print(f"[sample] tokeniser {getattr(enc, 'name_', getattr(enc, 'name', '?'))}; {args.num_samples} x {new} tokens in "
          f"{dt:.2f} s ({args.num_samples * new / dt:.0f} tok/s, {'hipGraph' if graphed else 'eager'} decode)")
    for i, row in enumerate(out.tolist()):
        print(f"--- sample {i} ---")
        print(row if args.token_ids else enc.decode([t for t in row if t < enc.n_vocab]))
    return out


if __name__ == "__main__":
    main()
