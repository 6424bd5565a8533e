"""Entry point: ep_fsdp training (expert parallel MoE; the reference lists EP as a goal only, README.md:7). Same flags as the reference trainer; see
distributed_pytorch_amd/config.py for the full list and distributed_pytorch_amd/trainer.py
for the loop."""
import os
import sys

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "../..")))

from distributed_pytorch_amd.config import configs_from_args  # noqa: E402
from distributed_pytorch_amd.trainer import train  # noqa: E402

if __name__ == "__main__":
    tc, mc = configs_from_args()
    train(tc, mc, mode="ep_fsdp")
