"""Entry point with the reference's notebook-script name: ZeRO-2 training, with real gradient sharding (reference: multi-gpu/zero1,2/kaggle-zero2.py). Same flags as the reference script of this name; the model, the engine and the loop are
shared (distributed_pytorch_amd/), so this file only selects the parallel mode."""
import os
import sys

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "../..")))

from distributed_pytorch_amd.config import configs_from_args  # noqa: E402
from distributed_pytorch_amd.trainer import train  # noqa: E402

if __name__ == "__main__":
    tc, mc = configs_from_args()
    train(tc, mc, mode="zero2")
