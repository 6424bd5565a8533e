"""Notebook entry for ZeRO-1 training (reference: multi-gpu/zero1,2/kaggle-zero1.py:1-10, run there as
``!torchrun --standalone --nproc_per_node=2 kaggle-train.py ...``).

One cell, one command: started as a plain ``python kaggle-zero1.py <flags>`` this script launches one rank per
visible GPU itself (``--nproc N`` or DPA_NPROC to choose, ``--max_restarts R`` for elastic relaunches with
``--resume auto``) and streams the job's output; started under ``torchrun`` it is an ordinary rank. The training
flags are the reference script's; the model, the engine and the loop are the package's."""
import os
import sys

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "../..")))

from distributed_pytorch_amd.launch import self_launch  # noqa: E402

if __name__ == "__main__":
    self_launch()   # returns in a launched rank (and in a one-GPU job), otherwise runs the launcher and exits
    from distributed_pytorch_amd.config import configs_from_args
    from distributed_pytorch_amd.trainer import train
    tc, mc = configs_from_args()
    train(tc, mc, mode="zero1")
