"""`from model import LLM, LLMconfig` as in the reference layout (reference: single-gpu/model.py,
multi-gpu/ddp/model.py): re-exports the one model implementation of this framework."""
import os
import sys

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                 "../.." if os.path.basename(os.path.dirname(os.path.dirname(
                                                     os.path.abspath(__file__)))) == "multi-gpu" else "..")))

from distributed_pytorch_amd.config import LLMconfig, Trainconfig  # noqa: E402,F401
from distributed_pytorch_amd.models.llm import (LLM, MLP, Attention, Block, Expert, FullMHLA, GQA, MoE,  # noqa: E402,F401
                                                NaiveMHLA)
