"""MI355X-native GPT pre-training framework (PyTorch-ROCm + HIP/CDNA4 kernels + RCCL)."""
import os as _os

# The ROCm runtime reads this when it starts, i.e. at the process's first HIP call (torch.cuda.is_available()
# included): dmabuf IPC is what lets the ranks of a node map one another's buffers for RCCL. Set at import time so
# that no later call is too late; an exported value wins.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from .config import LLMconfig, Trainconfig

__version__ = "0.1.0"
__all__ = ["LLMconfig", "Trainconfig", "__version__"]
