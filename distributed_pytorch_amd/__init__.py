"""MI355X-native GPT pre-training framework (PyTorch-ROCm + HIP/CDNA4 kernels + RCCL)."""
from .config import LLMconfig, Trainconfig

__version__ = "0.1.0"
__all__ = ["LLMconfig", "Trainconfig", "__version__"]
